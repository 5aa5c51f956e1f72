"""Per-kernel sums of a rocprofv3 --pmc pass: python3 tools/pmc_sum.py <output dir> [--json FILE]
(one line per w3:: kernel and counter: launches, value per launch)."""
import collections
import csv
import glob
import json
import sys


def summarise(out):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "w3::" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    res = {}
    for k, d in acc.items():
        n = max(len(launches[k]), 1)
        res[k] = {"launches": n, "per_launch": {c: v / n for c, v in d.items()}}
    return res


if __name__ == "__main__":
    res = summarise(sys.argv[1])
    for k, d in sorted(res.items()):
        print(k[-60:], "launches", d["launches"])
        for c, v in d["per_launch"].items():
            print("    %-28s %.5g per launch" % (c, v))
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
