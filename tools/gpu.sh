#!/bin/bash
# One parametrised recipe file for everything that is run on the GPU box through gpurun (it replaces the per-experiment
# tools/r3_*.sh scripts of round 3).  Output goes to gpurun_out/<tag>/ (scratch; what is to be judged is copied to profiles/).
#
#   tools/gpu.sh <tag> <recipe> [args...] [-- <recipe> [args...]]...
#
# recipes
#   tests [pytest args]         pytest -m gpu in ONE process (default: the whole suite, -x -q)
#   host [args]                 tools/host_api_rate.py (PCIe-inclusive rates of the host-buffer entry points)
#   bench [args]                python3 bench.py args  -> bench_<n>.json
#   stats [bench args]          rocprofv3 --kernel-trace --stats of bench.py --quick args
#   pmc <counters> [bench args] one rocprofv3 --pmc pass (counters comma-separated) of bench.py --quick args
#   timeline [bench args]       rocprofv3 --kernel-trace of bench.py --quick args -> timeline_<n>.txt (tools/pipeline_timeline.py: which
#                               kernels of step k+1 run beside which of step k) + kernel_stats_<n>.csv
#   traffic <model> [bench args] the evidence set of one model: kernel stats + FETCH_SIZE pass + WRITE_SIZE pass (separate runs, as
#                               MI355X_MICROARCH.md prescribes) of bench.py --quick --steps 2 --warmup 1 --model <model> args
#                               -> traffic_<model>.json (entries in the format of profiles/r*_traffic.json) + the per-pass summaries
#   pmcpy <counters> <script> [args]   one rocprofv3 --pmc pass of python3 <script> args (e.g. tools/decode_rate.py); per-kernel sums
#   statspy <script> [args]     rocprofv3 --kernel-trace --stats of python3 <script> args
#   decode [model] [bytes]      tools/decode_rate.py
#   py <script> [args]          python3 <script> args
#   ubench <file.hip> [args]    hipcc a tools/*.hip microbenchmark and run it
# Steps are joined with && semantics: the first failing step ends the call (no GPU step after a failed one).
set -o pipefail
TAG=${1:?tag}; shift
DST=$PWD/gpurun_out/$TAG; mkdir -p "$DST"
export TMPDIR=/tmp
n=0
run_recipe() {
    local r=$1; shift
    n=$((n + 1))
    case $r in
    tests)
        if [ $# -eq 0 ]; then set -- tests -x -q; fi
        timeout -k 10 1100 python3 -m pytest -m gpu "$@" > "$DST/pytest_$n.txt" 2>&1; local rc=$?
        echo "pytest rc=$rc"; tail -8 "$DST/pytest_$n.txt"; return $rc ;;
    host)
        timeout -k 10 900 python3 tools/host_api_rate.py "$@" --json "$DST/host_$n.json" > "$DST/host_$n.txt" 2>&1; local rc=$?
        cat "$DST/host_$n.txt" | grep -v '^{' | tail -20; return $rc ;;
    bench)
        timeout -k 10 1100 python3 bench.py "$@" > "$DST/bench_$n.json" 2> "$DST/bench_$n.err"; local rc=$?
        echo "bench rc=$rc"; python3 - "$DST/bench_$n.json" <<'EOF'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ("value", "ms_per_step")}, "roofline", {k: r["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_ms")})
    for k in ("host_path", "roofline_solo", "decode", "one_call_at_a_time"):
        if r.get(k): print(k, json.dumps(r[k])[:600])
except Exception as e:
    print("no JSON line:", e)
EOF
        tail -3 "$DST/bench_$n.err"; return $rc ;;
    stats)
        (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$DST/stats_$n" -o run -- python3 "$OLDPWD/bench.py" --quick "$@" > "$DST/stats_$n.json" 2> "$DST/stats_$n.err"); local rc=$?
        find "$DST/stats_$n" -name '*kernel_stats.csv' | head -1 | xargs -r head -25; return $rc ;;
    pmc)
        local ctr=$1; shift
        (cd /tmp && timeout -k 10 900 rocprofv3 --pmc ${ctr//,/ } --output-format csv -d "$DST/pmc_${n}" -o run -- python3 "$OLDPWD/bench.py" --quick "$@" > "$DST/pmc_$n.json" 2> "$DST/pmc_$n.err"); local rc=$?
        python3 tools/pmc_sum.py "$DST/pmc_$n" 2>&1 | tail -40; return $rc ;;
    timeline)
        (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$DST/kt_$n" -o kt -- python3 "$OLDPWD/bench.py" --steps 6 --warmup 2 --quick "$@" > "$DST/timeline_$n.json" 2> "$DST/timeline_$n.err"); local rc=$?
        find "$DST/kt_$n" -name '*kernel_stats.csv' -exec cp {} "$DST/kernel_stats_$n.csv" \;
        local tr=$(find "$DST/kt_$n" -name '*kernel_trace.csv' | head -1)
        [ -n "$tr" ] && python3 tools/pipeline_timeline.py "$tr" 260 > "$DST/timeline_$n.txt"
        rm -rf "$DST/kt_$n"; tail -45 "$DST/timeline_$n.txt"; return $rc ;;
    traffic)
        local model=$1; shift
        local A="--quick --steps 2 --warmup 1 --model $model $*"
        (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$DST/tr_${model}_kt" -o kt -- python3 "$OLDPWD/bench.py" $A > "$DST/tr_${model}_kt.json" 2> "$DST/tr_${model}_kt.err") || return 1
        find "$DST/tr_${model}_kt" -name '*kernel_stats.csv' -exec cp {} "$DST/${model}_kernel_stats.csv" \;
        rm -rf "$DST/tr_${model}_kt"
        for c in FETCH_SIZE WRITE_SIZE; do
            (cd /tmp && timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d "$DST/tr_${model}_$c" -o p -- python3 "$OLDPWD/bench.py" $A > "$DST/tr_${model}_$c.json" 2> "$DST/tr_${model}_$c.err") || return 1
            python3 tools/pmc_sum.py "$DST/tr_${model}_$c" --json "$DST/${model}_pmc_$c.json" > /dev/null
            rm -rf "$DST/tr_${model}_$c"
        done
        python3 tools/traffic_json.py "$model" "$DST" "$@" | tail -30; return $? ;;
    pmcpy)
        local ctr=$1; shift
        (cd /tmp && timeout -k 10 900 rocprofv3 --pmc ${ctr//,/ } --output-format csv -d "$DST/pmcpy_${n}" -o run -- python3 "$OLDPWD/$1" "${@:2}" > "$DST/pmcpy_$n.txt" 2> "$DST/pmcpy_$n.err"); local rc=$?
        python3 tools/pmc_sum.py "$DST/pmcpy_$n" --json "$DST/pmcpy_$n.json" 2>&1 | tail -30; rm -rf "$DST/pmcpy_$n"; tail -2 "$DST/pmcpy_$n.txt"; return $rc ;;
    statspy)
        (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$DST/statspy_$n" -o run -- python3 "$OLDPWD/$1" "${@:2}" > "$DST/statspy_$n.txt" 2> "$DST/statspy_$n.err"); local rc=$?
        find "$DST/statspy_$n" -name '*kernel_stats.csv' -exec cp {} "$DST/statspy_${n}_kernel_stats.csv" \;
        rm -rf "$DST/statspy_$n"; head -12 "$DST/statspy_${n}_kernel_stats.csv"; tail -2 "$DST/statspy_$n.txt"; return $rc ;;
    decode)
        timeout -k 10 900 python3 tools/decode_rate.py "$@" > "$DST/decode_$n.txt" 2>&1; local rc=$?
        tail -12 "$DST/decode_$n.txt"; return $rc ;;
    py)
        timeout -k 10 1100 python3 "$@" > "$DST/py_$n.txt" 2>&1; local rc=$?
        tail -40 "$DST/py_$n.txt"; return $rc ;;
    ubench)
        local src=$1; shift
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o "/tmp/ub_$n" "$src" && timeout -k 10 600 "/tmp/ub_$n" "$@" > "$DST/ubench_$n.txt" 2>&1; local rc=$?
        tail -40 "$DST/ubench_$n.txt"; return $rc ;;
    *) echo "unknown recipe $r"; return 2 ;;
    esac
}
args=()
while [ $# -gt 0 ]; do
    if [ "$1" == "--" ]; then
        run_recipe "${args[@]}" || { echo "step failed: ${args[*]}"; exit 1; }
        args=()
    else
        args+=("$1")
    fi
    shift
done
[ ${#args[@]} -gt 0 ] && { run_recipe "${args[@]}" || { echo "step failed: ${args[*]}"; exit 1; }; }
echo "all steps ok"
