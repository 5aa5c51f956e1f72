#!/bin/bash
# [coder beside rank kernels] phase: priorities and shapes
for t in "--tune 0" "--tune 128" "--tune 16" "--tune 144" "--tune 8" "--tune 64" "--tune 192"; do
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 3 --quick --no-verify $t > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python3 -c "
import json
d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][0])
k={r['kernel'].split(' ')[0]: r['avg_launch_ms'] for r in d['roofline_kernels']}
print('$t', d['value'], d['ms_per_step'], k)"
done
