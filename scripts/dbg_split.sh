#!/bin/bash
# timing experiment on the giant-read kernel: 2 = skip DP, 4 = skip LDS chunk sorts, 8 = skip merge rounds (results invalid)
for d in 0 2 6 10 14; do
  SCRUBBY_HIP_DBG=$d timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep "^{" > /tmp/o.json
  python3 -c "import json; d=json.load(open('/tmp/o.json')); print('dbg', $d, d['ms_per_step'], list(d['roofline']['stage_ms_per_step'].values()))"
done
