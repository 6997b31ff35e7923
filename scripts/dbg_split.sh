#!/bin/bash
# timing experiment: bit0 = skip DP in the LDS sort kernels, bit1 = skip DP in the giant kernel (results invalid)
for d in 0 1 2 3; do
  SCRUBBY_HIP_DBG=$d timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep "^{" > /tmp/o.json
  python3 -c "import json; d=json.load(open('/tmp/o.json')); print('dbg', $d, d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
done
