#!/bin/bash
for v in 0 1 2 3 0 3; do
  SCRUBBY_HIP_STREAMS=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | grep "^{" > /tmp/o.json
  python3 -c "import json; d=json.load(open('/tmp/o.json')); print('streams', $v, d['value'], d['ms_per_step'], list(d['roofline']['stage_ms_per_step'].values()))"
done
