#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... -- [bench args]; prints value / ms per step / stage split for each setting
VAR=$1; shift
VALS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VALS+=("$1"); shift; done
shift
for v in "${VALS[@]}"; do
  env $VAR=$v python3 bench.py --steps 5 --warmup 2 --no-cpu "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'], d['result']['reads_removed_rank0'])"
done
