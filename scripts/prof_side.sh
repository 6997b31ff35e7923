#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for SIDE in 0 1; do
  for NREC in 20000000 2500000; do
    SCRUBBY_HIP_SIDE=$SIDE python3 bench.py --records $NREC --steps 4 --warmup 1 --no-cpu > gpurun_out/r3side_${SIDE}_${NREC}.log 2>&1 || exit 1
    echo "side $SIDE records $NREC: $(grep -h '^{' gpurun_out/r3side_${SIDE}_${NREC}.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["result"]["reads_removed"])')"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3exp_0 -- python3 bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/r3exp_0.log 2>&1
python3 scripts/timeline.py gpurun_out/r3exp_0 3
