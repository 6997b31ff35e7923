#!/bin/bash
# A/B of the parallel fill: kernel stats with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3pfA -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r3pfA.log 2>&1 || exit 1
python3 scripts/prof_summary.py gpurun_out/r3pfA 16 > gpurun_out/r3pfA_summary.txt
export SCRUBBY_HIP_NO_TOPBT=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3pfB -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r3pfB.log 2>&1 || exit 1
python3 scripts/prof_summary.py gpurun_out/r3pfB 16 > gpurun_out/r3pfB_summary.txt
