#!/bin/bash
# kernel stats of one bench run: bash scripts/prof_one.sh TAG [bench args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG} -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > gpurun_out/${TAG}.log 2>&1 || exit 1
python3 scripts/prof_summary.py gpurun_out/${TAG} 24 > gpurun_out/${TAG}_summary.txt
