#!/bin/bash
# timing experiments: SCRUBBY_HIP_DBG values given as arguments, one rocprof run each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in "$@"; do
  SCRUBBY_HIP_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3exp_$D -- python3 bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/r3exp_$D.log 2>&1 || exit 1
  echo "== dbg $D"; python3 scripts/timeline.py gpurun_out/r3exp_$D 3 | grep "k_giant"
done
