cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 0 3; do
  SCRUBBY_HIP_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3ab_dbg${D} -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r3ab_dbg${D}.log 2>&1 || exit 1
  python3 scripts/prof_summary.py gpurun_out/r3ab_dbg${D} 30 > gpurun_out/r3ab_dbg${D}_summary.txt
done
