#!/bin/bash
# per-kernel time of the default bench (rocprofv3 --kernel-trace --stats), top kernels only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pq_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pq_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/pq_bench.log 2>&1
python3 scripts/prof_summary.py gpurun_out/pq_stats 22
grep "^{" gpurun_out/pq_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
