#!/bin/bash
# kernel-time split of the long-read stand-in (bench.py --workload ont)
R=${1:-r01}; N=${2:-65536}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_ont_stats -- python3 bench.py --workload ont --records $N --steps 2 --warmup 1 --no-cpu > gpurun_out/${R}_ont_bench.log 2>&1
python3 scripts/prof_summary.py gpurun_out/${R}_ont_stats 24 > gpurun_out/${R}_ont_kernel_summary.txt
grep "^{" gpurun_out/${R}_ont_bench.log | cut -c1-600
head -30 gpurun_out/${R}_ont_kernel_summary.txt
