#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench, then HBM traffic counters (separate --pmc passes).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/${R}_bench_under_rocprof.log 2>&1
python3 scripts/prof_summary.py gpurun_out/${R}_stats 24 > gpurun_out/${R}_kernel_summary.txt
cp $(find gpurun_out/${R}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_kernel_stats.csv
grep "^{" gpurun_out/${R}_bench_under_rocprof.log > gpurun_out/${R}_bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python3 scripts/pmc_summary.py gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write > gpurun_out/${R}_pmc_traffic.txt
cat gpurun_out/${R}_kernel_summary.txt | head -16; cat gpurun_out/${R}_pmc_traffic.txt
