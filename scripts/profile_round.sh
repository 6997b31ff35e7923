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
python3 scripts/make_traffic.py gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write 20000000 gpurun_out/traffic.json
cat gpurun_out/${R}_kernel_summary.txt | head -16; cat gpurun_out/${R}_pmc_traffic.txt

# the two non-headline workloads: kernel-time split only
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_ont_stats -- python3 bench.py --workload ont --steps 3 --warmup 1 --no-cpu > gpurun_out/${R}_ont_under_rocprof.log 2>&1
python3 scripts/prof_summary.py gpurun_out/${R}_ont_stats 20 > gpurun_out/${R}_ont_kernel_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_k2_stats -- python3 bench.py --workload k2 --steps 3 --warmup 1 --no-cpu > gpurun_out/${R}_k2_under_rocprof.log 2>&1
python3 scripts/prof_summary.py gpurun_out/${R}_k2_stats 12 > gpurun_out/${R}_k2_kernel_summary.txt
head -8 gpurun_out/${R}_ont_kernel_summary.txt gpurun_out/${R}_k2_kernel_summary.txt
