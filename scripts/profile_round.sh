#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench, then HBM traffic counters (separate --pmc passes, kernel-trace only).
# usage (on the GPU box): bash scripts/profile_round.sh r02 [sr] [ont] [k2]      -> gpurun_out/<round>_*; copy what is to be judged into profiles/
R=${1:-r02}; shift
WHAT=${*:-sr ont k2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in $WHAT; do
  case $W in
    sr)  ARGS="--no-secondary"; TAG=""; NREC=20000000; LAUNCHES=1; TOP=28 ;;
    ont) ARGS="--workload ont"; TAG="_ont"; NREC=1000000; LAUNCHES=2; TOP=20 ;;
    k2)  ARGS="--workload k2"; TAG="_k2"; NREC=40000000; LAUNCHES=1; TOP=12 ;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}${TAG}_stats -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu > gpurun_out/${R}${TAG}_bench_under_rocprof.log 2>&1 || exit 1
  python3 scripts/prof_summary.py gpurun_out/${R}${TAG}_stats $TOP > gpurun_out/${R}${TAG}_kernel_summary.txt
  cp $(find gpurun_out/${R}${TAG}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}${TAG}_kernel_stats.csv
  grep "^{" gpurun_out/${R}${TAG}_bench_under_rocprof.log > gpurun_out/${R}${TAG}_bench_under_rocprof.json
  echo "== $W: kernel stats done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}${TAG}_pmc_fetch -- python3 bench.py $ARGS --steps 1 --warmup 0 --no-cpu > gpurun_out/${R}${TAG}_pmc_fetch.log 2>&1 || exit 1
  echo "== $W: FETCH_SIZE done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}${TAG}_pmc_write -- python3 bench.py $ARGS --steps 1 --warmup 0 --no-cpu > gpurun_out/${R}${TAG}_pmc_write.log 2>&1 || exit 1
  NREC=$(grep "^{" gpurun_out/${R}${TAG}_pmc_write.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c.get('records_per_launch', c.get('records_per_gpu', $NREC)))")
  python3 scripts/pmc_summary.py gpurun_out/${R}${TAG}_pmc_fetch gpurun_out/${R}${TAG}_pmc_write > gpurun_out/${R}${TAG}_pmc_traffic.txt
  python3 scripts/make_traffic.py gpurun_out/${R}${TAG}_pmc_fetch gpurun_out/${R}${TAG}_pmc_write $NREC gpurun_out/traffic${TAG}.json $W $LAUNCHES
  head -14 gpurun_out/${R}${TAG}_kernel_summary.txt
done
