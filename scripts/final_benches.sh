#!/bin/bash
# the round's bench lines, one after the other: bash scripts/final_benches.sh r03   -> gpurun_out/<round>_*_bench.{log,json}
R=${1:-r03}
run() { tag=$1; shift; timeout -k 10 $TMO python bench.py "$@" > gpurun_out/${R}_${tag}.log 2>&1 || { echo "FAILED $tag"; tail -3 gpurun_out/${R}_${tag}.log; return 1; }; grep "^{" gpurun_out/${R}_${tag}.log | tail -1 > gpurun_out/${R}_${tag}.json; python - <<PY
import json; d=json.load(open("gpurun_out/${R}_${tag}.json")); print("${tag}", d.get("value"), d.get("unit"), d.get("ms_per_step"))
PY
}
TMO=300 run bench --steps 5 --warmup 1 --no-secondary &&
TMO=200 run chain_only_bench --chain-only --steps 5 --warmup 1 &&
TMO=300 run srdiv_bench --workload sr-div --steps 3 --warmup 1 &&
TMO=200 run k2_bench --workload k2 &&
TMO=120 run shard_10m --records 10000000 --no-cpu --steps 5 &&
TMO=120 run shard_5m --records 5000000 --no-cpu --steps 5 &&
TMO=120 run shard_2p5m --records 2500000 --no-cpu --steps 5 &&
TMO=500 run ont_bench --workload ont --steps 2 --warmup 1 --cpu-seconds 20 &&
TMO=400 run e2e_bench --workload e2e
