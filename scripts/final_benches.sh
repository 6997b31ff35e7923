#!/bin/bash
# round 5: the bench lines and profiles kept under profiles/ (run on the GPU box from the repo root)
R=r05
run() { tag=$1; shift; timeout -k 10 $TMO python bench.py "$@" > gpurun_out/${R}_${tag}.log 2>&1 || { echo "FAILED $tag"; tail -3 gpurun_out/${R}_${tag}.log; return 1; }; grep "^{" gpurun_out/${R}_${tag}.log | tail -1 > gpurun_out/${R}_${tag}.json; python - <<PY
import json; d=json.load(open("gpurun_out/${R}_${tag}.json")); print("${tag}", d.get("value"), d.get("unit"), d.get("ms_per_step"))
PY
}
TMO=300 run bench --steps 5 --warmup 1 --no-secondary &&
TMO=200 run chain_only_bench --chain-only --steps 5 --warmup 1 &&
TMO=300 run srdiv_bench --workload sr-div --steps 3 --warmup 1 &&
TMO=200 run k2_bench --workload k2 &&
TMO=120 run shard_10m --records 10000000 --no-cpu --steps 5 --no-secondary &&
TMO=120 run shard_5m --records 5000000 --no-cpu --steps 5 --no-secondary &&
TMO=120 run shard_2p5m --records 2500000 --no-cpu --steps 5 --no-secondary &&
TMO=700 run ont_bench --workload ont --steps 2 --warmup 1 --cpu-seconds 60 &&
SCRUBBY_HIP_RMQ_EXACT_MAX=4096 TMO=600 run ont_open_ties --workload ont --steps 1 --warmup 1 --cpu-seconds 40 &&
TMO=400 run e2e_bench --workload e2e
python - <<PY
import json
for tag, out in (("ont_bench", "ont_stratified"), ("ont_open_ties", "ont_open_ties_stratified")):
    try:
        d = json.load(open("gpurun_out/r05_%s.json" % tag))
        json.dump({"command": "bench.py --workload ont" + (" (SCRUBBY_HIP_RMQ_EXACT_MAX=4096: ties of reads above 4096 chain anchors left open)" if "open" in tag else ""), "value_reads_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                   "result": {k: d["result"][k] for k in ("reads_removed", "rmq_rechained", "rmq_tied", "rmq_exact", "rmq_open", "ext_unresolved", "ext_ondemand", "locus_redone")},
                   "stratified_parity": d["stratified_parity"], "cpu_baseline": d["cpu_baseline"]}, open("gpurun_out/r05_%s.json" % out, "w"), indent=1)
    except Exception as e:
        print("no", tag, e)
PY
