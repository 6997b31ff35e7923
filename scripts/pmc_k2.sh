#!/bin/bash
# instruction / wait counters of k_k2_classify (bench.py --workload k2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-8000000}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/k2_pmc1 -- python3 bench.py --workload k2 --records $N --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD --output-format csv -d gpurun_out/k2_pmc2 -- python3 bench.py --workload k2 --records $N --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/k2_pmc1", "gpurun_out/k2_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if "k_k2_classify" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
        for k, v in acc.items():
            print(d.split("/")[-1], k, f"{v:.4g}")
PY
