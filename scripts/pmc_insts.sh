#!/bin/bash
# VALU / wait counters per kernel of the default bench (one step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-sr}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pi1 -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pi2 -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for d in ("gpurun_out/pi1", "gpurun_out/pi2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if d.endswith("pi1"):
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                dur[row["Kernel_Name"].split("(")[0][:40]] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
print("%-42s %8s " % ("kernel", "ms") + " ".join("%10s" % n[3:13] for n in names) + "  valu_ms  wait%")
for k in sorted(dur, key=lambda k: -dur[k])[:16]:
    a = acc[k]
    valu_ms = a["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3
    wait = 100 * a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] else 0
    print("%-42s %8.2f " % (k, dur[k]) + " ".join("%10.3g" % a[n] for n in names) + "  %7.2f  %5.1f" % (valu_ms, wait))
PY
