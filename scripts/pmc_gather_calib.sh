#!/bin/bash
# Calibration of FETCH_SIZE for this path's access pattern (MI355X_MICROARCH.md, HBM section: widths other than wide streams are
# uncalibrated): sh_bench_gather issues a known number of random 16-B slot gathers over the index table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gather
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_gather -- python3 bench.py --steps 1 --warmup 0 --no-cpu --gather-bench > gpurun_out/pmc_gather.log 2>&1
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/pmc_gather/**/*counter_collection.csv", recursive=True)[0]
tot, calls = 0.0, 0
for r in csv.DictReader(open(f)):
    if "gather" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        tot += float(r["Counter_Value"]) * 1024.0; calls += 1
line = [l for l in open("gpurun_out/pmc_gather.log") if l.startswith("{")][-1]
g = json.loads(line)["gather_ceiling"]
probes = g["probes"] * calls
print(f"gather kernel launches {calls}, probes {probes}, FETCH_SIZE {tot / 1e9:.2f} GB -> {tot / probes:.1f} B per 16-B probe (a 64-B sector each would be 64.0); ceiling {g}")
PY
