#!/bin/bash
# kernel timeline of one small step (per-call fixed cost of the repeat path): rocprofv3 kernel trace, last step only
N=${1:-200000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_small
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_small -- python3 bench.py --records $N --steps 1 --warmup 1 --no-cpu > gpurun_out/trace_small.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_small/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last k_sketch_probe
last = max(i for i, r in enumerate(rows) if "k_sketch_probe" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
prev_end = t0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4}  {name}")
    prev_end = max(prev_end, e)
PY
