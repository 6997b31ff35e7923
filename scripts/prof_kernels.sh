#!/bin/bash
# per-call kernel times of one bench step under a given environment: prof_kernels.sh TAG [bench args]
TAG=${1:-base}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pk_$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pk_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > gpurun_out/pk_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pk_$TAG/**/*kernel_trace.csv", recursive=True)[0]
sel = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_") and not any(x in n for x in ("synth", "ref_", "table", "multi", "fa_")):
        sel[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("== $TAG")
for k, v in sorted(sel.items(), key=lambda kv: -sum(kv[1])):
    n = max(len(v) // 3, 1)
    if sum(v[-n:]) > 0.5: print(k.ljust(30), " ".join(f"{x:.1f}" for x in v[-n:]))
PY
grep "^{" gpurun_out/pk_$TAG.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], {k: d['result'][k] for k in ('pair_decided','ext_shortcut_reads','ext_reads')})"
