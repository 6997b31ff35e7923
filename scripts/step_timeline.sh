set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# kernel timeline (start, duration, queue, name) of ONE timed step of the default bench: usage  bash scripts/step_timeline.sh <tag> [bench args]
v=${1:-step}; shift
for once in 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$v -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-secondary "$@" > gpurun_out/prof_$v.log 2>&1
  f=$(find gpurun_out/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/prof_${v}_stats.csv
  t=$(find gpurun_out/prof_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$t" gpurun_out/prof_${v}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [i for i, r in enumerate(rows) if "k_sketch_probe" in r["Kernel_Name"] or "k_long_sketch" in r["Kernel_Name"]]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
big = [i for i in ks if dur(rows[i]) > 1000000]      # the timed launches' sketches (long reads: two kernels a launch, milliseconds apart)
i0 = big[-1]
while i0 - 1 in big or (big.index(i0) > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[big[big.index(i0) - 1]]["End_Timestamp"]) < 5000000): i0 = big[big.index(i0) - 1]
i1 = min([i for i in ks if i > i0 and int(rows[i]["Start_Timestamp"]) - int(rows[i0]["End_Timestamp"]) > 50000000] + [len(rows)])
t0 = int(rows[i0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in rows[i0:i1]:
        f.write("%9.3f %9.3f q%s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, dur(r) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
  rm -rf gpurun_out/prof_$v
done
