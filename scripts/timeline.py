"""Per-call kernel timeline of the last step of a rocprofv3 --kernel-trace run (start ms, duration ms, kernel), calls longer than a threshold."""
import csv, glob, sys
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0; marker = sys.argv[3] if len(sys.argv) > 3 else "k_long_segtable"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(marker)]
s = idx[-1] if idx else 0
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:]:
    st = (int(r["Start_Timestamp"]) - t0) / 1e6; du = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if du > thr: print(f"{st:9.1f} {du:8.1f}  {r['Kernel_Name'][:80]}")
