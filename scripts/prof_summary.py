"""Condense a rocprofv3 kernel_stats.csv / kernel_trace.csv pair into a short table."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
print(f"{'kernel':60s} {'calls':>5s} {'total_ms':>10s} {'avg_ms':>9s} {'pct':>6s} {'max_ms':>9s}")
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.2f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(10),
          ("%.3f" % (float(r["AverageNs"]) / 1e6)).rjust(9), r["Percentage"].rjust(6), ("%.3f" % (float(r["MaxNs"]) / 1e6)).rjust(9))
