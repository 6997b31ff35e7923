"""Kernel timeline of the last full-size step in a rocprofv3 kernel trace: python scripts/timeline.py gpurun_out/<dir> [min_ms]"""
import csv, glob, sys
d = sys.argv[1]; min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
import os
f = max(glob.glob(f'{d}/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sk = [i for i, r in enumerate(rows) if 'k_sketch_probe' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 15e6]
i0 = sk[-1]; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if s > 400: break
    if dur > min_ms: print('  %8.2f +%7.2f ms  %s' % (s, dur, r['Kernel_Name'][:60]))
