"""Kernel study: prints the kernel timeline (start offset, duration, gap to the previous kernel) of the tail of a rocprofv3
--kernel-trace --output-format csv run:  python tools/trace_timeline.py <dir> [last_n]"""
import csv, glob, os, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>9}  {r['Kernel_Name'][:70]}")
    prev_end = e
