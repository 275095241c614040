"""Print the last call's kernel timeline (start offset, duration, gap to the previous kernel) from a
rocprofv3 --kernel-trace output directory: python tools/trace_timeline.py DIR [N_KERNELS]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us")
