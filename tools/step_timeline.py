"""Kernel timeline of two steady-state bench steps from a rocprofv3 --kernel-trace directory (start, end,
duration in us, stream): python tools/step_timeline.py DIR"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_mfma_scan<" in r["Kernel_Name"] and ", 0, " in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1e6]
i0, i1 = idx[-4], idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %9.1f dur %8.1f s%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r["Stream_Id"], r["Kernel_Name"][:56]))
