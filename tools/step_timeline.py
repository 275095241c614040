"""Kernel timeline of two steady-state bench steps from a rocprofv3 --kernel-trace directory (start, end,
duration in us, stream): python tools/step_timeline.py DIR [main-pass kernel name [index of the first launch shown]]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "k_mfma_scan16<12, 4, 64>"  # the headline's main pass
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1e6]
first = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 4  # which launch to start at
i0, i1 = idx[first], idx[first + 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %9.1f dur %8.1f s%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r["Stream_Id"], r["Kernel_Name"][:56]))
