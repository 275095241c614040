"""Print per-kernel averages from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if len(sys.argv) < 3 or any(k in r["Name"] for k in sys.argv[2:]):
        print(f'{r["Name"][:64]:64s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
