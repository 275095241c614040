"""Kernel timeline of single scoring calls from a rocprofv3 kernel trace (sqlite output).
usage: python tools/call_timeline.py <results.db> [call indices...]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
conv = [i for i, r in enumerate(rows) if 'k_convert_q' in r[0]]
for ci in [int(x) for x in sys.argv[2:]] or [len(conv) // 2]:
    if ci + 1 >= len(conv):
        continue
    i0, i1 = conv[ci], conv[ci + 1]
    t0 = rows[i0][1]
    print(f'--- call {ci}: {(rows[i1][1] - t0) / 1e3:.1f} us to the next call')
    for r in rows[i0:i1]:
        print(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:8.1f}  {re.sub(r'[(].*', '', r[0])[:70]}")
