#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: share, calls, average per kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    nm = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  calls {r['Calls']:>6}  avg {float(r['AverageNs'])/1e3:8.1f} us  {nm[:110]}")
