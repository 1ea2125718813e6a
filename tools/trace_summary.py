#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per-kernel totals per step and one step's timeline."""
import csv, sys, collections
path = sys.argv[1]
show = len(sys.argv) > 2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, nm in enumerate(names) if 'rel_lp_fwd' in nm]
a, b = idx[-3], idx[-2]
step = rows[a:b]
wall = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3
agg = collections.OrderedDict()
for r in step:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    c, t = agg.get(nm, (0, 0.0))
    agg[nm] = (c + 1, t + d)
print(f"one step: {len(step)} launches, wall {wall:.1f} us")
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us  {c:3d} x {t/c:6.1f}  {nm}")
if show:
    t0 = int(step[0]['Start_Timestamp'])
    for r in step:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} "
              f"{r['Kernel_Name'].replace('(anonymous namespace)::','')[:80]} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg={r['Workgroup_Size_X']}")
