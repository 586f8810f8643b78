#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per-kernel timeline of the first FRI commit at a given grid."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else 'k_leaf_subtree'
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 0
norm = lambda k: k.split('(')[0].replace('void ', '').split('<')[0]
out = [(norm(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X']), int(r['Workgroup_Size_X'])) for r in rows]
idx = [i for i, o in enumerate(out) if o[0] == first]
occ = min(occ, max(0, len(idx) - 2))          # default to the last complete commit but one when there are fewer
a, b = idx[occ], (idx[occ + 1] if occ + 1 < len(idx) else len(out))
prev = None; tot = 0
for o in out[a:b]:
    gap = (o[1] - prev) / 1e3 if prev else 0
    print(f"{o[0]:24s} grid={o[3]:9d} wg={o[4]:5d} dur={(o[2]-o[1])/1e3:9.1f} us gap={gap:8.1f} us")
    prev = o[2]; tot += o[2] - o[1]
print("sum kernel us", tot / 1e3, "span us", (out[b - 1][2] - out[a][1]) / 1e3)
