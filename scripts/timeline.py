#!/usr/bin/env python3
"""Print the kernel timeline of the last search in a rocprofv3 --kernel-trace csv."""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'), key=lambda p: __import__('os').path.getmtime(p))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
last = 'select_kernel' if any('select_kernel' in n and 'coarse' not in n for n in names) else 'final_merge_kernel'
idx = [i for i, n in enumerate(names) if last in n and 'coarse' not in n]
seg = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0
for r in seg:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    n = re.sub(r'vi::\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)[:50]
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}  grid={r['Grid_Size_X']}")
    tot += e - s
print('sum kernel us', tot / 1e3, 'span', (int(seg[-1]['End_Timestamp']) - t0) / 1e3)
