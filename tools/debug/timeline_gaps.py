"""Idle time and per-kernel sums of the last replayed steps in a rocprofv3 --kernel-trace CSV.

usage: python tools/debug/timeline_gaps.py <kernel_trace.csv> [steps=6]
A step ends with the optimiser's once-per-step kernel.  Prints, per step averaged: wall time, busy time (union of
kernel intervals), the idle gaps longer than 3 us with the kernels on either side, and kernel counts."""
import csv
import re
import sys
from collections import Counter


def short(n):
    return re.sub(r'\(.*\)$', '', n).replace('void ', '')[:70]


rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r['Start_Timestamp']))
anchor = [i for i, r in enumerate(rows) if 'adam_advance_kernel' in r['Kernel_Name']]
steps = [rows[anchor[-1 - k - 1] + 1:anchor[-1 - k] + 1] for k in range(nsteps)]
wall = busy = 0.0
gaps = Counter()
gapsum = Counter()
for st in steps:
    t0, t1 = int(st[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in st)
    wall += t1 - t0
    end, prev = t0, None
    for r in st:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > end:
            if s - end > 3000 and prev is not None:
                key = (short(prev), short(r['Kernel_Name']))
                gaps[key] += 1
                gapsum[key] += s - end
        else:
            s = end
        if e > end:
            busy += e - s
            end, prev = e, r['Kernel_Name']
print(f'{len(steps[0])} kernels per step, wall {wall / nsteps / 1e3:.1f} us, busy {busy / nsteps / 1e3:.1f} us, '
      f'idle {(wall - busy) / nsteps / 1e3:.1f} us')
for key, n in sorted(gaps.items(), key=lambda kv: -gapsum[kv[0]])[:25]:
    print(f'{gapsum[key] / nsteps / 1e3:8.1f} us/step  x{n / nsteps:.1f}  after {key[0]}  before {key[1]}')
