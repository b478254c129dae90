"""Per-kernel table of the replayed training step from a rocprofv3 --kernel-trace CSV.

The replays of a captured step are the equally long intervals between the optimiser's once-per-step kernel at the
end of the trace.  Prints launches per step, average and summed time
per kernel symbol over the last N replays.

usage: python tools/graph_kernel_table.py <trace_kernel_trace.csv> [replays=10] > table.csv
"""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r'\(.*\)$', '', name).replace('void ', '')
    return name[:110]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    # steps are delimited by a kernel that runs once per step (the optimiser's step counter); the replayed steps
    # are the intervals of equal length at the end of the trace
    anchor = [i for i, n in enumerate(names) if 'adam_advance_kernel' in n]
    if len(anchor) < 3:
        raise SystemExit('fewer than three optimiser steps in the trace')
    period = anchor[-1] - anchor[-2]
    n_ok = 0
    while n_ok + 2 <= len(anchor) - 1 and anchor[-1 - n_ok] - anchor[-2 - n_ok] == period:
        n_ok += 1
    n_rep = min(n_rep, n_ok)
    tail = rows[anchor[-1 - n_rep] + 1:anchor[-1] + 1]
    agg = OrderedDict()
    for r in tail:
        k = short(r['Kernel_Name'])
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += d
    span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / n_rep / 1e3
    summed = sum(a[1] for a in agg.values()) / n_rep / 1e3
    print(f'# last {n_rep} replays: {period} kernels per step, {span:.1f} us per step first start to last end, '
          f'{summed:.1f} us summed kernel time')
    print('kernel,launches_per_step,avg_us,us_per_step')
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%.1f,%.2f,%.1f' % (k, n / n_rep, d / n / 1e3, d / n_rep / 1e3))


if __name__ == '__main__':
    main()
