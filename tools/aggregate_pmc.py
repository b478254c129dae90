"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel symbol.

usage: python tools/aggregate_pmc.py <dir with *_counter_collection.csv> [<dir> ...] > table.md
Counters are summed over dispatches of the same kernel; derived columns:
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)   (GUI_ACTIVE is summed over 8 XCDs)
  hbm bytes = 2*FETCH_SIZE*1024 (gfx950 reports half the bytes of wide streaming reads) + WRITE_SIZE*1024
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                c = r['Counter_Name']
                agg[k][c] += float(r['Counter_Value'])
                calls[k][c] += 1
    rows = []
    for k, cs in agg.items():
        n = max(calls[k].values())
        gui = cs.get('GRBM_GUI_ACTIVE', 0.0)
        mfma = cs.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        util = mfma / (gui / 8 * 256 * 4) if gui else float('nan')
        fetch = cs.get('FETCH_SIZE')
        write = cs.get('WRITE_SIZE')
        rows.append((gui, k, n, util, cs.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0),
                     None if fetch is None else 2 * fetch * 1024 / n, None if write is None else write * 1024 / n))
    rows.sort(reverse=True)
    print('| kernel | dispatches | MFMA busy / (active cycles x SIMDs) | f32 MFMA MOPS per dispatch | HBM read B/dispatch (2xFETCH_SIZE) | HBM write B/dispatch |')
    print('|---|---|---|---|---|---|')
    for gui, k, n, util, mops, rd, wr in rows[:40]:
        f = lambda v: '-' if v is None else f'{v:,.0f}'
        print(f'| `{k[:90]}` | {n} | {util:.3f} | {mops / n:,.0f} | {f(rd)} | {f(wr)} |')


if __name__ == '__main__':
    main()
