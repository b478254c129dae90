"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel symbol.

usage: python tools/aggregate_pmc.py <dir with *_counter_collection.csv> [<dir> ...] [--json=out.json] [--steps=N] > table.md
--steps: the steps the profiled command ran (warm-up included); adds the HBM bytes of one step, all kernels, to the JSON
Counters are summed over dispatches of the same kernel; derived columns:
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)   (GUI_ACTIVE is summed over 8 XCDs)
  hbm bytes = 2*FETCH_SIZE*1024 (gfx950 reports half the bytes of wide streaming reads) + WRITE_SIZE*1024
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def bench_label(name):
    """the key bench.py's kernel table uses: the demangled symbol without return type and parameter list"""
    return re.sub(r'\(.*\)$', '', name).replace('void ', '').strip()


def main():
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    json_out = [a[7:] for a in sys.argv[1:] if a.startswith('--json=')]
    steps = [int(a[8:]) for a in sys.argv[1:] if a.startswith('--steps=')]      # steps the profiled command ran
    for d in args:
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
    if json_out:
        table = {}
        for gui, k, n, util, mops, rd, wr in rows:
            lab = bench_label(k)
            if lab is None or rd is None or wr is None:
                continue
            e = table.setdefault(lab, {'dispatches': 0, 'hbm_read_bytes': 0.0, 'hbm_write_bytes': 0.0,
                                       'mfma_busy_frac_weighted': 0.0, 'gui': 0.0})
            e['dispatches'] += n
            e['hbm_read_bytes'] += rd * n
            e['hbm_write_bytes'] += wr * n
            e['mfma_busy_frac_weighted'] += util * gui
            e['gui'] += gui
        out = {}
        for lab, e in table.items():
            out[lab] = {'dispatches': e['dispatches'],
                        'hbm_bytes_per_launch': round((e['hbm_read_bytes'] + e['hbm_write_bytes']) / e['dispatches']),
                        'hbm_read_bytes_per_launch': round(e['hbm_read_bytes'] / e['dispatches']),
                        'hbm_write_bytes_per_launch': round(e['hbm_write_bytes'] / e['dispatches']),
                        'mfma_busy_frac': round(e['mfma_busy_frac_weighted'] / e['gui'], 4) if e['gui'] else None}
        json.dump({'note': 'rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM in separate runs) of '
                           'bench.py --steps 2 --warmup 1 --graph 0; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 '
                           'reports half the bytes of wide streaming reads), write bytes = WRITE_SIZE x 1024; '
                           'per launch = summed over the dispatches of the kernel / their number',
                   'kernels': out,
                   'hbm_bytes_per_step': (round(sum(e['hbm_read_bytes'] + e['hbm_write_bytes'] for e in table.values())
                                                / steps[0]) if steps else None),
                   'steps_profiled': steps[0] if steps else None},
                  open(json_out[0], 'w'), indent=1, sort_keys=True)
    print('| kernel | dispatches | MFMA busy / (active cycles x SIMDs) | f32 MFMA MOPS per dispatch | HBM read B/dispatch (2xFETCH_SIZE) | HBM write B/dispatch |')
    print('|---|---|---|---|---|---|')
    for gui, k, n, util, mops, rd, wr in rows[:40]:
        f = lambda v: '-' if v is None else f'{v:,.0f}'
        print(f'| `{k[:90]}` | {n} | {util:.3f} | {mops / n:,.0f} | {f(rd)} | {f(wr)} |')


if __name__ == '__main__':
    main()
