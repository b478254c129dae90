"""Per-kernel summary of a rocprofv3 kernel trace (rocpd sqlite output): calls, total / average
duration and share of GPU time, shortest name that keeps the template arguments apart.

usage: python tools/kernel_stats.py <results.db> [steps] [--csv out.csv]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*\)$', '', name)
    name = name.replace('void ', '')
    name = re.sub(r'Tile<(\d+), (\d+), \d+, \d+, (true|false), (true|false)>', r'T\1x\2', name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
    rows = db.execute('select name, count(*), sum(end - start), min(end - start), max(end - start) '
                      'from kernels group by name order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    out = ['kernel,calls,calls_per_step,total_ms,avg_us,min_us,max_us,percent,ms_per_step']
    for name, n, tot, mn, mx in rows:
        out.append('"%s",%d,%.1f,%.3f,%.2f,%.2f,%.2f,%.2f,%.4f' % (short(name), n, n / steps, tot / 1e6, tot / n / 1e3,
                                                                  mn / 1e3, mx / 1e3, 100.0 * tot / total,
                                                                  tot / 1e6 / steps))
    out.append('"TOTAL",%d,%.1f,%.3f,,,,100,%.4f' % (sum(r[1] for r in rows), sum(r[1] for r in rows) / steps,
                                                    total / 1e6, total / 1e6 / steps))
    text = '\n'.join(out)
    if '--csv' in sys.argv:
        open(sys.argv[sys.argv.index('--csv') + 1], 'w').write(text + '\n')
    else:
        print(text)


if __name__ == '__main__':
    main()
