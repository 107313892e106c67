"""Per-kernel statistics from a rocprofv3 rocpd (.db) kernel trace -> markdown/CSV summary (kept under profiles/)."""
import sqlite3
import sys


def main(db, out=None, skip_first=0):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        short = name.split('(')[0]
        d = (e - s) / 1e3
        st = stats.setdefault(short, [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    lines = ['kernel,calls,total_us,avg_us,min_us,max_us,percent']
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{v[0]},{v[1]:.1f},{v[1] / v[0]:.2f},{v[2]:.2f},{v[3]:.2f},{100 * v[1] / total:.2f}')
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
