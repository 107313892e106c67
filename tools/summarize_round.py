"""Summarise one tools/gpu_round.sh output directory: bench lines + per-kernel times of the last profiled step."""
import json
import os
import sqlite3
import sys

d = sys.argv[1]
for f in sorted(os.listdir(d)):
    if f.startswith('bench') and f.endswith('.log'):
        for l in open(os.path.join(d, f)):
            if l.startswith('{'):
                j = json.loads(l)
                print(f'{f}: {j["value"]} utt/s, {j["ms_per_step"]} ms/step, stages {j["stage_ms"]}, fbank frac {j["roofline"]["frac"]}, '
                      f'backbone TF {j["roofline_backbone"]["achieved"]}, parity {j.get("parity", {}).get("max_one_minus_cos")}, cpu {j.get("cpu_baseline", {}).get("value")}')
for f in sorted(os.listdir(d)):
    if f.endswith('.log') and 'pytest' in f:
        for l in open(os.path.join(d, f)):
            if ' passed' in l or ' failed' in l or l.startswith('FAILED') or l.startswith('ERROR'):
                print(f, l.strip()[:300])
for sub in sorted(os.listdir(d)):
    db = os.path.join(d, sub, 'bench_results.db')
    if not os.path.exists(db):
        continue
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, start, end, grid_x, workgroup_x from kernels order by start').fetchall()
    first = [i for i, r in enumerate(rows) if 'fbank' in r[0] or 'stft' in r[0]]
    if not first:
        continue
    seen = {}
    for r in rows[first[-1]:]:
        nm = r[0].split('(')[0][-30:]
        seen.setdefault((nm, r[3] // r[4]), []).append((r[2] - r[1]) / 1e3)
    print(f'--- {sub}: kernels of the last step')
    tot = 0
    for k, v in seen.items():
        print(f'{k[0]:32s} grid {k[1]:7d} n={len(v):3d} avg {sum(v) / len(v):8.1f} us total {sum(v):8.1f}')
        tot += sum(v)
    print(f'sum {tot:.1f} us')
