"""Per-variant, per-kernel average durations out of ONE rocprofv3 --kernel-trace run of tools/bench_walk.py: the trace is cut into forwards
at the front-end launches (one fbank launch per forward), forwards into variants by bench_walk's fixed schedule (3 warm-up + `steps` timed
forwards per variant and round).  usage: python tools/walk_trace_summary.py <dir with *_kernel_trace.csv> <steps> <rounds>"""
import csv, glob, os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
root, steps, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_walk.py')).read()
VARIANTS = eval(re.search(r'VARIANTS = (\[.*?\])\n\n', src, re.S).group(1))
f = sorted(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
per = 3 + steps
fwd = -1
acc = collections.defaultdict(lambda: collections.defaultdict(list))   # variant -> kernel -> per-forward total ns
cur = collections.defaultdict(float)
def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('mv::', '')
    m = re.match(r'_ZN2mv\d+([a-z0-9_]+)', n)
    return m.group(1) if m else n
def flush():
    if fwd < 0:
        return
    g = fwd // per
    if g >= len(VARIANTS) * rounds or fwd % per < 3:
        return
    v = VARIANTS[g % len(VARIANTS)]
    for k, t in cur.items():
        acc[v][k].append(t)
for r in rows:
    name = short(r['Kernel_Name'])
    if 'fbank' in name:
        flush()
        cur = collections.defaultdict(float)
        fwd += 1
    if fwd >= 0:
        cur[name] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
flush()
kernels = sorted({k for v in acc for k in acc[v]}, key=lambda k: -sum(acc[VARIANTS[0]].get(k, [0])))
base = {k: sum(acc[VARIANTS[0]][k]) / max(1, len(acc[VARIANTS[0]][k])) for k in kernels}
print('per-forward kernel time in us (sum over the launches of a forward), difference to the first variant in brackets')
for v in VARIANTS:
    tot = 0.0
    parts = []
    for k in kernels:
        xs = acc[v].get(k, [])
        if not xs:
            continue
        m = sum(xs) / len(xs)
        tot += m
        if base[k] > 15e3 or abs(m - base[k]) > 3e3:
            parts.append(f'{k[:28]} {m / 1e3:.0f} [{(m - base[k]) / 1e3:+.0f}]')
    print(f'MV_WALK={v[0]:3d} chunks={v[1]}  total {tot / 1e3:7.1f} us | ' + ' | '.join(parts))
