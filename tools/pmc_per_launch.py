"""Per-LAUNCH HBM traffic and L2 hit rate of one kernel from the passes of tools/gpu_pmc.sh, launches grouped by their position inside
the step (which launches of conv1d_ring_persistent_kernel carry the read over-fetch -- the six K = 1024 layers or the
K = 3072 MFA layer).  usage: python tools/pmc_per_launch.py gpurun_out/<tag> <kernel substring> <launches per step>"""
import csv, glob, os, sys, collections
root, sub, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = collections.defaultdict(lambda: collections.defaultdict(list))  # counter -> slot -> values
for f in sorted(glob.glob(os.path.join(root, '*', 'pmc_counter_collection.csv'))):
    seq = collections.defaultdict(int)
    for row in sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id'])):
        if sub not in row['Kernel_Name']:
            continue
        c = row['Counter_Name']
        rows[c][seq[c] % per].append(float(row['Counter_Value']))
        seq[c] += 1
print(f'{sub}: launches grouped by position in the step ({per} per step); FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950')
for slot in range(per):
    f = rows['FETCH_SIZE'][slot]
    w = rows['WRITE_SIZE'][slot]
    h, m = rows['TCC_HIT_sum'][slot], rows['TCC_MISS_sum'][slot]
    if not f:
        continue
    fm, wm = sum(f) / len(f), sum(w) / len(w) if w else 0.0
    hit = sum(h) / (sum(h) + sum(m)) if h and (sum(h) + sum(m)) > 0 else float('nan')
    print(f'  launch {slot}: reads {2 * fm / 1e3:8.1f} MB  writes {wm / 1e3:8.1f} MB  total {(2 * fm + wm) / 1e3:8.1f} MB   L2 hit rate {hit:.3f}   (n = {len(f)})')
