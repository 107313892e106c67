"""HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc.sh -> profiles/pmc_traffic.json
(read by bench.py for roofline.traffic).   python tools/pmc_traffic.py gpurun_out/<tag> <round-tag>

Units and corrections as prescribed by MI355X_MICROARCH.md (HBM section): the counters are in KiB-like units of 1000
bytes as printed by rocprofv3 (kilobytes); on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so it
is doubled; WRITE_SIZE is taken as is (it matches the known byte count of the Fbank kernel: 2 x 24.4 MB)."""
import csv, glob, json, os, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('fetch', 'write'):
    for f in glob.glob(os.path.join(root, sub, 'pmc_counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
out = {'source': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {tag}; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1000 / launches',
       'kernels': {}}
for name, c in acc.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        f = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE'])
        w = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
        out['kernels'][name] = {'fetch_raw_kb': round(f, 1), 'write_kb': round(w, 1), 'hbm_bytes_per_launch': int((2 * f + w) * 1000),
                                'launches_sampled': len(c['FETCH_SIZE'])}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
json.dump(out, open(dst, 'w'), indent=1, sort_keys=True)
print(dst, list(out['kernels']))
