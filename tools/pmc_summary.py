"""Per-kernel averages of the rocprofv3 --pmc passes written by tools/gpu_pmc.sh: python tools/pmc_summary.py gpurun_out/<tag>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
table = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, '*', 'pmc_counter_collection.csv'))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
            table[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name, ctr in sorted(table.items()):
    print(name)
    for c, v in sorted(ctr.items()):
        print(f'    {c:36s} n={len(v):3d} mean={sum(v) / len(v):.4g}')
