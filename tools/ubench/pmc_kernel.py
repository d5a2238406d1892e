"""per-launch PMC counters of one kernel: `python tools/ubench/pmc_kernel.py <rocprofv3 output dir> <kernel substring>`
(prints, per launch geometry, the mean of every counter found in the run's *_counter_collection.csv)"""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        key = (r["Grid_Size"], r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"]); acc[key][1].add(r["Dispatch_Id"])
for (grid, name), (v, ids) in sorted(acc.items()):
    print(f"grid {grid:>9} {name:44s} {v / max(len(ids), 1):16.1f}  ({len(ids)} launches)")
