#!/bin/bash
# HBM-side bytes of the residual kernel per launch (one scan in flight), FETCH_SIZE pass only: bash tools/diag/pmc_residual.sh [ENV=val ...]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/pmc_res; rm -rf $O; mkdir -p $O
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --inflight 1 --batch 1 > $O/pf.json 2> $O/pf.err
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_res/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "icp_residual" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Grid_Size"]].append(float(r["Counter_Value"]))
for g, v in agg.items():
    v = v[-20:]
    print("icp_residual grid", g, "launches", len(v), "read MB per launch (2 x FETCH_SIZE KiB): mean %.2f min %.2f max %.2f" % (2 * 1024 * sum(v) / len(v) / 1e6, 2 * 1024 * min(v) / 1e6, 2 * 1024 * max(v) / 1e6))
PY
rm -rf $O/pf
