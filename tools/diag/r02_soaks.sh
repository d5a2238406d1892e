#!/bin/bash
# the long forms of the parity soaks (GPU box); summaries under gpurun_out/soak_r02/
O=gpurun_out/soak_r02; mkdir -p $O
python tests/soak_parity.py 200 2024 > $O/lidar.log 2>&1; tail -1 $O/lidar.log > $O/r02_parity_soak.json; echo "lidar rc=$?"
python tests/soak_tracker.py 400 2025 > $O/tracker.log 2>&1; tail -1 $O/tracker.log > $O/r02_parity_soak_tracker.json; echo tracker done
python tests/soak_voxel_knn.py 300 31 > $O/voxel.log 2>&1; tail -1 $O/voxel.log > $O/r02_parity_soak_voxel_knn.json; echo voxel done
python tests/soak_frows.py 100 5 > $O/frows.log 2>&1; tail -1 $O/frows.log > $O/r02_parity_soak_frows.json; echo frows done
python - <<'PY'
import json
for f in ("r02_parity_soak","r02_parity_soak_tracker","r02_parity_soak_voxel_knn","r02_parity_soak_frows"):
    try:
        d=json.load(open("gpurun_out/soak_r02/%s.json" % f)); print(f, {k:v for k,v in d.items() if k!="report"}, len(d.get("report",[])))
    except Exception as e: print(f, "ERR", e)
PY
