#!/bin/bash
# the long forms of the parity soaks (GPU box); summaries under gpurun_out/soak_r03/
O=gpurun_out/soak_r03; mkdir -p $O
python tests/soak_parity.py 300 3024 > $O/lidar.log 2>&1; tail -1 $O/lidar.log > $O/r03_parity_soak.json; echo "lidar rc=$?"
python tests/soak_tracker.py 600 3025 > $O/tracker.log 2>&1; tail -1 $O/tracker.log > $O/r03_parity_soak_tracker.json; echo tracker done
python tests/soak_voxel_knn.py 400 331 > $O/voxel.log 2>&1; tail -1 $O/voxel.log > $O/r03_parity_soak_voxel_knn.json; echo voxel done
python tests/soak_frows.py 150 35 > $O/frows.log 2>&1; tail -1 $O/frows.log > $O/r03_parity_soak_frows.json; echo frows done
python - <<'PY'
import json
for f in ("r03_parity_soak","r03_parity_soak_tracker","r03_parity_soak_voxel_knn","r03_parity_soak_frows"):
    try:
        d=json.load(open("gpurun_out/soak_r03/%s.json" % f)); print(f, {k:v for k,v in d.items() if k!="report"}, len(d.get("report",[])))
    except Exception as e: print(f, "ERR", e)
PY
