#!/bin/bash
# diagnostic: headline only (4 handles x 8 scans, faithful map plan), no cpu / tracker / sequential legs; prints value and windows
O=gpurun_out; mkdir -p $O
python3 bench.py --no-cpu --no-tracker --sequential-scans 0 --cached-plan-steps 0 --profile-steps 0 "$@" > $O/qh.json 2> $O/qh.err || { tail -5 $O/qh.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/qh.json').read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "windows", d.get("windows") or d.get("config",{}).get("windows"))
PY
