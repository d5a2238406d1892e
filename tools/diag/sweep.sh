#!/bin/bash
# env-knob sweep on the headline configuration (GPU box): bash tools/diag/sweep.sh "VAR=a" "VAR=b" ...
for kv in "$@"; do
  env $kv python bench.py --no-cpu --no-tracker --sequential-scans 0 --repeats 3 --profile-steps 0 > gpurun_out/sw.json 2> gpurun_out/sw.err || tail -3 gpurun_out/sw.err
  python -c "
import json
d=json.load(open('gpurun_out/sw.json')); print('%-28s' % '$kv', d['value_windows']['scans_per_sec']['all'], d['results_ok'])"
done
