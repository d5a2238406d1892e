#!/bin/bash
run() { env $1 python bench.py $2 --no-cpu --no-tracker --steps 20 --repeats 3 --profile-steps 0 --sequential-scans 0 > /tmp/sw.json 2>/tmp/sw.err || tail -3 /tmp/sw.err; python -c "
import json
d=json.load(open('/tmp/sw.json')); print('$1 | $2 ->', d['value_windows']['scans_per_sec']['all'], d['results_ok'])"; }
run "X=1" "--batch 4 --inflight 4"
run "LVI_EXPERIMENT_EXTRA_LAUNCH=1" "--batch 4 --inflight 4"
