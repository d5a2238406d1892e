#!/bin/bash
# what a scan's throughput time is made of: the headline configuration with one part knocked out at a time (GPU box)
run() { python bench.py $1 --no-cpu --no-tracker --sequential-scans 0 --repeats 3 --profile-steps 0 > gpurun_out/k.json 2> gpurun_out/k.err || tail -3 gpurun_out/k.err; python -c "
import json
d=json.load(open('gpurun_out/k.json')); v=d['value_windows']['scans_per_sec']['all']; print('%-40s' % '$1', v, 'us/scan %.1f' % (1e6/ (sum(v)/len(v))))"; }
run ""
run "--frozen-map"
run "--icp-iters 5"
run "--icp-iters 1"
run "--frozen-map --icp-iters 1"
run "--map-source assemble"
run "--batch 4 --inflight 4"
run "--batch 1 --inflight 4"
run "--batch 1 --inflight 1"
run "--batch 8 --inflight 2"
run "--no-share-map"
