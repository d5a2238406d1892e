run() { python bench.py --no-cpu --no-tracker --sequential-scans 0 --profile-steps 0 --repeats 2 --cached-plan-steps 0 > gpurun_out/sweep.json 2> gpurun_out/sweep.err; python -c "
import json
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('$1', d['value'], d['value_windows']['scans_per_sec']['all'], d['results_ok'])"; }
run base
LVI_ICP_G1=4 run g1_4
