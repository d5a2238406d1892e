# throughput of the headline configuration under a few environment settings: bash tools/diag/r03_env_sweep.sh
run() { python bench.py --no-cpu --no-tracker --sequential-scans 0 --profile-steps 0 --repeats 2 --cached-plan-steps 0 > gpurun_out/sweep.json 2> gpurun_out/sweep.err; python -c "
import json
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('$1', d['value'], d['value_windows']['scans_per_sec']['all'], d['results_ok'])"; }
run base
LVI_ICP_WIDE_FROM=2 run wide_from_2
LVI_ICP_WIDE_FROM=3 run wide_from_3
LVI_ICP_G1=4 run g1_4
LVI_ICP_G0=8 run g0_8
LVI_ICP_G0=2 run g0_2
LVI_ICP_G1=4 LVI_ICP_WIDE_FROM=2 run g1_4_wf2
