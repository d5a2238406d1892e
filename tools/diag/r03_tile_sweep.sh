cd /tmp && export TMPDIR=/tmp && cd - >/dev/null; O=gpurun_out/prof_r03; mkdir -p $O
for NT in 0 1; do
LVI_KNN_NO_TILES=$NT rocprofv3 --kernel-trace --output-format csv -d $O/b1 -- python3 bench.py --steps 12 --warmup 3 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --cached-plan-steps 0 --inflight 1 --batch 1 > $O/b1.json 2> $O/b1.err; python3 tools/diag/gn_trace.py $O/b1 > $O/b1_trace_nt$NT.txt; rm -rf $O/b1; echo NO_TILES=$NT; grep "icp_\|feat_order\|step wall" $O/b1_trace_nt$NT.txt
LVI_KNN_NO_TILES=$NT python bench.py --no-cpu --no-tracker --sequential-scans 0 --profile-steps 0 --repeats 3 --cached-plan-steps 0 > gpurun_out/r03_b7_$NT.json 2> gpurun_out/r03_b7.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_b7_$NT.json').read().strip().splitlines()[-1])
print('NO_TILES=$NT', d['value'], d['value_windows']['scans_per_sec']['all'], d['results_ok'])"
done
