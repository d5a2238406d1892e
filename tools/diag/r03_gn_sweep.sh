cd /tmp && export TMPDIR=/tmp && cd - >/dev/null; O=gpurun_out/prof_r03; mkdir -p $O
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t4.log 2>&1; echo rc=$? >> gpurun_out/r03_t4.log; tail -3 gpurun_out/r03_t4.log
for WF in 1 3 99; do
LVI_ICP_WIDE_FROM=$WF rocprofv3 --kernel-trace --output-format csv -d $O/b1 -- python3 bench.py --steps 12 --warmup 3 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --inflight 1 --batch 1 > $O/b1.json 2> $O/b1.err; python3 tools/diag/gn_trace.py $O/b1 > $O/b1_trace_$WF.txt; rm -rf $O/b1; echo WF=$WF; grep "icp_\|step wall" $O/b1_trace_$WF.txt
LVI_ICP_WIDE_FROM=$WF python bench.py --no-cpu --no-tracker --sequential-scans 0 --profile-steps 0 --repeats 3 > gpurun_out/r03_b4_$WF.json 2> gpurun_out/r03_b4.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_b4_$WF.json').read().strip().splitlines()[-1])
print('WF=$WF', d['value'], d['value_windows']['scans_per_sec']['all'], d['results_ok'])"
done
