cd /tmp && export TMPDIR=/tmp && cd - >/dev/null; O=gpurun_out/prof_r03; mkdir -p $O
COMMON="--steps 12 --warmup 3 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --cached-plan-steps 0 --inflight 4 --batch 8"
for CFG in "LVI_ICP_WIDE_FROM=1" "LVI_ICP_WIDE_FROM=3" "LVI_ICP_WIDE_FROM=99" "LVI_ICP_WIDE_FROM=99 LVI_ICP_G1=4"; do
  env $CFG true
  ( export $CFG; rocprofv3 --kernel-trace --output-format csv -d $O/sw -- python3 bench.py $COMMON > $O/sw.json 2> $O/sw.err )
  python3 tools/summarize_prof.py steady $O/sw $O/sw.md 12 > /dev/null
  echo "== $CFG"; grep "icp_gn\|steady state" $O/sw.md; rm -rf $O/sw
done
