#!/bin/bash
# Profiles of the default bench command for profiles/<tag>_*: run ON THE GPU BOX from the repo root:
#   bash scripts/profile_round.sh r02 [workload]
# (1) rocprofv3 --kernel-trace --stats of `bench.py` (the timed command itself);
# (2-4) separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy): counters are never combined with the trace domains
#       other than --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC slots; gpurun rules).
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -e
TAG=${1:-r02}
WL=${2:-resnext50_full_b8_1024}
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-f32x3 --no-other-configs"
# (1a) kernels ALONE on the GPU (auxiliary streams off), like bench.py's instrumented roofline step: these average
#      durations are the ones `roofline.avg_launch_us` must agree with
export MASKLAB_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format rocpd -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1
python3 scripts/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/kernel_trace_summary.md > /dev/null
unset MASKLAB_SIDE_STREAM
# (1b) the headline command as it is timed: towers and the semantic head on auxiliary streams -- kernels overlap, so a
#      kernel's duration here includes the time it shares the chip with another
rocprofv3 --kernel-trace --stats --output-format rocpd -d $OUT/trace_c -o trace -- $BENCH > $OUT/bench_trace_concurrent.log 2>&1
python3 scripts/rocpd_summary.py $(find $OUT/trace_c -name "*.db" | head -1) $OUT/kernel_trace_concurrent_summary.md > /dev/null
python3 scripts/rocpd_timeline.py $(find $OUT/trace_c -name "*.db" | head -1) $OUT/timeline_concurrent.md > /dev/null
echo "trace done"
export MASKLAB_SIDE_STREAM=0
for pass in fetch:FETCH_SIZE write:WRITE_SIZE mfma:SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --kernel-trace --pmc ${ctr//,/ } --output-format rocpd -d $OUT/pmc_$name -o $name -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32x3 --no-other-configs > $OUT/bench_pmc_$name.log 2>&1
  python3 scripts/rocpd_pmc_summary.py $(find $OUT/pmc_$name -name "*.db" | head -1) $OUT/pmc_$name.md > /dev/null
  echo "pmc $name done"
done
python3 scripts/pmc_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) $OUT/traffic.json $WL
grep '^{' $OUT/bench_trace.log | tail -1 > $OUT/bench_line.json
rm -rf $OUT/trace $OUT/trace_c $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma     # the SQLite results are large; the summaries are what travels back
