#!/bin/bash
# Collects the per-round evidence under gpurun_out/<round>/ on the GPU box (run through gpurun); copy the summaries into
# profiles/ afterwards (tools/collect_profiles.sh r01 && cp ...).  rocprofv3 wraps python3 directly (no env / bash hop).
set -u
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. kernel trace + stats of the default bench command (speculated steady state) and of the unspeculated path
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/kt_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_nospec -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --render-options speculative=0 > $OUT/kt_nospec_bench.log 2>&1
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one), kernel trace only
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_summary.csv $OUT/pmc_traffic.json cfg4:1:k_project
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
# 3. the bench lines
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --no-cpu-baseline --pass-timing all > $OUT/bench_all_passes.json 2>> $OUT/bench.err
python3 bench.py --no-cpu-baseline --render-options speculative=0 --pass-timing all > $OUT/bench_nospec.json 2>> $OUT/bench.err
for w in cfg2 cfg3; do python3 bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench.err; done
python3 bench.py --pod half/half --no-cpu-baseline > $OUT/bench_half_half.json 2>> $OUT/bench.err
python3 bench.py --pod norm8/half --no-cpu-baseline > $OUT/bench_norm8_half.json 2>> $OUT/bench.err
python3 tools/kernel_breakdown.py $OUT/kt 174 > $OUT/kt_breakdown.txt
python3 tools/kernel_breakdown.py $OUT/kt_nospec 174 > $OUT/kt_nospec_breakdown.txt
tail -n +1 $OUT/bench.json | cut -c1-400
