#!/bin/bash
# Round 6: collects the evidence kept under profiles/r06_* on the GPU box (run through gpurun; tools/keep_r06.sh copies the summaries).
# rocprofv3 wraps python3 directly (no env / bash hop).  usage: tools/collect_r06.sh [A|B|C|D]
set -u
PART=${1:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
rows=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
open(sys.argv[1],'w').write(rows[-1]+'\n' if rows else '')" "$1"; }
if want A; then
# 1. the bench line as the driver runs it, and with its default window
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
GSX_SPEC_DEBUG=2 python3 bench.py > $OUT/bench.json 2> $OUT/bench_frames.err   # (per-frame mode lines on stderr: which frames the tuner ran unspeculated)
# 2. kernel trace + stats of the same command: k_project<3,0,0> (the full-projection loop) and k_project_geom averages must agree with
#    roofline.avg_launch_us / roofline_speculated.avg_launch_us of the line the traced run prints itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline --no-pmc --no-extra-legs --no-robustness --no-cfg5 > $OUT/kt_bench.log 2>&1
# 3. per-frame kernel breakdowns: the speculated loop, the unspeculated loop (slab shading), the fully projecting loop, the sharded frame at world 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_spec -o kt -- python3 bench.py --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --no-cfg5 --render-options speculative=1 > $OUT/kt_spec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_nospec -o kt -- python3 bench.py --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --no-cfg5 --render-options speculative=0 > $OUT/kt_nospec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_full -o kt -- python3 bench.py --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --no-cfg5 --render-options speculative=0,slab_shading=0 > $OUT/kt_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_dist -o kt -- python3 bench.py --force-dist --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --dist-frames-in-flight 1 > $OUT/kt_dist.log 2>&1
for k in spec nospec full dist; do python3 tools/kernel_breakdown.py $OUT/kt_$k 143 40 > $OUT/kt_${k}_breakdown.txt; done
python3 tools/kernel_gaps.py $OUT/kt_spec 60 k_project timeline 1 > $OUT/kt_spec_timeline.txt 2>&1
python3 tools/kernel_gaps.py $OUT/kt_nospec 40 k_project timeline 1 > $OUT/kt_nospec_timeline.txt 2>&1
# 4. HBM traffic, separate PMC passes (what bench.py's child passes do, kept here as raw per-kernel averages)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --pmc-child --steps 4 --warmup 3 > $OUT/pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_summary.csv
fi
if want B; then
# 5. the other workloads / pods / the index-sharded path on one rank (frames in flight 1, 2, 3), lanes on the single-GPU path
for w in cfg2 cfg3; do python3 bench.py --workload $w --no-cpu-baseline --no-pmc --no-cfg5 > $OUT/bench_$w.json 2>> $OUT/bench.err; done
python3 bench.py --pod half/half --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs --no-cfg5 > $OUT/bench_half_half.json 2>> $OUT/bench.err
python3 bench.py --pod norm8/half --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs --no-cfg5 > $OUT/bench_norm8_half.json 2>> $OUT/bench.err
for L in 1 2 3; do python3 bench.py --force-dist --no-cpu-baseline --no-pmc --dist-frames-in-flight $L > $OUT/bench_index_world1_inflight$L.json 2>> $OUT/bench.err; line $OUT/bench_index_world1_inflight$L.json; done
python3 bench.py --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs --no-cfg5 --frames-in-flight 1 > $OUT/bench_inflight1.json 2>> $OUT/bench.err
python3 bench.py --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs --no-cfg5 --frames-in-flight 3 > $OUT/bench_inflight3.json 2>> $OUT/bench.err
# 6. BASELINE configs[4] on one GPU, long form: with and without the stored selection + edit, and through the sharded call at world 1
(python3 tools/bench_cfg5.py; python3 tools/bench_cfg5.py --edit 0; python3 tools/bench_cfg5.py --edit 0 --shard 1) 2>> $OUT/bench.err | grep '^{' > $OUT/bench_cfg5.json
tools/bench_hbm > $OUT/bench_hbm.txt 2>&1
for a in "8460000 32 depth" "3000000 32 depth" "1000000 32 depth" "310000 32 depth" "40000 32 depth" "100000 32 dup" "870000 8"; do echo "== $a"; tools/bench_sort $a 2>&1 | grep -E "us per sort|bucket sort|mismatch"; done > $OUT/bench_sort.txt 2>&1
python3 tools/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err
tools/ab_robust.sh "GSX_DEFAULT=1" > $OUT/robustness.txt 2>&1
fi
if want C; then
# 7. the bench as the driver launches it with N ranks on ONE GPU (RCCL between processes over sockets: the N > 1 code path, the rccl field)
for N in 2 8; do
  GSX_BENCH_ONE_DEVICE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_ranks_on_one_gpu_world$N.json 2>> $OUT/bench.err
  line $OUT/bench_ranks_on_one_gpu_world$N.json
done
# 8. same-box A/Bs of this round's pieces on the bench line
BENCH_ARGS="--no-cfg5" tools/ab_env.sh GSX_BUCKET_SORT=0 GSX_BUCKET_SORT=1 GSX_BIN_FUSED=0 GSX_BIN_FUSED=1 GSX_SLAB_SHADING=0 GSX_SLAB_SHADING=1 "GSX_BUCKET_SORT=0 GSX_BIN_FUSED=0 GSX_SLAB_SHADING=0" "GSX_ALL=1" > $OUT/ab_round6.txt 2>&1
fi
if want D; then
# 9. every rank of an N-rank frame alone on the GPU over the native replay transport: cfg4 (worlds 2, 4, 8; orbit and open sky) and cfg5
python3 tools/rank_alone.py --frames 60 --out $OUT/rank_alone.json > /dev/null 2> $OUT/rank_alone.err
python3 tools/rank_alone.py --workload cfg5 --worlds 2,4,8 --scenes orbit --frames 50 --out $OUT/rank_alone_cfg5.json > /dev/null 2>> $OUT/rank_alone.err
python3 tools/rank_table.py $OUT/rank_alone.json > $OUT/rank_table.txt 2>&1
python3 tools/rank_table.py $OUT/rank_alone_cfg5.json > $OUT/rank_table_cfg5.txt 2>&1
fi
ls $OUT | head -80
