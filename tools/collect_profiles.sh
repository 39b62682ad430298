#!/bin/bash
# Collects the per-round evidence under gpurun_out/<round>/ on the GPU box (run through gpurun); copy the summaries into
# profiles/ afterwards.  rocprofv3 wraps python3 directly (no env / bash hop).  usage: tools/collect_profiles.sh r05 [A|B|C|D]
set -u
R=${1:-r05}
PART=${2:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if want A; then
# 1. the bench line exactly as the driver runs it (PMC child passes, every timed loop, frame check, synchronised / steady / per-pass
#    legs, robustness legs, full CPU baseline), and with the driver's short window
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
# 2. kernel trace + stats of the same command (the timed loops are in it: k_project<3,0,0> and k_project_geom averages must agree
#    with roofline.avg_launch_us / roofline_speculated.avg_launch_us of the line it prints itself)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline --no-pmc --no-extra-legs --no-robustness > $OUT/kt_bench.log 2>&1
# 3. per-frame kernel breakdowns of the speculated and of the unspeculated loop alone, and of the sharded frame at world 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_spec -o kt -- python3 bench.py --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --render-options speculative=1 > $OUT/kt_spec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_nospec -o kt -- python3 bench.py --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --render-options speculative=0 > $OUT/kt_nospec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_dist -o kt -- python3 bench.py --force-dist --steps 100 --warmup 10 --min-steps 0 --no-cpu-baseline --no-pmc --dist-frames-in-flight 1 > $OUT/kt_dist.log 2>&1
python3 tools/kernel_breakdown.py $OUT/kt_spec 143 40 > $OUT/kt_spec_breakdown.txt
python3 tools/kernel_breakdown.py $OUT/kt_nospec 143 40 > $OUT/kt_nospec_breakdown.txt
python3 tools/kernel_breakdown.py $OUT/kt_dist 143 40 > $OUT/kt_dist_breakdown.txt
# 4. HBM traffic, separate PMC passes (what bench.py's child passes do, kept here as raw per-kernel averages)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --pmc-child --steps 4 --warmup 3 > $OUT/pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_summary.csv
fi
if want B; then
# 5. the other workloads / pods / the index-sharded path on one rank (frames in flight 1, 2, 3)
for w in cfg2 cfg3; do python3 bench.py --workload $w --no-cpu-baseline --no-pmc > $OUT/bench_$w.json 2>> $OUT/bench.err; done
python3 bench.py --pod half/half --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs > $OUT/bench_half_half.json 2>> $OUT/bench.err
python3 bench.py --pod norm8/half --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs > $OUT/bench_norm8_half.json 2>> $OUT/bench.err
for L in 1 2 3; do python3 bench.py --force-dist --no-cpu-baseline --no-pmc --dist-frames-in-flight $L > $OUT/bench_index_world1_inflight$L.json 2>> $OUT/bench.err; done
# 6. frames in flight on the single-GPU path: one / three lanes for the headline loop
python3 bench.py --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs --frames-in-flight 1 > $OUT/bench_inflight1.json 2>> $OUT/bench.err
python3 bench.py --no-cpu-baseline --no-pmc --no-robustness --no-extra-legs --frames-in-flight 3 > $OUT/bench_inflight3.json 2>> $OUT/bench.err
# 7. BASELINE configs[4] on one GPU: 4 x 6 M Gaussians at 3840x2160, with and without the stored selection + edit
(python3 tools/bench_cfg5.py; python3 tools/bench_cfg5.py --edit 0; python3 tools/bench_cfg5.py --edit 0 --shard 1) 2>> $OUT/bench.err | grep '^{' > $OUT/bench_cfg5.json
tools/bench_hbm > $OUT/bench_hbm.txt 2>&1
tools/bench_sort 8460000 32 depth > $OUT/bench_sort.txt 2>&1
tools/bench_sort 310000 32 depth >> $OUT/bench_sort.txt 2>&1
tools/bench_sort 870000 8 >> $OUT/bench_sort.txt 2>&1
python3 tools/shard_host_time.py > $OUT/shard_host_time.txt 2>&1
python3 tools/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err
fi
if want C; then
# 8. round 4: what a kernel boundary costs on a stream / in a graph, what the launch traces do to a frame, every rank of an
#    N-rank frame alone on the GPU (the predicted 2 / 4 / 8-GPU rates), the bench as the driver launches it with N ranks on one GPU
tools/bench_launch > $OUT/bench_launch.txt 2>&1
python3 tools/graph_probe.py 200 > $OUT/graph_probe.txt 2>> $OUT/bench.err
for N in 2 8; do
  GSX_BENCH_ONE_DEVICE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_ranks_on_one_gpu_world$N.json 2>> $OUT/bench.err
done

# 9. round 4: per-tile profile of the block compositor (index order against expensive-first), the A/B of that order on the bench
#    line, what atomics on one address cost
python3 tools/tile_profile.py > $OUT/tile_profile.txt 2>> $OUT/bench.err
GSX_TILE_ORDER=0 python3 tools/tile_profile.py > $OUT/tile_profile_index_order.txt 2>> $OUT/bench.err
tools/ab_env.sh GSX_TILE_ORDER=0 GSX_TILE_ORDER=1 GSX_TILE_ORDER=0 GSX_TILE_ORDER=1 > $OUT/ab_tile_order.txt 2>&1
tools/bench_atomic > $OUT/bench_atomic.txt 2>&1
python3 tools/rank_alone.py --worlds 8 --scenes open_sky --lanes 1 --balance 0 --frames 60 --out $OUT/rank_alone_equal_bands.json > /dev/null 2>> $OUT/rank_alone.err
fi
if want D; then
# 10. round 5: every rank of an N-rank frame alone on the GPU over the native replay transport, cfg4 and cfg5; the sharded frame and
#     cfg5 against the round-4 protocol (build_variants/libgsx_base.so) on this box; the block size A/B; the radix tile size A/B
python3 tools/rank_alone.py --frames 60 --out $OUT/rank_alone.json > /dev/null 2> $OUT/rank_alone.err
python3 tools/rank_alone.py --workload cfg5 --worlds 2,4,8 --scenes orbit --frames 50 --out $OUT/rank_alone_cfg5.json > /dev/null 2>> $OUT/rank_alone.err
python3 tools/rank_alone.py --worlds 8 --scenes orbit --lanes 1 --speculate 1 --frames 60 --python-replay --out $OUT/rank_alone_python_replay.json > /dev/null 2>> $OUT/rank_alone.err
[ -f build_variants/libgsx_base.so ] && tools/ab_shard.sh base new > $OUT/ab_shard.txt 2>&1
python3 tools/ab_blocks.py > $OUT/ab_blocks.txt 2>> $OUT/bench.err
# 11. layered frames model by model (frames in flight) against all models at once with device-decided repairs: cfg5 at world 1 and every rank of a
#     world-8 cfg5 frame alone; thousands of layered frames, every one looked at one call late
(for P in 0 1; do echo "== GSX_SHARD_LAYER_PIPELINE=$P: cfg5 through gsx_shard_render_frame_keys at world 1"; GSX_SHARD_LAYER_PIPELINE=$P python3 tools/bench_cfg5.py --edit 0 --shard 1 2>/dev/null | tail -1
 echo "== GSX_SHARD_LAYER_PIPELINE=$P: every rank of a world-8 cfg5 frame alone, two frames in flight"; GSX_SHARD_LAYER_PIPELINE=$P python3 tools/rank_alone.py --workload cfg5 --worlds 8 --scenes orbit --lanes 2 --speculate 1 --frames 50 --no-pass-replay 2>&1 >/dev/null | grep predicted; done) > $OUT/ab_layer_pipeline.txt 2>&1
python3 tools/long_run_layers.py 8000 2>&1 | grep "world\|long run\|Error" > $OUT/long_run_layers.txt
for n in 100000 310000 870000; do
  echo "== n=$n, 2048-element tiles off" >> $OUT/ab_radix_small.txt; tools/bench_sort $n 32 depth >> $OUT/ab_radix_small.txt 2>&1
  echo "== n=$n, default" >> $OUT/ab_radix_small.txt; tools/bench_sort $n 32 depth >> $OUT/ab_radix_small.txt 2>&1
done
fi
for f in $OUT/bench.json $OUT/bench_driver_args.json $OUT/bench_index_world1_inflight2.json; do [ -s $f ] && cut -c1-300 $f; done
