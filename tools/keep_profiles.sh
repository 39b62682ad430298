#!/bin/bash
# Copies what tools/collect_profiles.sh wrote under gpurun_out/<dir>/ into profiles/ under the round's names (run here, after the
# gpurun call has merged its output).  usage: tools/keep_profiles.sh r05 r05
set -u
S=gpurun_out/$1; R=$2; P=profiles
cpf() { [ -s "$1" ] && cp "$1" "$2" || echo "missing: $1"; }
cpf $S/bench.json $P/${R}_bench.json
cpf $S/bench_driver_args.json $P/${R}_bench_driver_args.json
st=$(ls $S/kt/*/*kernel_stats.csv $S/kt/*kernel_stats.csv 2>/dev/null | head -1); cpf "$st" $P/${R}_cfg4_kernel_stats.csv
grep '^{' $S/kt_bench.log | tail -1 > $P/${R}_cfg4_kernel_stats_bench_line.json
for k in spec nospec dist; do
  st=$(ls $S/kt_$k/*/*kernel_stats.csv $S/kt_$k/*kernel_stats.csv 2>/dev/null | head -1); cpf "$st" $P/${R}_cfg4_${k}_kernel_stats.csv
  cpf $S/kt_${k}_breakdown.txt $P/${R}_cfg4_${k}_kernel_breakdown.txt
done
cpf $S/pmc_summary.csv $P/${R}_cfg4_pmc_summary.csv
for f in bench_cfg2 bench_cfg3 bench_half_half bench_norm8_half bench_index_world1_inflight1 bench_index_world1_inflight2 bench_index_world1_inflight3 \
         bench_inflight1 bench_inflight3 bench_cfg5 bench_ranks_on_one_gpu_world2 bench_ranks_on_one_gpu_world8 rank_alone rank_alone_equal_bands rank_alone_cfg5 \
         rank_alone_python_replay rows; do cpf $S/$f.json $P/${R}_$f.json; done
for f in bench_hbm bench_sort shard_host_time bench_launch graph_probe tile_profile tile_profile_index_order ab_tile_order bench_atomic ab_shard ab_blocks ab_radix_small ab_layer_pipeline long_run_layers; do cpf $S/$f.txt $P/${R}_$f.txt; done
ls -la $P/${R}_* | awk '{print $5, $9}'
