#!/bin/bash
# Copies what tools/collect_r06.sh wrote under gpurun_out/r06/ into profiles/r06_* (run here, after the gpurun calls have merged).
set -u
S=gpurun_out/r06; P=profiles
cpf() { [ -s "$1" ] && cp "$1" "$2" || echo "missing: $1"; }
tools/keep_profiles.sh r06 r06 2>&1 | grep -v "^missing: gpurun_out/r06/\(rank_alone_equal_bands\|rank_alone_python_replay\|shard_host_time\|bench_launch\|graph_probe\|tile_profile\|tile_profile_index_order\|ab_tile_order\|bench_atomic\|ab_shard\|ab_blocks\|ab_radix_small\|ab_layer_pipeline\|long_run_layers\)" | grep "^missing"
st=$(ls $S/kt_full/*/*kernel_stats.csv $S/kt_full/*kernel_stats.csv 2>/dev/null | head -1); cpf "$st" $P/r06_cfg4_full_kernel_stats.csv
cpf $S/kt_full_breakdown.txt $P/r06_cfg4_full_kernel_breakdown.txt
cpf $S/kt_spec_timeline.txt $P/r06_cfg4_spec_timeline.txt
cpf $S/kt_nospec_timeline.txt $P/r06_cfg4_nospec_timeline.txt
for f in robustness ab_round6 rank_table rank_table_cfg5; do cpf $S/$f.txt $P/r06_$f.txt; done
ls $P/r06_* | wc -l
