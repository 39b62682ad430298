#!/bin/bash
# Development: the sharded frame at world 1 (bench.py --force-dist: RCCL communicator, own slot = device copy) and cfg5 through
# gsx_shard_render_frame_keys, one library against another on ONE box, interleaved.  usage: tools/ab_shard.sh base new ...
# ("new" = the in-tree libgsx.so, anything else = build_variants/libgsx_<name>.so)
for rep in 1; do
  for v in "$@"; do
    if [ "$v" = new ]; then unset GSX_LIB; else export GSX_LIB=$PWD/build_variants/libgsx_$v.so; fi
    for L in 1 2; do
      r=$(python3 bench.py --force-dist --no-cpu-baseline --no-pmc --no-extra-legs --dist-frames-in-flight $L 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
pr=d.get('per_rank',{})
print('fps', d['value'], 'verdict_wait_us', pr.get('verdict_wait_us_per_frame'), 'rounds', pr.get('exchange_rounds_per_frame'), 'repair', pr.get('frames_with_repair_round'), 'launches', d.get('launches_per_frame'))")
      echo "rep $rep $v dist L=$L: $r"
    done
    r=$(python3 bench.py --no-cpu-baseline --no-pmc --no-extra-legs --no-robustness 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('single-GPU fps', d['value'], 'one lane', d.get('value_one_frame_in_flight'))")
    echo "rep $rep $v $r"
    python3 tools/bench_cfg5.py --edit 0 --shard 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   cfg5 shard: default', d['fps_default_schedule'], 'two lanes', d['fps_two_frames_in_flight'], 'unspeculated', d['fps_unspeculated'], d.get('shard_stats'))"
    python3 tools/bench_cfg5.py --edit 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   cfg5 single: default', d['fps_default_schedule'], 'two lanes', d['fps_two_frames_in_flight'], 'unspeculated', d['fps_unspeculated'])"
  done
done
