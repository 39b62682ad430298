#!/bin/bash
# Development: compare library variants (tools/build_variant.sh) on one box, interleaved.  usage: tools/ab.sh "<bench args>" base v1 v2 ...
ARGS=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset GSX_LIB; else export GSX_LIB=$PWD/build_variants/libgsx_$v.so; fi
    r=$(python bench.py --no-cpu-baseline --no-pmc $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fps', d['value'], 'one', d.get('value_one_frame_in_flight'), 'unspec', d.get('value_unspeculated'), 'project_us', d['roofline']['avg_launch_us'], 'geom_us', d.get('roofline_speculated',{}).get('avg_launch_us'), 'check', d.get('frame_check',{}).get('equal_to_unspeculated_single_pass'))")
    echo "rep $rep $v: $r"
  done
done
