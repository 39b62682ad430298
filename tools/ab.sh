#!/bin/bash
# Development: compare library variants (tools/build_variant.sh) on one box, interleaved.  usage: tools/ab.sh "<bench args>" base v1 v2 ...
ARGS=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset GSX_LIB; else export GSX_LIB=$PWD/wgpu_3dgs_viewer_app_amd/variants/libgsx_$v.so; fi
    r=$(python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('pass_ms_per_frame_rank0',''))")
    echo "rep $rep $v: $r"
  done
done
