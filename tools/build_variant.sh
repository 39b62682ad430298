#!/bin/bash
# Development: build libgsx with extra compiler defines into build_variants/libgsx_<name>.so (outside the package; git-ignored by
# `build_*/`, travels to the GPU box while it exists — delete build_variants/ when the comparison is settled) so that several
# builds can be compared on the SAME box: GSX_LIB=<path> python bench.py ...   (tools/ab.sh)
# usage: tools/build_variant.sh <name> "<extra flags>"     e.g.  tools/build_variant.sh geom2 "-DGSX_EXPERIMENT_GEOM2=1"
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/wgpu_3dgs_viewer_app_amd/csrc
B=$ROOT/build_variants/obj_$NAME
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result -I$SRC $EXTRA"
objs=""
for f in $(grep '^SRCS' $SRC/Makefile | cut -d= -f2); do
  o=$B/$f.o
  # only recompile what the defines can change: reuse the main build's object when the source does not mention GSX_EXPERIMENT_
  if grep -q "GSX_EXPERIMENT_" $SRC/$f $SRC/*.h || [ ! -f $SRC/build/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC/$f -o $o &
  else
    cp $SRC/build/$f.o $o
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/libgsx_$NAME.so $objs -ldl
echo built $ROOT/build_variants/libgsx_$NAME.so
