#!/bin/bash
# Development: build libgsx with extra compiler defines into wgpu_3dgs_viewer_app_amd/variants/libgsx_<name>.so (git-ignored,
# travels to the GPU box) so that several builds can be compared on the SAME box: GSX_LIB=<path> python bench.py ...
# usage: tools/build_variant.sh <name> "<extra flags>"
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/wgpu_3dgs_viewer_app_amd/csrc
B=$SRC/build_$NAME
mkdir -p $B $ROOT/wgpu_3dgs_viewer_app_amd/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result $EXTRA"
objs=""
for f in $(grep '^SRCS' $SRC/Makefile | cut -d= -f2); do
  o=$B/$f.o
  # only recompile what the defines can change: reuse the main build's object when the source does not mention GSX_VAR_
  if grep -q "GSX_VAR_" $SRC/$f || [ ! -f $SRC/build/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC/$f -o $o &
  else
    cp $SRC/build/$f.o $o
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/wgpu_3dgs_viewer_app_amd/variants/libgsx_$NAME.so $objs -ldl
echo built $ROOT/wgpu_3dgs_viewer_app_amd/variants/libgsx_$NAME.so
