#!/bin/bash
# bench.py --gpus N as the driver launches it, with the N ranks as processes on ONE GPU (GSX_BENCH_ONE_DEVICE=1: every rank takes
# device 0 and a host identity of its own, RCCL connects them over sockets on lo).  The N > 1 code runs for real, between
# processes; the rate on the line measures nothing.  usage: tools/bench_ranks_one_gpu.sh <out dir> [N ...]   (default 2 4 8)
OUT=${1:-gpurun_out/ranks}; shift
mkdir -p "$OUT"
export GSX_BENCH_ONE_DEVICE=1
for N in ${@:-2 4 8}; do
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
    bench.py --gpus "$N" > "$OUT/world$N.json" 2> "$OUT/world$N.err"
  echo "N=$N rc=$? $(cut -c1-160 "$OUT/world$N.json")"
done
