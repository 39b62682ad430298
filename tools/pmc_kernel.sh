#!/bin/bash
# Hardware counters of one kernel over a short bench run, one rocprofv3 pass per counter group (never with sys-trace).
# (--no-pmc: the profiled process must not start rocprofv3 children of its own once its GPU is initialised)
# usage: tools/pmc_kernel.sh <out dir under gpurun_out> <kernel substring> [bench args...]
OUT=gpurun_out/$1; K=$2; shift 2
mkdir -p $OUT
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 bench.py --steps 6 --warmup 3 --min-steps 0 --no-cpu-baseline --no-pmc "$@" > $OUT/g$i.log 2>&1
done
python3 - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, k = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for c, v in sorted(acc.items()):
        line = f"{c:28s} launches {len(v):5d}  mean {sum(v)/len(v):16.1f}  max {max(v):16.1f}"
        print(line); fo.write(line + "\n")
PY
