#!/bin/bash
# SQ counter passes over tools/rbbench.py --fused-only (run on the GPU box from the repo root):
#   tools/rbpmc.sh <outdir under gpurun_out/> [rbbench args]
# Prints, per kernel symbol, each counter averaged over its dispatches.
set -e
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
  --kernel-trace --output-format csv -d $out/p1 -- python3 $GRAFT_REPO_ROOT/tools/rbbench.py --fused-only "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/p2 -- python3 $GRAFT_REPO_ROOT/tools/rbbench.py --fused-only "$@" > $out/p2.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rb_kernel" not in k and "k1_kernel" not in k: continue
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
