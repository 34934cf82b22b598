#!/bin/bash
# Counter passes for one K1 shape: tools/pmc_sq.sh <outdir> C T B [pre_elu] [resid]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
  --output-format csv -d $GRAFT_REPO_ROOT/$out/p1 -- python3 $GRAFT_REPO_ROOT/tools/kone.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/p2 -- python3 $GRAFT_REPO_ROOT/tools/kone.py "$@" > /dev/null 2>&1
