#!/bin/bash
# SQ counter pass for one K1 shape: tools/pmc_sq.sh <outdir> C T B
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
  --output-format csv -d $GRAFT_REPO_ROOT/$out -- python3 $GRAFT_REPO_ROOT/tools/kone.py "$@" > /dev/null 2>&1
