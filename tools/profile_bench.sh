#!/bin/bash
# rocprofv3 evidence for the headline bench command (run on the GPU box from the repo root):
#   tools/profile_bench.sh <outdir under gpurun_out/>
# 1. --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu-baseline`
# 2. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM traffic) and the SQ busy counters (MFMA utilisation)
set -e
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --output-format csv -d $out/pmc_sq -- python3 $B > $out/pmc_sq.log 2>&1
# stamp every pass with the library build it ran on (tools/pmc_traffic.py refuses passes without a stamp or with different ones)
for d in stats pmc_fetch pmc_write pmc_sq; do (cd $GRAFT_REPO_ROOT && python3 -c "from waveverify_amd import _lib; print(_lib.load().wv_version().decode())") > $out/$d/library.txt; done
find $out -name "*.csv" | head -20
