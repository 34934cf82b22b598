#!/bin/bash
# Evidence of the f16-operand mode (run on the GPU box from the repo root):  tools/profile_f16.sh <outdir under gpurun_out/>
# the mode's own bench line (embed + detect, 256 clips: BASELINE configs[1]), rocprofv3 kernel stats of the same command, separate
# FETCH_SIZE / WRITE_SIZE / SQ counter passes, the detector_stress line (configs[4]) and the per-kernel tables next to the exact mode.
set -e
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --precision f16 --steps 10 --warmup 3 > $out/bench_f16.json 2> $out/bench_f16.err
python3 bench.py --workload detector_stress --precision f16 --steps 10 --warmup 3 > $out/bench_f16_detector_stress.json 2> $out/bench_f16_ds.err
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/f16_stats -- python3 $R/bench.py --precision f16 --steps 10 --warmup 3 --no-cpu-baseline > $out/f16_stats.log 2>&1
echo "kernel stats done"
B16="$R/bench.py --precision f16 --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f16_pmc_fetch -- python3 $B16 > $out/f16_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/f16_pmc_write -- python3 $B16 > $out/f16_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --output-format csv -d $out/f16_pmc_sq -- python3 $B16 > $out/f16_pmc_sq.log 2>&1
echo "counter passes done"
cd $R
for d in f16_stats f16_pmc_fetch f16_pmc_write f16_pmc_sq; do python3 -c "from waveverify_amd import _lib; print(_lib.load().wv_version().decode())" > $out/$d/library.txt; done
python3 tools/pmc_traffic.py traffic $out/f16_pmc_fetch $out/f16_pmc_write "bench.py --precision f16 --steps 5 --warmup 2 --no-cpu-baseline\` (tools/profile_f16.sh)" > $out/pmc_traffic_f16.json
python3 tools/pmc_traffic.py busy $out/f16_pmc_sq > $out/mfma_busy_f16.json
python3 tools/g16time.py > $out/g16time.txt 2>/dev/null
python3 tools/h16time.py > $out/h16time.txt 2>/dev/null
python3 tools/rhbench.py > $out/rhbench.txt 2>/dev/null
echo "all done"
