#!/bin/bash
# Evidence of the detector's f16-operand mode (run on the GPU box from the repo root):  tools/profile_f16.sh <outdir under gpurun_out/>
# its bench line, rocprofv3 kernel stats of the same command, the per-kernel table next to the exact mode.
set -e
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --workload detector_stress --precision f16 --steps 10 --warmup 3 > $out/bench_f16.json 2> $out/bench_f16.err
echo "bench line done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/f16_stats -- python3 $R/bench.py --workload detector_stress --precision f16 --steps 10 --warmup 3 --no-cpu-baseline > $out/f16_stats.log 2>&1
echo "kernel stats done"
cd $R
python3 tools/h16time.py > $out/h16time.txt 2>/dev/null
echo "all done"
