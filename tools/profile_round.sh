#!/bin/bash
# All rocprofv3 / bench evidence of a round in one go (run on the GPU box from the repo root):
#   tools/profile_round.sh <outdir under gpurun_out/>
# headline: kernel stats + separate FETCH_SIZE / WRITE_SIZE / SQ counter passes (tools/profile_bench.sh); training step: bench line +
# kernel stats; the other workloads' bench lines; the one-launch ResnetBlock kernels alone (times, traffic, phase shares).
# (The detector's f16 mode: tools/profile_f16.sh, its own call.)
set -e
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
R=$GRAFT_REPO_ROOT
$R/tools/profile_bench.sh $1 > $out/profile_bench.log 2>&1
echo "headline passes done" 
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_stats -- python3 $R/bench.py --workload train_step --steps 5 --warmup 2 > $out/train_stats.log 2>&1
echo "train stats done"
cd $R
python3 bench.py --workload train_step --steps 10 --warmup 3 > $out/train_step_bench.json 2> $out/train_step_bench.err
{
  echo "# bench.py workloads on the round's final binary, one box, back to back"
  echo "## longform (configs[3])"; python3 bench.py --workload longform --steps 5 --warmup 2 2>/dev/null
  echo "## detector_stress (configs[4])"; python3 bench.py --workload detector_stress --steps 10 --warmup 3 2>/dev/null
  echo "## embed_detect under torch.distributed.run, one rank (RCCL init, barrier, MAX all-reduce of the time)"
  WV_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null
  echo "## grad_allreduce, one rank (no collective is issued with one rank)"; python3 bench.py --workload grad_allreduce --steps 5 --warmup 2 2>/dev/null
} > $out/workloads.txt
echo "workloads done"
python3 tools/rbbench.py > $out/rbbench.txt 2>/dev/null
tools/rbtraffic.sh $1/rbtraffic > $out/rbtraffic.txt 2>&1
python3 tools/pmc_traffic.py traffic $out/pmc_fetch $out/pmc_write > $out/pmc_traffic.json
python3 tools/pmc_traffic.py busy $out/pmc_sq > $out/mfma_busy.json
python3 bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
echo "all done"
