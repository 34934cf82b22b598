#!/bin/bash
# HBM traffic of the fused ResnetBlock kernels (tools/rbbench.py --fused-only), separate FETCH_SIZE / WRITE_SIZE passes:
#   tools/rbtraffic.sh <outdir under gpurun_out/> [rbbench args]
# gfx950: traffic = 2*FETCH_SIZE + WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM section)
set -e
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- python3 $GRAFT_REPO_ROOT/tools/rbbench.py --fused-only "$@" > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -- python3 $GRAFT_REPO_ROOT/tools/rbbench.py --fused-only "$@" > $out/w.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"RB<(\d+), (\d+), (\d+), (\d+), (\d+)>", k)
        if not m: continue
        acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
B = 256
for k, cs in sorted(acc.items()):
    C = int(re.search(r"RB<(\d+)", k).group(1)); T = 16000 if C in (64, 96) else 8000
    fe = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    alg = 4.0 * B * C * T
    print(f"{k}: fetch {2 * fe * 1024 / 1e9:.3f} GB (x = {alg / 1e9:.3f} GB -> {2 * fe * 1024 / alg:.2f}x)   write {wr * 1024 / 1e9:.3f} GB ({wr * 1024 / alg:.2f}x)")
PY
