#!/usr/bin/env python3
"""Reduce two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) to HBM bytes per launch per kernel.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

Counters are KiB; on gfx950 FETCH_SIZE counts 64 B requests as 32 B, hence traffic = 2*FETCH + WRITE
(MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    """Kernel symbol -> the name the library's profiler (and bench.py) uses."""
    m = re.match(r"void wv::(\w+)_kernel<wv::Tile<(\d+), (\d+), (\d+), (\d+)>((?:, [-\w]+)*)\s*>", name)
    if not m:
        m2 = re.match(r"(?:void )?wv::(\w+)_kernel", name)
        return m2.group(1) if m2 else name
    base, bm, bn, wm, wn, extra = m.groups()
    ex = [e.strip() for e in extra.split(",") if e.strip()]
    if base == "dw_pw":
        base = {"0": "pw", "1": "dwconv_pw"}.get(ex[0] if ex else "", base)
    elif base == "pw_dw":
        ks = ex[0] if ex else "0"
        rm = ex[1] if len(ex) > 1 else "-1"
        res = ex[2] if len(ex) > 2 else "true"
        if rm == "-2":
            base = "spec_add"
        elif rm != "-1":
            base = "convtr_pw"
        elif ks == "5":
            base = "pw_dw_k5" if res in ("true", "1") else "pw_dw_k5_nr"
    elif base == "pw_dw_h":
        base = "pw_dw_k5_h" if ex and ex[0] == "5" else "pw_dw_h"
    return f"{base}<{bm},{bn},{wm},{wn}>"


def reduce_pass(d: str, counter: str):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter or "wv::" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"]) * 1024.0
                n[k] += 1
    return tot, n


def main():
    fetch, nf = reduce_pass(sys.argv[1], "FETCH_SIZE")
    write, nw = reduce_pass(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline` (f32); counters are KiB; gfx950 correction per "
                   "MI355X_MICROARCH.md section HBM: traffic = 2*FETCH_SIZE + WRITE_SIZE",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
        if not nw.get(k):
            continue
        f, w = fetch[k] / nf[k], write[k] / nw[k]
        out["kernels"][k] = {"launches": nf[k], "fetch_size_bytes_per_launch": f,
                             "write_size_bytes_per_launch": w, "traffic_bytes_per_launch": 2 * f + w}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
