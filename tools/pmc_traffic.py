#!/usr/bin/env python3
"""Reduce rocprofv3 counter passes over the headline bench command (tools/profile_bench.sh) to per-kernel figures.

    python tools/pmc_traffic.py traffic <pmc_fetch dir> <pmc_write dir> ["<command> (<script>)"]  > profiles/rNN_pmc_traffic.json
    python tools/pmc_traffic.py busy    <pmc_sq dir>                     > profiles/rNN_mfma_busy.json

traffic: HBM bytes per launch.  Counters are KiB; on gfx950 FETCH_SIZE counts 64 B requests as 32 B, hence
         traffic = 2*FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section).
busy:    MFMA utilisation per kernel symbol = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs), with the
         kernel's cycles taken from GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), plus the effective clock."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    """Kernel symbol -> the name the library's profiler (and bench.py) uses."""
    m = re.match(r"void wv::k1_kernel<wv::K1<(\d+), (\d+), (\d+)>, (-?\d+), (\d+), (true|false|[01]), (\d+)>", name)
    if m:
        nt, bkc, bm, epi, ldr, res, ns = m.groups()
        epi, ldr = int(epi), int(ldr)
        res = res in ("true", "1")
        base = "convtr_pw" if ldr >= 2 else ("pw_dw_k5" if epi == 0 and res else "pw_dw_k5_nr" if epi == 0 else "pw_dw" if epi == 1 else "pw_dw_s")
        tag = "reg" if ldr else ("dma3" if ns == "3" else "dma")
        return f"{base}<{bm},{32 * int(nt)},{tag}>"          # spec_add launches share the pw_dw_k5 symbol
    m = re.match(r"void wv::stft_k1_kernel<wv::K1<(\d+), (\d+), (\d+)>, (true|false|[01])>", name)
    if m:
        fused = m.group(4) in ("true", "1")
        return f"{'stft_spec' if fused else 'stft_logmag'}<{m.group(3)},{32 * int(m.group(1))},k1>"
    m = re.search(r"rb_kernel<wv::\(anonymous namespace\)::RB<(\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        c, ng, nt = int(m.group(1)), int(m.group(2)), int(m.group(3))
        return f"resblock<{c},{ng * (32 * nt - 4) + 4}>"
    # the f16 mode's kernels (csrc/wv_h16.hip), named as the library's profiler names them where the symbol allows
    m = re.search(r"rh_kernel<wv::\(anonymous namespace\)::RH<(\d+), (\d+), (\d+),", name)
    if m:
        c, ng, nt = int(m.group(1)), int(m.group(2)), int(m.group(3))
        return f"resblock16<{c},{ng * (32 * nt - 4) + 4}>"
    m = re.search(r"spec16_kernel<wv::\(anonymous namespace\)::SP<(\d+), (\d+)>", name)
    if m:
        return f"spec16<{m.group(1)},hop{m.group(2)}>"
    m = re.search(r"conv16s_kernel<(\d+), (true|false|[01])>", name)
    if m:
        return f"conv16<k{m.group(1)},lds>"                     # one symbol per tap count (the profiler's name also carries M x K)
    m = re.search(r"conv16u_kernel<(true|false|[01])>", name)
    if m:
        return f"conv16<k2,up{',short' if m.group(1) in ('true', '1') else ''},lds>"   # the decoder's upsample units (one symbol for all four)
    m = re.search(r"conv16_kernel<(\d+), (\d+), (true|false|[01])>", name)
    if m:
        return f"conv16<{m.group(1)}x{m.group(2)}{',flat' if m.group(3) in ('true', '1') else ''}>"   # several layers share a symbol
    name = name.replace("(anonymous namespace)::", "")
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
    return f"{base}<{bm},{bn},{wm},{wn}>"


def stamp(d: str) -> str:
    """The library build a counter pass ran on: tools/profile_bench.sh / profile_f16.sh write wv_version() into <pass dir>/library.txt
    right after the pass.  Passes without a stamp, or stamps that differ between the passes merged into one file, are refused: a
    figure built from two builds is tied to neither."""
    import os
    p = os.path.join(d, "library.txt")
    if not os.path.exists(p):
        raise SystemExit(f"{d}: no library.txt stamp (run the pass through tools/profile_bench.sh or tools/profile_f16.sh)")
    return open(p).read().strip()


def rows(d: str, suffix: str):
    for f in glob.glob(f"{d}/**/*_{suffix}.csv", recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


def reduce_pass(d: str, counters):
    tot = {c: defaultdict(float) for c in counters}
    n = {c: defaultdict(int) for c in counters}
    for r in rows(d, "counter_collection"):
        c = r["Counter_Name"]
        if c not in tot or "wv::" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        tot[c][k] += float(r["Counter_Value"])
        n[c][k] += 1
    return tot, n


def durations(d: str):
    t, n = defaultdict(float), defaultdict(int)
    for r in rows(d, "kernel_trace"):
        if "wv::" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n[k] += 1
    return t, n


def traffic(fetch_dir, write_dir, command="bench.py --steps 5 --warmup 2 --no-cpu-baseline` (tools/profile_bench.sh)"):
    lib_f, lib_w = stamp(fetch_dir), stamp(write_dir)
    if lib_f != lib_w:
        raise SystemExit(f"refusing to merge counter passes from different builds: {fetch_dir}: {lib_f} / {write_dir}: {lib_w}")
    f, nf = reduce_pass(fetch_dir, ["FETCH_SIZE"])
    w, nw = reduce_pass(write_dir, ["WRITE_SIZE"])
    f, nf, w, nw = f["FETCH_SIZE"], nf["FETCH_SIZE"], w["WRITE_SIZE"], nw["WRITE_SIZE"]
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `" + command + "; counters are KiB; gfx950 correction per "
                   "MI355X_MICROARCH.md section HBM: traffic = 2*FETCH_SIZE + WRITE_SIZE",
           "library": lib_f, "kernels": {}}
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0.0))):
        if not nw.get(k):
            continue
        fb, wb = f[k] * 1024.0 / nf[k], w[k] * 1024.0 / nw[k]
        out["kernels"][k] = {"launches": nf[k], "fetch_size_bytes_per_launch": fb,
                             "write_size_bytes_per_launch": wb, "traffic_bytes_per_launch": 2 * fb + wb}
    return out


def busy(sq_dir):
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
             "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"]
    tot, n = reduce_pass(sq_dir, names)
    dur, nd = durations(sq_dir)
    out = {"note": "rocprofv3 --pmc pass over the bench command (tools/profile_bench.sh).  mfma_busy = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of all SIMD cycles in "
                   "which the matrix pipe is busy; clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel time (profiled passes "
                   "clock ~3 % lower than un-profiled ones); wait_any / wait_inst = SQ_WAIT_ANY / SQ_WAIT_INST_ANY "
                   "over SQ_WAVE_CYCLES (waves parked at s_waitcnt or a barrier / stalled at issue, e.g. behind the "
                   "matrix pipe); valu_per_mfma = other VALU instructions per MFMA",
           "library": stamp(sq_dir), "kernels": {}}
    tt = sum(dur.values())
    for k in sorted(dur, key=lambda k: -dur[k]):
        L = n["GRBM_GUI_ACTIVE"].get(k, 0)
        if not L:
            continue
        cyc = tot["GRBM_GUI_ACTIVE"][k] / 8.0
        mf = tot["SQ_INSTS_MFMA"][k]
        out["kernels"][k] = {
            "launches": L, "share_of_kernel_time": round(dur[k] / tt, 4),
            "avg_us": round(dur[k] / nd[k] / 1e3, 1),
            "mfma_busy": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"][k] / (cyc * 1024.0), 4) if cyc else None,
            "clock_ghz": round(cyc / dur[k], 3) if dur[k] else None,
            "wait_any": round(tot["SQ_WAIT_ANY"][k] / tot["SQ_WAVE_CYCLES"][k], 3) if tot["SQ_WAVE_CYCLES"][k] else None,
            "wait_inst": round(tot["SQ_WAIT_INST_ANY"][k] / tot["SQ_WAVE_CYCLES"][k], 3) if tot["SQ_WAVE_CYCLES"][k] else None,
            "valu_per_mfma": round((tot["SQ_INSTS_VALU"][k] - mf) / mf, 2) if mf else None}
    return out


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        json.dump(traffic(sys.argv[2], sys.argv[3], *sys.argv[4:5]), sys.stdout, indent=1)
    else:
        json.dump(busy(sys.argv[2]), sys.stdout, indent=1)
    print()
