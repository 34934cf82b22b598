"""Build libwaveverify_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m waveverify_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libwaveverify_hip.so")
SOURCES = ["wv_kernels.hip", "wv_k1.hip", "wv_rb.hip", "wv_h16.hip", "wv_model.hip", "wv_ops.hip", "wv_train.hip", "wv_aug.hip", "wv_fx.hip"]
HEADERS = [os.path.join(CSRC, "wv_kernels.h"), os.path.join(CSRC, "wv_dev.h"),
           os.path.join(os.path.dirname(HERE), "include", "waveverify_hip.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """sha256 over the kernel sources and headers, 12 hex digits: wv_version() carries it, so that measurements keyed to a build
    (profiles/*_pmc_traffic.json) can tell when the kernels have changed since."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES] + HEADERS):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:12]


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src.replace(".hip", ".o"))
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
               "-ffp-contract=fast" if src in ("wv_kernels.hip", "wv_k1.hip", "wv_rb.hip", "wv_h16.hip") else "-ffp-contract=off"]
        if src == "wv_model.hip":
            cmd.append(f'-DWV_SRC_HASH="{source_hash()}"')
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
