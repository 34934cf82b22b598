"""Static fence for the cross-lane (DPP) read hazard in the shipped gfx950 code objects.

On gfx9 a DPP instruction reads its src0 from ANOTHER lane's register; a VALU instruction that wrote that VGPR must be at least two
wait states older (one instruction or one s_nop cycle = one wait state).  The compiler keeps that distance for the DPP instructions it
emits itself, but not for those inside inline asm (the stencils' `v_fmac_f32_dpp`, csrc/wv_rb.hip / wv_h16.hip / wv_k1.hip): it does
not know they read across lanes.  A build that spilled differently once returned wrong values in one lane pair.

    python tools/dpp_hazard.py [libwaveverify_hip.so]      -> lists every DPP instruction whose src0 was written by a VALU
                                                              instruction fewer than two wait states earlier; exit 1 if any

`scan_library()` is what tests/test_dpp_hazard.py calls.  Works on the built library alone (no GPU, no sources): the device code
objects are unbundled with llvm-objdump --offloading into a scratch directory and disassembled.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
from typing import Dict, List, Tuple

OBJDUMP_CANDIDATES = ("/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump")
WAIT_STATES = 2
_REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def objdump() -> str:
    for c in OBJDUMP_CANDIDATES:
        if os.path.exists(c):
            return c
    found = shutil.which("llvm-objdump")
    if not found:
        raise RuntimeError("llvm-objdump not found")
    return found


def _vgprs(op: str) -> range:
    m = _REG.match(op.strip())
    if not m:
        return range(0)
    if m.group(1) is not None:
        return range(int(m.group(1)), int(m.group(1)) + 1)
    return range(int(m.group(2)), int(m.group(3)) + 1)


def _split(line: str) -> Tuple[str, List[str]]:
    code = line.split("//")[0].strip()
    if not code or code.endswith(":") or code.startswith(("<", ".")):
        return "", []
    parts = code.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def is_dpp(mn: str, ops: List[str]) -> bool:
    if "_dpp" in mn:
        return True
    tail = " ".join(ops)
    return any(k in tail for k in ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "wave_shl:", "wave_shr:", "wave_rol:", "wave_ror:",
                                   "row_mirror", "row_half_mirror", "row_bcast:", "row_newbcast:"))


def valu_written(mn: str, ops: List[str]) -> range:
    """VGPRs a VALU instruction writes (its first operand when that is a vector register)."""
    if not mn.startswith("v_") or not ops:
        return range(0)
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return range(0)
    return _vgprs(ops[0].split()[0])


def scan_asm(text: str) -> Tuple[int, List[Dict]]:
    """-> (number of DPP instructions, hazards)."""
    insts: List[Tuple[str, List[str], str, str]] = []      # (mnemonic, operands, raw line, function)
    func = "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            func = m.group(1)
            insts.append(("<label>", [], line, func))          # a function start: nothing before it counts
            continue
        mn, ops = _split(line)
        if mn:
            insts.append((mn, ops, line.strip(), func))
    n_dpp, hazards = 0, []
    for i, (mn, ops, raw, fn) in enumerate(insts):
        if not is_dpp(mn, ops) or len(ops) < 2:
            continue
        n_dpp += 1
        src = _vgprs(ops[1].split()[0])                      # src0: the operand read from the other lane
        if not src:
            continue
        ws, j = 0, i - 1
        while j >= 0 and ws < WAIT_STATES:
            pmn, pops, praw, _ = insts[j]
            if pmn == "<label>":
                break
            if pmn == "s_nop":
                ws += int(pops[0], 0) + 1
            else:
                w = valu_written(pmn, pops)
                if w and set(w) & set(src):
                    hazards.append({"function": fn, "dpp": raw, "writer": praw, "wait_states": ws})
                    break
                ws += 1
            j -= 1
    return n_dpp, hazards


def scan_library(lib_path: str) -> Dict:
    od = objdump()
    tmp = tempfile.mkdtemp(prefix="wv_dpp_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([od, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not cos:
            raise RuntimeError(f"no gfx950 code object found in {lib_path}")
        total, hazards, per = 0, [], {}
        for co in cos:
            text = subprocess.run([od, "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
            n, hz = scan_asm(text)
            per[co.split(".")[-2] if False else co] = n
            total += n
            hazards += hz
        return {"code_objects": len(cos), "dpp_instructions": total, "hazards": hazards, "per_object": per}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "waveverify_amd", "lib", "libwaveverify_hip.so")
    r = scan_library(lib)
    print(f"{lib}: {r['code_objects']} code objects, {r['dpp_instructions']} DPP instructions, {len(r['hazards'])} hazards")
    for h in r["hazards"][:50]:
        print(f"  {h['function'][:90]}\n    writer: {h['writer']}\n    dpp:    {h['dpp']}   (wait states between: {h['wait_states']})")
    sys.exit(1 if r["hazards"] else 0)
