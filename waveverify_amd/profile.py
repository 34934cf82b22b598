"""Python view of the library's per-launch HIP-event profiler (wv_profile_* in the C ABI)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

from . import _lib


def enable(on: bool = True) -> None:
    _lib.check(_lib.load().wv_profile_enable(int(on)))


def reset() -> None:
    _lib.check(_lib.load().wv_profile_reset())


def collect() -> List[Dict]:
    """Synchronise and return [{name, kernel, role, launches, ms, flops, bytes}] (totals)."""
    lib = _lib.load()
    n = lib.wv_profile_collect(-1, None, 0, None, None, None, None)
    out = []
    name = C.create_string_buffer(256)
    launches = C.c_int64()
    ms, fl, by = C.c_double(), C.c_double(), C.c_double()
    for i in range(n):
        _lib.check(lib.wv_profile_collect(i, name, 256, C.byref(launches), C.byref(ms), C.byref(fl),
                                          C.byref(by)))
        full = name.value.decode()
        kernel, _, role = full.partition("|")
        out.append(dict(name=full, kernel=kernel, role=role, launches=launches.value, ms=ms.value,
                        flops=fl.value, bytes=by.value))
    return out
