"""TEST INFRASTRUCTURE: loads tests/simt/_build/libdiamond_simt.so -- diamond_amd/csrc/*.hip compiled as host C++ against the
SIMT interpreter (tests/simt/simt.h) -- for `pytest -m "not gpu"`.  Entry points take numpy arrays; only the ctypes struct
layouts and the signature table are shared with diamond_amd.native, which itself refuses to load this library."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from diamond_amd import native as nv

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libdiamond_simt.so")
_lib: Optional[C.CDLL] = None


def build() -> None:
    """build.sh under an exclusive file lock: several processes may get here at once (the ranks of a gloo test, pytest-xdist
    workers) and must not compile the same objects side by side."""
    import fcntl

    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["bash", os.path.join(_HERE, "build.sh")], check=True, stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()  # incremental: seconds when nothing changed
        L = C.CDLL(LIB_PATH)
        assert L.dmd_simt_host_build() == 1
        nv.declare_signatures(L)
        _lib = L
    return _lib


def ptr(a: Optional[np.ndarray]) -> Optional[int]:
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "contiguous arrays only"
    return a.ctypes.data


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().dmd_last_error().decode()}")
