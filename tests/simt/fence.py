"""TEST INFRASTRUCTURE: numpy arrays between inaccessible pages.  A kernel (run by the SIMT interpreter) that reads or writes
beyond a tensor it was handed faults immediately -- on the GPU such an access lands in a neighbouring allocation and goes
unnoticed.  `fenced(a)` copies `a` so that its LAST byte touches a PROT_NONE page (`tight="start"`: its first byte follows one)."""
from __future__ import annotations

import ctypes
import mmap

import numpy as np

_libc = ctypes.CDLL(None, use_errno=True)
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
PAGE = mmap.PAGESIZE
_keep = []  # the mappings live as long as the process (tests)


def fenced(a: np.ndarray, tight: str = "end") -> np.ndarray:
    a = np.ascontiguousarray(a)
    nbytes = max(a.nbytes, 1)
    body = (nbytes + PAGE - 1) // PAGE * PAGE
    m = mmap.mmap(-1, body + 2 * PAGE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    for off in (0, PAGE + body):
        if _libc.mprotect(base + off, PAGE, 0) != 0:  # PROT_NONE
            raise OSError(ctypes.get_errno(), "mprotect")
    start = PAGE + (body - nbytes if tight == "end" else 0)
    if tight == "end":
        start -= start % 16 if (body - nbytes) >= 16 else 0  # keep 16-byte alignment (at most 15 bytes of slack)
    out = np.frombuffer(m, dtype=a.dtype, count=a.size, offset=start).reshape(a.shape)
    out[...] = a
    _keep.append(m)
    return out
