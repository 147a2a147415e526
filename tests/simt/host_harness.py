"""TEST INFRASTRUCTURE: drives the product's HOST-side orchestration (engine.py, blocks.py, inner_model.py, denoiser.py,
unet_train.py: parameter packing, FiLM tables, statistics plumbing, launch parameters, the recorded-tape backward) on CPU
tensors against the SIMT-interpreter build of the kernels, so that `pytest -m "not gpu"` can hold whole-network results
against the reference's goldens.

The product cannot do this by itself -- diamond_amd.native refuses the interpreter library and rejects CPU tensors.  This
context manager monkeypatches exactly those two guards (native._lib, native.require_gpu) and the stream query for the duration of a test and restores them."""
from __future__ import annotations

import contextlib

from diamond_amd import native as nv

from . import loader as S


@contextlib.contextmanager
def engine_on_interpreter():
    saved = (nv._lib, nv.require_gpu, nv.stream)
    nv._lib = nv._Lib(S.lib())
    nv.require_gpu = lambda t: None
    nv.stream = lambda: None
    try:
        yield
    finally:
        nv._lib, nv.require_gpu, nv.stream = saved
