"""ctypes binding of libdiamond_hip.so (the C ABI declared in include/diamond_hip.h).

Only raw device pointers, sizes and a hipStream_t cross the boundary.  There is NO CPU or
PyTorch fallback: if the library is missing, or an op is given a non-GPU tensor, the call
raises.  The stream is always torch's current stream so that torch ops (allocation, RNG,
optimizer, RCCL) and the hand-written kernels are ordered and graph-capturable together.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIAMOND_LIB", os.path.join(_HERE, "libdiamond_hip.so"))  # override: development builds

PROLOGUE_NONE, PROLOGUE_NORM_SILU, PROLOGUE_NORM = 0, 1, 2
PRECISION_F32, PRECISION_F16X2 = 0, 1
GN_GROUP = 32


class NativeLibraryMissing(RuntimeError):
    pass


class Norm(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("stat_tiles", C.c_int32), ("mul_plus_one", C.c_int32), ("mul", C.c_void_p),
                ("add", C.c_void_p), ("mul_stride", C.c_int64), ("add_stride", C.c_int64)]


class ConvSrc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("C", C.c_int32), ("prologue", C.c_int32), ("norm", Norm)]


class ConvParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32),
                ("taps", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32), ("nsrc", C.c_int32),
                ("src", ConvSrc * 2), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("residual_norm", Norm), ("out", C.c_void_p), ("out_nchw", C.c_int32), ("precision", C.c_int32),
                ("out_stats", C.c_void_p), ("w_f16", C.c_void_p), ("proj_nsrc", C.c_int32), ("proj_C", C.c_int32 * 2),
                ("proj_reserved", C.c_int32), ("proj_x", C.c_void_p * 2), ("proj_w_f16", C.c_void_p), ("proj_bias", C.c_void_p),
                ("valid_h", C.c_int32), ("valid_w", C.c_int32)]


class LinearParams(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("A", C.c_void_p), ("lda", C.c_int64),
                ("W", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("accumulate", C.c_int32), ("silu", C.c_int32)]


class GnBwdParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("identity_activation", C.c_int32), ("x", C.c_void_p),
                ("norm", Norm), ("da", C.c_void_p), ("dskip", C.c_void_p), ("dx", C.c_void_p), ("workspace", C.c_void_p),
                ("dmul", C.c_void_p), ("dadd", C.c_void_p), ("W", C.c_int32), ("valid_h", C.c_int32), ("valid_w", C.c_int32),
                ("reserved", C.c_int32)]


class WgradParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32), ("taps", C.c_int32),
                ("cin_real", C.c_int32), ("src", ConvSrc), ("dy", C.c_void_p), ("workspace", C.c_void_p), ("dw", C.c_void_p),
                ("dbias", C.c_void_p), ("precision", C.c_int32), ("valid_h", C.c_int32), ("valid_w", C.c_int32), ("defer_reduce", C.c_int32)]


class WgradReduceJob(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("num_wg", C.c_int32), ("NB", C.c_int32),
                ("NCO", C.c_int32), ("NCI", C.c_int32), ("taps", C.c_int32), ("cin_real", C.c_int32), ("ld_cin", C.c_int32),
                ("c0", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("k", C.c_int32),
                ("kind", C.c_int32), ("transposed", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32), ("CoutPad", C.c_int32),
                ("CinPad", C.c_int32), ("reserved", C.c_int32)]


PACK_F32, PACK_F16X2, PACK_BIAS = 0, 1, 2


class ChecksumJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("words", C.c_int64)]


CHECKSUM_PARTS = 8


class PoolRound(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("pad", C.c_void_p), ("act", C.c_void_p), ("hx", C.c_void_p), ("cx", C.c_void_p),
                ("is_f32", C.c_int32), ("rows", C.c_int32)]


class ResetSlotsParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("K", C.c_int32), ("T", C.c_int32), ("head", C.c_int32), ("per_frame", C.c_int64),
                ("hd", C.c_int32), ("reserved", C.c_int32), ("pool", PoolRound * 2), ("pool_base", C.c_int64), ("num_dead", C.c_void_p),
                ("slot_row", C.c_void_p), ("row_slot", C.c_void_p), ("next_obs", C.c_void_p), ("ctx", C.c_void_p),
                ("act_ring", C.c_void_p), ("hx", C.c_void_p), ("cx", C.c_void_p), ("enc_in", C.c_void_p)]


CHAIN_MAX_BLOCKS = 8


class ChainBlock(C.Structure):
    _fields_ = [("skip_slot", C.c_int32), ("save_slot", C.c_int32), ("film1_mul", C.c_int32 * 2), ("film1_add", C.c_int32 * 2),
                ("film2_mul", C.c_int32), ("film2_add", C.c_int32), ("has_attn", C.c_int32), ("reserved", C.c_int32),
                ("w1", C.c_void_p), ("w2", C.c_void_p), ("wproj", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p),
                ("bproj", C.c_void_p), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("wq", C.c_void_p), ("wk", C.c_void_p),
                ("wv", C.c_void_p), ("wo", C.c_void_p), ("bqkv", C.c_void_p), ("bo", C.c_void_p)]


class LowresChainParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("nblocks", C.c_int32), ("input_save_slot", C.c_int32), ("reserved", C.c_int32),
                ("x", C.c_void_p), ("out", C.c_void_p), ("table", C.c_void_p), ("table_stride", C.c_int64),
                ("blocks", ChainBlock * CHAIN_MAX_BLOCKS)]


EXPORTS = (
    "dmd_conv2d", "dmd_conv2d_kernel_name", "dmd_conv2d_naive", "dmd_conv_stat_tiles", "dmd_pack_conv_weight", "dmd_conv2d_f16x2_eligible",
    "dmd_conv1x1_stream_eligible", "dmd_conv2d_proj_eligible", "dmd_pack_jobs", "dmd_checksums",
    "dmd_pack_conv_weight_f16x2", "dmd_linear", "dmd_attention", "dmd_attention_valid", "dmd_attention_bwd", "dmd_attention_bwd_workspace_floats",
    "dmd_edm_pack_input", "dmd_cond_embed", "dmd_edm_denoised", "dmd_euler_step", "dmd_heun_step", "dmd_quantize_u8", "dmd_reset_state",
    "dmd_dequant_gather", "dmd_resolve_deaths", "dmd_reset_slots", "dmd_merge_slots", "dmd_merge_slots_bwd", "dmd_nchw_to_nhwc",
    "dmd_nhwc_to_nchw", "dmd_gn_stats", "dmd_gn_stats_valid", "dmd_maxpool2", "dmd_lstm_pointwise", "dmd_lstm_pointwise_bwd", "dmd_categorical_sample",
    "dmd_maxpool2_bwd", "dmd_gn_bwd_workspace_bytes", "dmd_gn_silu_bwd", "dmd_wgrad_workspace_floats", "dmd_conv2d_wgrad", "dmd_wgrad_job", "dmd_wgrad_reduce_jobs",
    "dmd_lowres_chain", "dmd_lowres_chain32", "dmd_last_error", "dmd_abi_version", "dmd_reload_env",
)

# entry points that launch kernels (everything except queries / packing helpers that bench.py does not time)
LAUNCHERS = frozenset(n for n in EXPORTS if n not in (
    "dmd_conv2d_kernel_name", "dmd_conv_stat_tiles", "dmd_conv2d_f16x2_eligible", "dmd_conv1x1_stream_eligible",
    "dmd_conv2d_proj_eligible", "dmd_attention_bwd_workspace_floats", "dmd_gn_bwd_workspace_bytes", "dmd_wgrad_workspace_floats", "dmd_wgrad_job", "dmd_last_error",
    "dmd_abi_version", "dmd_reload_env"))


class LaunchProfiler:
    """Optional per-launch HIP-event timing of EVERY C-ABI launch (bench.py's roofline pass): set `native.PROFILER` and
    every call through `lib()` is bracketed by two events on torch's current stream == the stream the kernels are
    launched on.  A caller that knows more than the entry point's name (dmd_conv2d: the kernel instantiation, the
    algorithmic FLOPs and bytes) calls `annotate()` right before its launch."""

    def __init__(self) -> None:
        self.records = []  # (key, flops, bytes, event0, event1)
        self._pending = None

    def annotate(self, key: str, flops: float, nbytes: float) -> None:
        self._pending = (key, flops, nbytes)

    def call(self, name: str, fn, args):
        key, flops, nbytes = self._pending or (name, 0.0, 0.0)
        self._pending = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.records.append((key, flops, nbytes, e0, e1))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


PROFILER: Optional[LaunchProfiler] = None


class _Lib:
    """The loaded library; launching entry points go through the profiler when one is installed."""

    def __init__(self, cdll: C.CDLL) -> None:
        self._cdll = cdll

    def __getattr__(self, name: str):
        fn = getattr(self._cdll, name)
        if name not in LAUNCHERS:
            return fn

        def launch(*args):
            prof = PROFILER
            return fn(*args) if prof is None else prof.call(name, fn, args)

        return launch


def declare_signatures(L: C.CDLL) -> None:
    """restype / argtypes of every entry point of include/diamond_hip.h on a loaded library; AttributeError if the ABI is
    incomplete.  (tests/simt declares the same signatures on its host build of the kernels.)"""
    L.dmd_last_error.restype = C.c_char_p
    L.dmd_reload_env.restype = None
    for name in EXPORTS:
        getattr(L, name)  # AttributeError if the ABI is incomplete
    L.dmd_conv2d.argtypes = [C.POINTER(ConvParams), C.c_void_p]
    L.dmd_conv2d_naive.argtypes = [C.POINTER(ConvParams), C.c_void_p]
    L.dmd_conv2d_kernel_name.argtypes = [C.POINTER(ConvParams), C.c_char_p, C.c_int]
    L.dmd_linear.argtypes = [C.POINTER(LinearParams), C.c_void_p]
    L.dmd_pack_conv_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_pack_conv_weight_f16x2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_conv2d_f16x2_eligible.argtypes = [C.POINTER(ConvParams)]
    L.dmd_conv1x1_stream_eligible.argtypes = [C.POINTER(ConvParams)]
    L.dmd_conv2d_proj_eligible.argtypes = [C.POINTER(ConvParams)]
    L.dmd_pack_jobs.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    L.dmd_checksums.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.dmd_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_attention_valid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_attention_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]
    L.dmd_attention_bwd_workspace_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    L.dmd_attention_bwd_workspace_floats.restype = C.c_int64
    L.dmd_edm_pack_input.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_cond_embed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_heun_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                C.c_void_p, C.c_int64, C.c_void_p]
    L.dmd_quantize_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.dmd_dequant_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                     C.c_int, C.c_void_p]
    L.dmd_reset_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.dmd_resolve_deaths.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.dmd_reset_slots.argtypes = [C.POINTER(ResetSlotsParams), C.c_void_p]
    L.dmd_merge_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.dmd_merge_slots_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_edm_denoised.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64,
                                   C.c_void_p]
    L.dmd_euler_step.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    L.dmd_nchw_to_nhwc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_nhwc_to_nchw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_gn_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_gn_stats_valid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_maxpool2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p]
    L.dmd_lstm_pointwise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.dmd_lstm_pointwise_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p]
    L.dmd_categorical_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.dmd_conv_stat_tiles.argtypes = [C.c_int, C.c_int]
    L.dmd_maxpool2_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.dmd_gn_bwd_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.dmd_gn_bwd_workspace_bytes.restype = C.c_int64
    L.dmd_gn_silu_bwd.argtypes = [C.POINTER(GnBwdParams), C.c_void_p]
    L.dmd_wgrad_workspace_floats.argtypes = [C.POINTER(WgradParams)]
    L.dmd_wgrad_workspace_floats.restype = C.c_int64
    L.dmd_conv2d_wgrad.argtypes = [C.POINTER(WgradParams), C.c_void_p]
    L.dmd_wgrad_job.argtypes = [C.POINTER(WgradParams), C.POINTER(WgradReduceJob)]
    L.dmd_wgrad_reduce_jobs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.dmd_lowres_chain.argtypes = [C.POINTER(LowresChainParams), C.c_void_p]
    L.dmd_lowres_chain32.argtypes = [C.POINTER(LowresChainParams), C.c_void_p]


_lib: Optional[_Lib] = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(diamond_amd has no CPU/PyTorch fallback for its kernels)")
        L = C.CDLL(LIB_PATH)
        if hasattr(L, "dmd_simt_host_build"):
            raise NativeLibraryMissing(
                f"{LIB_PATH} is the SIMT-interpreter build of the kernels (tests/simt, test infrastructure): "
                "diamond_amd runs on libdiamond_hip.so only, there is no CPU path")
        declare_signatures(L)
        _lib = _Lib(L)
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().dmd_last_error().decode()}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def check_current_device(dev: torch.device) -> None:
    """Every launch goes to the CURRENT device's stream with raw pointers (stream() above): a module that lives on another GPU
    than the current one would launch on the wrong device.  torch ops switch devices by themselves, ctypes launches cannot; the
    coarse entry points (an env reset, a training forward) check it once and say what to do.  (The reference's Trainer sets the
    device per rank, trainer.py:52-53.)"""
    if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"diamond_amd: the module lives on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                           f"call torch.cuda.set_device({dev.index}) (or wrap the call in torch.cuda.device({dev.index}))")


def require_gpu(t: Tensor) -> None:
    """The one place that says it: there is no CPU path."""
    if not t.is_cuda:
        raise RuntimeError("diamond_amd kernels need GPU tensors (there is no CPU path)")


def ptr(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    require_gpu(t)
    return t.data_ptr()


def fptr(t: Optional[Tensor]) -> Optional[int]:
    if t is not None:
        assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.shape, t.stride())
    return ptr(t)


def conv_stat_tiles(h: int, w: int) -> int:
    return (h // 8) * (w // (8 if w % 16 else 16))


def cout_pad(cout: int) -> int:
    return (cout + 15) // 16 * 16


def make_norm(stats: Optional[Tensor] = None, stat_tiles: int = 0, mul: Optional[Tensor] = None,
              add: Optional[Tensor] = None, mul_stride: int = 0, add_stride: int = 0, plus_one: bool = False) -> Norm:
    n = Norm()
    n.stats = ptr(stats)
    n.stat_tiles = stat_tiles
    n.mul_plus_one = int(plus_one)
    n.mul = ptr(mul)
    n.add = ptr(add)
    n.mul_stride = mul_stride
    n.add_stride = add_stride
    return n


def pack_conv_weight(w_oihw: Tensor, cout_padded: Optional[int] = None) -> Tensor:
    """OIHW (nn.Conv2d.weight) -> packed [CinPad/16][taps][CoutPad][16] on the same device."""
    cout, cin, k, _ = w_oihw.shape
    cp = cout_padded or cout_pad(cout)
    cinp = (cin + 15) // 16 * 16
    w = w_oihw.detach().contiguous().float()
    out = torch.empty(cinp // 16 * k * k * cp * 16, device=w.device, dtype=torch.float32)
    check(lib().dmd_pack_conv_weight(fptr(w), fptr(out), cout, cin, k, cp, cinp, stream()), "dmd_pack_conv_weight")
    return out


def pack_conv_weight_f16x2(w_oihw: Tensor) -> Tensor:
    """OIHW (Cout in {32, 64}, Cin, k, k), k in {1, 3} -> [CinPad/16][k*k][h|l][2][Cout][8] fp16 split pieces (w = h + l)."""
    cout, cin, k, _ = w_oihw.shape
    assert cout in (32, 64) and k in (1, 3), (cout, k)
    cinp = (cin + 15) // 16 * 16
    w = w_oihw.detach().contiguous().float()
    out = torch.empty(cinp // 16 * k * k * 2 * cout * 16, device=w.device, dtype=torch.float16)
    check(lib().dmd_pack_conv_weight_f16x2(fptr(w), ptr(out), cout, cin, k, cinp, stream()), "dmd_pack_conv_weight_f16x2")
    return out


def pad_vector(v: Optional[Tensor], n: int) -> Optional[Tensor]:
    if v is None:
        return None
    out = torch.zeros(n, device=v.device, dtype=torch.float32)
    out[: v.numel()] = v.detach().float()
    return out


def linear(a: Tensor, w: Tensor, bias: Optional[Tensor], out: Tensor, accumulate: bool = False, silu: bool = False) -> Tensor:
    """out[M,N] (+)= a[M,K] @ w[N,K]^T + bias; row strides taken from the tensors."""
    p = LinearParams()
    p.M, p.K = a.shape
    p.N = w.shape[0]
    assert w.shape[1] == p.K and out.shape == (p.M, p.N), (a.shape, w.shape, out.shape)
    assert a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    p.A, p.lda = ptr(a), a.stride(0)
    p.W, p.ldw = ptr(w), w.stride(0)
    p.bias = ptr(bias)
    p.C, p.ldc = ptr(out), out.stride(0)
    p.accumulate, p.silu = int(accumulate), int(silu)
    if PROFILER is not None:
        PROFILER.annotate(f"linear_mfma_kernel<{'true' if p.K >= 512 else 'false'}>", 2.0 * p.M * p.N * p.K,
                          4.0 * (p.M * p.K + p.N * p.K + p.M * p.N))
    check(lib().dmd_linear(C.byref(p), stream()), "dmd_linear")
    return out
