// Shared helpers for libdiamond_hip (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/diamond_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DMD_GN_EPS 1e-5  // models/blocks.py:13
#define DMD_MAX_DEVICES 16  // per-device launch state (function attributes, CU counts) is indexed by hipGetDevice()

// The dynamic-LDS array of a kernel.  One spelling for all of them: tests/simt compiles these sources for the host, where
// LDS is ordinary memory, and defines the macro before this header is read.
#ifndef DMD_DYNAMIC_LDS
#define DMD_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

void dmd_set_error(const char* fmt, ...);

// An integer switch from the environment, read ONCE (and again after dmd_reload_env(), the tests' hook): the launchers do
// not call getenv per launch.     static DmdEnvInt cap{"DIAMOND_WGRAD_MAX_WG", 256};  ...  cap.get()
int dmd_env_generation();
// (launchers may be called from several host threads: generation and value travel together in one atomic word)
struct DmdEnvInt {
  const char* name;
  int def;
  std::atomic<long long> gen_val{-1};  // (generation << 32) | (unsigned) value; -1: never read
  int get() {
    const int g = dmd_env_generation();
    long long gv = gen_val.load(std::memory_order_acquire);
    if (gv < 0 || (int)(gv >> 32) != g) {
      const char* e = getenv(name);
      const int v = e ? atoi(e) : def;
      gv = ((long long)g << 32) | (unsigned int)v;
      gen_val.store(gv, std::memory_order_release);
    }
    return (int)(unsigned int)(gv & 0xffffffffll);
  }
};

#define DMD_CHECK_ARG(cond, ...)  \
  do {                            \
    if (!(cond)) {                \
      dmd_set_error(__VA_ARGS__); \
      return 1;                   \
    }                             \
  } while (0)

#define DMD_LAUNCH_CHECK()                                            \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      dmd_set_error("launch failed: %s", hipGetErrorString(e_));      \
      return 2;                                                       \
    }                                                                 \
  } while (0)

// F.silu = x * sigmoid(x).  Accurate expf and IEEE division on purpose (no fast-math): the
// conv kernels are MFMA-bound, the prologue VALU work is hidden.
__device__ __forceinline__ float dmd_silu(float v) { return v / (1.0f + expf(-v)); }
// v_exp_f32 / v_rcp_f32 form (1 ulp each): used where DMD_PRECISION_F16X2 is requested
__device__ __forceinline__ float dmd_silu_fast(float t) {
  return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
}
__device__ __forceinline__ float dmd_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// mean / rstd of a GroupNorm group from T per-tile fp64 partial sums (fixed order).
__device__ __forceinline__ void dmd_finalize_stats(const double* st, int T, double count, float* mean, float* rstd) {
  double s = 0.0, ss = 0.0;
  for (int t = 0; t < T; ++t) {
    s += st[2 * t];
    ss += st[2 * t + 1];
  }
  const double m = s / count;
  double var = ss / count - m * m;
  var = var < 0.0 ? 0.0 : var;
  *mean = (float)m;
  *rstd = (float)(1.0 / sqrt(var + (double)DMD_GN_EPS));
}

__device__ __forceinline__ double dmd_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// per-channel affine of a GroupNorm(+FiLM): y = (x - mean) * a + add
__device__ __forceinline__ void norm_entry(const dmd_norm& nm, int n, int c, int C, double count, float* mean,
                                           float* a, float* add) {
  const int G = C / DMD_GN_GROUP > 0 ? C / DMD_GN_GROUP : 1;
  const int g = c / DMD_GN_GROUP;
  float m, rstd;
  dmd_finalize_stats(nm.stats + ((size_t)(n * G + g) * nm.stat_tiles) * 2, nm.stat_tiles, count, &m, &rstd);
  float mul = nm.mul ? nm.mul[(size_t)n * nm.mul_stride + c] : 1.0f;
  if (nm.mul_plus_one) mul = 1.0f + mul;
  *mean = m;
  *a = rstd * mul;
  *add = nm.add ? nm.add[(size_t)n * nm.add_stride + c] : 0.0f;
}
