// TEST INFRASTRUCTURE -- stands in for <hip/hip_runtime.h> when tests/simt/build.sh compiles diamond_amd/csrc/*.hip as HOST C++
// for the SIMT interpreter (tests/simt/simt.h).  Never on the product's include path.
#pragma once
#include "../../simt.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
// the one dynamic-LDS array a kernel may declare (diamond_amd/csrc/dmd_common.h spells the declaration through this macro)
#define DMD_DYNAMIC_LDS(type, name) type* const name = reinterpret_cast<type*>(simt::g_block->dyn_lds)

#define threadIdx (simt::g_lane->tid)
#define blockIdx (simt::g_block->bid)
#define blockDim (simt::g_block->bdim)
#define gridDim (simt::g_block->gdim)
#define warpSize 64

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef unsigned uint2 __attribute__((ext_vector_type(2)));
typedef unsigned uint4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));

// ---- runtime API: launches run synchronously in the interpreter, everything else succeeds --------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "simt"; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
// few "CUs": the persistent kernels then walk several tiles per workgroup in the tests (SIMT_NUM_CUS overrides)
inline hipError_t hipDeviceGetAttribute(int* v, int, int) {
  const char* e = getenv("SIMT_NUM_CUS");
  *v = e ? atoi(e) : 3;
  return hipSuccess;
}
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n) { memcpy(dst, sym, n); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                                                     \
  do {                                                                                                                \
    const dim3 g_ = (grid), b_ = (block);                                                                             \
    (void)(stream);                                                                                                   \
    simt::launch({g_.x, g_.y, g_.z}, {b_.x, b_.y, b_.z}, (size_t)(lds), [=]() { kernel(__VA_ARGS__); });              \
  } while (0)

// ---- device functions ------------------------------------------------------------------------------------------------------
#define __syncthreads() simt::barrier()
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) { return simt::shfl_xor(v, mask, width); }

#define __builtin_amdgcn_mfma_f32_16x16x4f32 simt::mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_16x16x16f16 simt::mfma_f32_16x16x16f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 simt::mfma_f32_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 simt::mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_32x32x8f16 simt::mfma_f32_32x32x8f16
#define __builtin_amdgcn_perm simt::perm
#define __builtin_amdgcn_update_dpp simt::update_dpp
#define __builtin_amdgcn_readlane simt::readlane
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))

template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
inline int min(int a, int b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }

// dmd_attention.hip: lw = {fp16(p0 - h0), fp16(p1 - h1)}, hw = {h0, h1} (v_fma_mix{lo,hi}_f16: fp32 fma, one rounding to fp16)
inline unsigned simt_split_low_pair(float p0, float p1, unsigned hw) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, hw);
  const h2 l = {(_Float16)fmaf(p0, 1.0f, -(float)h.x), (_Float16)fmaf(p1, 1.0f, -(float)h.y)};
  return __builtin_bit_cast(unsigned, l);
}
#define ATT_SPLIT_LOW_PAIR(lw, p0, p1, hw) (lw) = simt_split_low_pair((p0), (p1), (hw))

// dmd_conv_f16ws.hip: its inline-assembly helpers in C++ (loads are synchronous here, waits and register ties are no-ops)
#define WS_SYNCTHREADS 1
#define WS_HOST_HELPERS 1
template <class V> inline void ws_aload(V& dst, const V* src) { dst = *src; }
template <bool FIRST, class V> inline void ws_aload(V& dst, const void* base, unsigned voff) { dst = *(const V*)((const char*)base + voff); }
template <int N, class V> inline void ws_await(V&) {}
inline unsigned ws_low_pair(float x0, float x1, unsigned h01) { return simt_split_low_pair(x0, x1, h01); }
template <int N, class V, int M> inline void ws_await_set(V (&)[M]) {}
template <int N, class V, int M> inline void ws_use_all(V (&)[M]) {}
#define WS_OPAQUE(t) ((void)(t))
#define WS_WAIT_VM(n) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) simt::readfirstlane(x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) simt::global_load_lds((const void*)(uintptr_t)(g), (void*)(uintptr_t)(l), (size), (off))
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline int __mul24(int a, int b) { return (int)(((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
