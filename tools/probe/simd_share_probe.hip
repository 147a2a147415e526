// What does an instruction of a CO-RESIDENT wave cost the MFMA stream of a SIMD?  (round 3, HISTORY.md §3)
// 768-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue dependent-pair v_mfma_f32_32x32x16_f16 like the
// consumers of conv_f16ws_kernel; waves 8-11 (their SIMD partners) issue K instructions of one kind per 108 MFMAs;
// waves 4-7 idle at the barriers.  One s_barrier per "step" of 108 MFMAs, as in the real kernel.  Output: shader-clock
// ticks (s_memtime) the MFMA wave needs per MFMA, by partner instruction kind and density, and the partner-only time.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/simd_share_probe.hip -o tools/probe/simd_share_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { K_NONE = 0, K_FMA, K_EXP, K_CVT, K_DSW, K_DSR, K_MIX, K_GLOAD, K_NKINDS };
static const char* KN[] = {"none", "v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "ds_write_b64", "ds_read_b128", "mix(real staging item)", "global_load_dwordx4"};

template <int KIND>
__device__ __forceinline__ void partner_ops(int k, float& a0, float& a1, float& a2, float& a3, unsigned char* lds, const f32x4* g, int lane) {
  for (int i = 0; i < k; i += 4) {
    if (KIND == K_FMA) {
      asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (KIND == K_EXP) {
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (KIND == K_CVT) {
      asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_pk_f16_f32 %1, %1, %2\n\tv_cvt_pk_f16_f32 %2, %2, %3\n\tv_cvt_pk_f16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (KIND == K_DSW) {
      uint2 v = {__float_as_uint(a0), __float_as_uint(a1)};
      uint2* p = (uint2*)(lds + 65536) + lane;
      p[0] = v; p[64] = v; p[128] = v; p[192] = v;
    } else if (KIND == K_DSR) {
      const f32x4* p = (const f32x4*)(lds + 65536) + lane;
      f32x4 r0 = p[0], r1 = p[64], r2 = p[128], r3 = p[192];
      asm volatile("" :: "v"(r0), "v"(r1), "v"(r2), "v"(r3));
    } else if (KIND == K_MIX) {
      // the per-element staging sequence of the real kernel: 2 fma, exp, add, rcp, mul, cndmask, (cvt_pk + fma_mix) / element
      asm volatile("v_fma_f32 %0, %1, %2, %3\n\tv_fma_f32 %1, %2, %3, %0\n\tv_exp_f32 %2, %1\n\tv_add_f32 %2, 1.0, %2\n\tv_rcp_f32 %2, %2\n\tv_mul_f32 %3, %0, %2\n\t"
                   "v_cndmask_b32 %0, 0, %3, vcc\n\tv_cvt_pk_f16_f32 %1, %0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");
      i += 4;  // 8 instructions per round
    } else if (KIND == K_GLOAD) {
      f32x4 r0 = g[lane * 16 + (i & 1023) * 64], r1 = g[lane * 16 + ((i + 1) & 1023) * 64 + 4096], r2 = g[lane * 16 + ((i + 2) & 1023) * 64 + 8192], r3 = g[lane * 16 + ((i + 3) & 1023) * 64 + 12288];
      asm volatile("" :: "v"(r0), "v"(r1), "v"(r2), "v"(r3));
    }
  }
}

template <int KIND, int NMFMA_WAVES_PER_SIMD, bool PRIO>
__global__ __launch_bounds__(768, 3) void probe(int steps, int k_per_step, int mfma_per_step, unsigned long long* ticks, const f32x4* g, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const bool is_mfma = wave < 4 * NMFMA_WAVES_PER_SIMD;
  const bool is_partner = wave >= 8;
  f32x16 acc0 = {}, acc1 = {};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * lane + i); b[i] = (_Float16)(0.002f * lane - i); }
  float a0 = 0.5f + lane, a1 = 0.25f, a2 = 0.125f, a3 = 0.3f;
  if (is_mfma && PRIO) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    if (is_mfma) {
      for (int m = 0; m < mfma_per_step; m += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
      }
    } else if (is_partner) {
      partner_ops<KIND>(k_per_step, a0, a1, a2, a3, lds, g, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 37) ticks[wave] = t1 - t0;
  if (acc0[0] + acc1[3] + a0 + a1 + a2 + a3 == 1.2345f) sink[threadIdx.x] = acc0[1];
}

template <int KIND, int NM, bool PRIO>
static void run(const char* label, int k, int mfma, unsigned long long* dticks, const f32x4* g, float* sink) {
  const int steps = 200;
  hipFuncSetAttribute((const void*)&probe<KIND, NM, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND, NM, PRIO><<<256, 768, 131072>>>(steps, k, mfma, dticks, g, sink);
  hipEventRecord(e0);
  probe<KIND, NM, PRIO><<<256, 768, 131072>>>(steps, k, mfma, dticks, g, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[12]; hipMemcpy(h, dticks, sizeof(h), hipMemcpyDeviceToHost);
  const double per_step_us = ms * 1e3 / steps;
  printf("%-28s K=%4d mfma=%3d x%d prio=%d: %7.3f us/step  %6.0f ticks/step", label, k, mfma, NM, (int)PRIO, per_step_us, (double)h[0] / steps);
  if (mfma) printf("  -> %.1f ticks/MFMA (%.1f ns)", (double)h[0] / steps / (mfma * NM), per_step_us * 1e3 / (mfma * NM));
  if (k) printf("  | %.1f ns per partner instr", per_step_us * 1e3 / k);
  printf("\n");
}

int main() {
  unsigned long long* dticks; hipMalloc(&dticks, 12 * 8);
  f32x4* g; hipMalloc(&g, 64 << 20); hipMemset(g, 0, 64 << 20);
  float* sink; hipMalloc(&sink, 4096);
  printf("-- MFMA waves alone (1 per SIMD), 108 MFMAs per barrier step\n");
  run<K_NONE, 1, true>("none", 0, 108, dticks, g, sink);
  run<K_NONE, 2, true>("none, 2 MFMA waves / SIMD", 0, 108, dticks, g, sink);
  printf("-- partner alone (no MFMAs)\n");
  run<K_FMA, 1, true>(KN[K_FMA], 432, 0, dticks, g, sink);
  run<K_EXP, 1, true>(KN[K_EXP], 432, 0, dticks, g, sink);
  run<K_CVT, 1, true>(KN[K_CVT], 432, 0, dticks, g, sink);
  run<K_MIX, 1, true>(KN[K_MIX], 432, 0, dticks, g, sink);
  run<K_DSW, 1, true>(KN[K_DSW], 432, 0, dticks, g, sink);
  run<K_DSR, 1, true>(KN[K_DSR], 432, 0, dticks, g, sink);
  run<K_GLOAD, 1, true>(KN[K_GLOAD], 48, 0, dticks, g, sink);
  printf("-- MFMA wave + partner on the same SIMD\n");
  for (int k : {108, 216, 432}) run<K_FMA, 1, true>(KN[K_FMA], k, 108, dticks, g, sink);
  for (int k : {108, 216, 432}) run<K_EXP, 1, true>(KN[K_EXP], k, 108, dticks, g, sink);
  for (int k : {108, 216, 432}) run<K_CVT, 1, true>(KN[K_CVT], k, 108, dticks, g, sink);
  for (int k : {108, 216, 432}) run<K_MIX, 1, true>(KN[K_MIX], k, 108, dticks, g, sink);
  for (int k : {108, 216, 432}) run<K_DSW, 1, true>(KN[K_DSW], k, 108, dticks, g, sink);
  for (int k : {108, 216, 432}) run<K_DSR, 1, true>(KN[K_DSR], k, 108, dticks, g, sink);
  for (int k : {24, 48}) run<K_GLOAD, 1, true>(KN[K_GLOAD], k, 108, dticks, g, sink);
  printf("-- the same without s_setprio on the MFMA waves\n");
  run<K_FMA, 1, false>(KN[K_FMA], 432, 108, dticks, g, sink);
  run<K_MIX, 1, false>(KN[K_MIX], 432, 108, dticks, g, sink);
  printf("-- two MFMA waves per SIMD (54 MFMAs each) + partner\n");
  run<K_MIX, 2, true>(KN[K_MIX], 432, 54, dticks, g, sink);
  run<K_FMA, 2, true>(KN[K_FMA], 432, 54, dticks, g, sink);
  return 0;
}
