// What does a transcendental cost on a gfx950 SIMD, and does it share the port with plain VALU work?  (round 4, HISTORY.md §6:
// the attention kernel's bound at head_dim 8 is one exponential per (query, key) pair.)  s_memtime ticks are effective shader
// cycles (profiles/r04_clock.json), so ticks / instruction of a dependency-free stream = issue cycles per wave64 instruction.
//   exp, rcp       : 16 independent chains of v_exp_f32 / v_rcp_f32
//   fma, pkfma     : v_fma_f32 / v_pk_fma_f32 (two fp32 per lane)
//   exp+fma 1:N    : one v_exp_f32 then N v_fma_f32, all independent, same wave: does the fma hide beside the exp?
//   exp|fma        : two waves per SIMD, one issues only exps, the other only fmas: the same question across waves
//   exp+mfma       : one 16x16x32 f16 MFMA (8 passes) per K exps in the same wave
// One wave per SIMD unless stated.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/trans_probe.hip -o tools/probe/trans_probe && tools/probe/trans_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define EXP(k) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]))
#define RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]))
#define FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(m), "v"(b))
#define PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(z[k]) : "v"(m2), "v"(b2))
// one-instruction streams (16 independent destinations), the vector instructions the kernels' staging / softmax code is made of
#define MIXLO(k) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(x[k]) : "v"(y[k]), "v"(y[(k + 1) & 15]))
#define MIXHI(k) asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(x[k]) : "v"(y[k]), "v"(y[(k + 1) & 15]))
#define CVTPK(k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[k]) : "v"(y[k]), "v"(y[(k + 1) & 15]))
#define CVTF16(k) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(x[k]) : "v"(y[k]))
#define CVTF32(k) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[k]) : "v"(y[k]))
#define MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y[k]), "v"(y[(k + 1) & 15]))
#define PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(z[k]) : "v"(m2))
#define PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(z[k]) : "v"(m2))
#define ANDB(k) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x[k]) : "v"(y[k]), "v"(y[(k + 1) & 15]))
#define PERM(k) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[k]) : "v"(y[k]), "v"(y[(k + 1) & 15]), "v"(y[(k + 2) & 15]))
#define RSQ(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[k]))
#define MULF(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y[k]) : "v"(m))
#define ADDF(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(y[k]) : "v"(m))
#define PKFMAF16(k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(y[k]) : "v"(m), "v"(b))
#define R16(OP) _Pragma("unroll") for (int k = 0; k < 16; ++k) OP(k);
#define R8x2(OP) _Pragma("unroll") for (int k = 0; k < 8; ++k) OP(k); _Pragma("unroll") for (int k = 0; k < 8; ++k) OP(k);

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ sink, unsigned long long* ticks, int iters) {
  float x[16], y[16];
  f32x2 z[8];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    x[k] = in[(threadIdx.x + 64 * k) & 1023] * 1e-3f;  // exp2 of a small number stays near 1: no overflow, bits keep toggling
    y[k] = in[(threadIdx.x + 64 * k + 5) & 1023];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = f32x2{y[2 * k], y[2 * k + 1]};
  const float m = -1.0f + in[threadIdx.x & 63] * 1e-7f, b = in[(threadIdx.x + 7) & 63];
  const f32x2 m2 = {m, m}, b2 = {b, b};
  const h8 ha = *(const h8*)(in + 8 * (threadIdx.x & 63)), hb = *(const h8*)(in + 512 + 8 * (threadIdx.x & 63));
  f32x4 acc0 = {}, acc1 = {};
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) RCP(k);
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) FMA(k);
    } else if (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) PKFMA(k);
#pragma unroll
      for (int k = 0; k < 8; ++k) PKFMA(k);
    } else if (MODE == 4) {  // 1 : 1
#pragma unroll
      for (int k = 0; k < 16; ++k) { EXP(k); FMA(k); }
    } else if (MODE == 5) {  // 1 : 2
#pragma unroll
      for (int k = 0; k < 16; ++k) { EXP(k); FMA(k); FMA((k + 8) & 15); }
    } else if (MODE == 6) {  // 1 : 4
#pragma unroll
      for (int k = 0; k < 16; ++k) { EXP(k); FMA(k); FMA((k + 4) & 15); FMA((k + 8) & 15); FMA((k + 12) & 15); }
    } else if (MODE == 7) {  // waves 0-3 exps, waves 4-7 fmas (same SIMDs)
      if (wave < 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) EXP(k);
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) FMA(k);
#pragma unroll
        for (int k = 0; k < 16; ++k) FMA(k);
#pragma unroll
        for (int k = 0; k < 16; ++k) FMA(k);
#pragma unroll
        for (int k = 0; k < 16; ++k) FMA(k);
      }
    } else if (MODE == 8) {  // 4 MFMAs (16x16x32 f16: 16 cycles each) + 16 exps
#pragma unroll
      for (int k = 0; k < 16; k += 8) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc0, 0, 0, 0);
        EXP(k); EXP(k + 1); EXP(k + 2); EXP(k + 3);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc1, 0, 0, 0);
        EXP(k + 4); EXP(k + 5); EXP(k + 6); EXP(k + 7);
      }
    } else if (MODE == 9) {  // the MFMAs alone
#pragma unroll
      for (int k = 0; k < 16; k += 8) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc1, 0, 0, 0);
      }
    } else if (MODE == 10) {  // both waves of a SIMD issue exps
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
    } else if (MODE == 11) { R16(MIXLO)
    } else if (MODE == 12) { R16(MIXHI)
    } else if (MODE == 13) { R16(CVTPK)
    } else if (MODE == 14) { R16(CVTF16)
    } else if (MODE == 15) { R16(CVTF32)
    } else if (MODE == 16) { R16(MAX3)
    } else if (MODE == 17) { R8x2(PKADD)
    } else if (MODE == 18) { R8x2(PKMUL)
    } else if (MODE == 19) { R16(ANDB)
    } else if (MODE == 20) { R16(PERM)
    } else if (MODE == 21) { R16(RSQ)
    } else if (MODE == 22) { R16(MULF)
    } else if (MODE == 23) { R16(ADDF)
    } else if (MODE == 24) { R16(PKFMAF16)
    } else if (MODE == 25) {  // the attention kernel's split of 8 weights: 4 cvt_pk + 4 (mixlo, mixhi) pairs on the fresh h words
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[k]) : "v"(y[k]), "v"(y[k + 1]));
        asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                     : "=&v"(x[k + 1]) : "v"(y[k]), "v"(y[k + 1]), "v"(x[k]));
      }
#pragma unroll
      for (int k = 8; k < 16; k += 2) {
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[k]) : "v"(y[k]), "v"(y[k + 1]));
        asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                     : "=&v"(x[k + 1]) : "v"(y[k]), "v"(y[k + 1]), "v"(x[k]));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = acc0[0] + acc1[1] + acc0[2] + acc1[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) s += x[k] + y[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += z[k][0] + z[k][1];
  if (s == 1.2345e30f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

struct Case {
  const char* name;
  int mode, threads, trans, valu, mfma;  // instructions per iteration and wave (of the wave kind that issues them)
};

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int nwg = prop.multiProcessorCount;
  float* in;
  hipMalloc(&in, 4096 * 4);
  float* hf = (float*)malloc(4096 * 4);
  srand(1);
  for (int i = 0; i < 4096; ++i) hf[i] = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(in, hf, 4096 * 4, hipMemcpyHostToDevice);
  float* sink;
  hipMalloc(&sink, 4096);
  unsigned long long *dt, *ht = (unsigned long long*)malloc(64 * nwg);
  hipMalloc(&dt, 64 * nwg);
  const Case cases[] = {
      {"v_exp_f32 x16", 0, 256, 16, 0, 0},
      {"v_rcp_f32 x16", 1, 256, 16, 0, 0},
      {"v_fma_f32 x16", 2, 256, 0, 16, 0},
      {"v_pk_fma_f32 x16", 3, 256, 0, 16, 0},
      {"exp + fma 1:1 (same wave)", 4, 256, 16, 16, 0},
      {"exp + fma 1:2 (same wave)", 5, 256, 16, 32, 0},
      {"exp + fma 1:4 (same wave)", 6, 256, 16, 64, 0},
      {"16 exps in waves 0-3 | 64 fmas in waves 4-7", 7, 512, 16, 64, 0},
      {"4 mfma 16x16x32 + 16 exps (same wave)", 8, 256, 16, 0, 4},
      {"4 mfma 16x16x32 alone", 9, 256, 0, 0, 4},
      {"exps in both waves of a SIMD", 10, 512, 16, 0, 0},
      {"v_fma_mixlo_f16 x16", 11, 256, 0, 16, 0},
      {"v_fma_mixhi_f16 x16", 12, 256, 0, 16, 0},
      {"v_cvt_pk_f16_f32 x16", 13, 256, 0, 16, 0},
      {"v_cvt_f16_f32 x16", 14, 256, 0, 16, 0},
      {"v_cvt_f32_f16 x16", 15, 256, 0, 16, 0},
      {"v_max3_f32 x16", 16, 256, 0, 16, 0},
      {"v_pk_add_f32 x16", 17, 256, 0, 16, 0},
      {"v_pk_mul_f32 x16", 18, 256, 0, 16, 0},
      {"v_and_b32 x16", 19, 256, 0, 16, 0},
      {"v_perm_b32 x16", 20, 256, 0, 16, 0},
      {"v_rsq_f32 x16", 21, 256, 16, 0, 0},
      {"v_mul_f32 x16", 22, 256, 0, 16, 0},
      {"v_add_f32 x16", 23, 256, 0, 16, 0},
      {"v_pk_fma_f16 x16", 24, 256, 0, 16, 0},
      {"8 x (cvt_pk; mixlo; mixhi) = split of 16 values", 25, 256, 0, 24, 0},
  };
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (const Case& c : cases) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
#define L(M) case M: hipLaunchKernelGGL(probe<M>, dim3(nwg), dim3(c.threads), 0, 0, (const float*)in, sink, dt, iters); break;
      switch (c.mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(ht, dt, 64 * nwg, hipMemcpyDeviceToHost);
    const int waves = c.threads / 64;
    double lo = 0, hi = 0;  // mean ticks of waves 0-3 and of waves 4-7
    for (int i = 0; i < nwg; ++i)
      for (int w = 0; w < waves; ++w) (w < 4 ? lo : hi) += (double)ht[i * 8 + w];
    lo /= nwg * 4.0;
    hi /= nwg * 4.0;
    printf("%-46s %7.3f ms  %8.1f ticks / iteration (waves 0-3)", c.name, ms, lo / iters);
    if (waves > 4) printf("  %8.1f (waves 4-7)", hi / iters);
    printf("   [%d trans, %d valu, %d mfma per iteration]  %.3f ticks/ns\n", c.trans, c.valu, c.mfma, lo / (ms * 1e6));
  }
  return 0;
}
