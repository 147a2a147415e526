// Does gfx950 keep a transcendental's SOURCE register safe from the next instruction of the same wave, and its RESULT from an early reader,
// when several waves of a SIMD issue transcendentals?  (round 6: wgrad_ps_kernel with the SiLU as straight-line code gave run-to-run
// different gradients, only with two staging waves per SIMD, profiles/r06n_wgrad_race.txt; hipcc had emitted
//     v_add_f32 v92, 1.0, v92 / v_rcp_f32 v98, v92 / v_add_f32 v92, 1.0, v93 / v_rcp_f32 v99, v92 / v_cvt_pk_f16_f32 v92, v84, v85 .)
// Every variant computes r0 = rcp(1 + a), r1 = rcp(1 + b) twice: once with `s_nop 7` x 2 behind every instruction (the reference: nothing
// can overlap) and once as the pattern under test; bits compared, mismatches counted.
//   WAR   : the source of a v_rcp overwritten by the next instruction (the compiler's sequence above)
//   WAW   : the destination of a v_rcp written again by the next (plain) instruction, then the rcp's value is expected to LOSE
//   RAW0/1: v_exp's result read by the next instruction / one independent instruction later (the ISA manual asks for 1 wait state)
//   SILU  : the whole compiled sequence (mul, exp, add, rcp, mul for two values, registers reused as hipcc reused them)
// waves per SIMD = workgroups per CU (256-thread workgroups, 256 CUs x k workgroups).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/trans_war_probe.hip -o tools/probe/trans_war_probe && tools/probe/trans_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define NOPS "s_nop 7\n\ts_nop 7\n\t"

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float urand(unsigned& s) { return (float)(lcg(s) >> 8) * (4.0f / 16777216.0f) + 0.01f; }  // (0.01, 4.01)

template <int V>
__global__ __launch_bounds__(256) void probe(unsigned long long* bad, int iters) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  unsigned long long n = 0;
  for (int i = 0; i < iters; ++i) {
    const float a = urand(s), b = urand(s);
    float r0, r1, q0, q1, t, u;
    // reference: every instruction drained
    asm volatile("v_add_f32 %2, 1.0, %3\n\t" NOPS "v_rcp_f32 %0, %2\n\t" NOPS "v_add_f32 %2, 1.0, %4\n\t" NOPS "v_rcp_f32 %1, %2\n\t" NOPS
                 : "=&v"(q0), "=&v"(q1), "=&v"(t) : "v"(a), "v"(b));
    if (V == 0) {  // WAR
      asm volatile("v_add_f32 %2, 1.0, %3\n\t"
                   "v_rcp_f32 %0, %2\n\t"
                   "v_add_f32 %2, 1.0, %4\n\t"
                   "v_rcp_f32 %1, %2\n\t"
                   "v_cvt_pk_f16_f32 %2, %3, %4\n\t" NOPS
                   : "=&v"(r0), "=&v"(r1), "=&v"(t) : "v"(a), "v"(b));
    } else if (V == 1) {  // WAW: the plain move must win (program order)
      asm volatile("v_add_f32 %2, 1.0, %3\n\t" NOPS
                   "v_rcp_f32 %0, %2\n\t"
                   "v_mov_b32 %0, %3\n\t" NOPS
                   "v_add_f32 %2, 1.0, %4\n\t" NOPS
                   "v_rcp_f32 %1, %2\n\t"
                   "v_mov_b32 %1, %4\n\t" NOPS
                   : "=&v"(r0), "=&v"(r1), "=&v"(t) : "v"(a), "v"(b));
      q0 = a;
      q1 = b;
    } else if (V == 2 || V == 3) {  // RAW on a v_exp result, 0 / 1 instruction in between: r = rcp(1 + exp2(-a)) against the drained form
      asm volatile("v_exp_f32 %2, -%3\n\t" NOPS "v_add_f32 %2, 1.0, %2\n\t" NOPS "v_rcp_f32 %0, %2\n\t" NOPS
                   "v_exp_f32 %2, -%4\n\t" NOPS "v_add_f32 %2, 1.0, %2\n\t" NOPS "v_rcp_f32 %1, %2\n\t" NOPS
                   : "=&v"(q0), "=&v"(q1), "=&v"(t) : "v"(a), "v"(b));
      if (V == 2)
        asm volatile("v_exp_f32 %2, -%4\n\t"
                     "v_add_f32 %2, 1.0, %2\n\t"
                     "v_rcp_f32 %0, %2\n\t"
                     "v_exp_f32 %3, -%5\n\t"
                     "v_add_f32 %3, 1.0, %3\n\t"
                     "v_rcp_f32 %1, %3\n\t" NOPS
                     : "=&v"(r0), "=&v"(r1), "=&v"(t), "=&v"(u) : "v"(a), "v"(b));
      else
        asm volatile("v_exp_f32 %2, -%4\n\t"
                     "v_exp_f32 %3, -%5\n\t"
                     "v_add_f32 %2, 1.0, %2\n\t"
                     "v_add_f32 %3, 1.0, %3\n\t"
                     "v_rcp_f32 %0, %2\n\t"
                     "v_rcp_f32 %1, %3\n\t" NOPS
                     : "=&v"(r0), "=&v"(r1), "=&v"(t), "=&v"(u) : "v"(a), "v"(b));
    } else {  // SILU: a * sigmoid(a), b * sigmoid(b) with hipcc's register reuse (t is the v92 of the listing)
      asm volatile("v_mul_f32 %2, 0xbfb8aa3b, %3\n\t" NOPS "v_exp_f32 %2, %2\n\t" NOPS "v_add_f32 %2, 1.0, %2\n\t" NOPS "v_rcp_f32 %0, %2\n\t" NOPS
                   "v_mul_f32 %0, %0, %3\n\t" NOPS
                   "v_mul_f32 %2, 0xbfb8aa3b, %4\n\t" NOPS "v_exp_f32 %2, %2\n\t" NOPS "v_add_f32 %2, 1.0, %2\n\t" NOPS "v_rcp_f32 %1, %2\n\t" NOPS
                   "v_mul_f32 %1, %1, %4\n\t" NOPS
                   : "=&v"(q0), "=&v"(q1), "=&v"(t) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %2, 0xbfb8aa3b, %4\n\t"
                   "v_exp_f32 %2, %2\n\t"
                   "v_mul_f32 %3, 0xbfb8aa3b, %5\n\t"
                   "v_exp_f32 %3, %3\n\t"
                   "v_add_f32 %2, 1.0, %2\n\t"
                   "v_rcp_f32 %0, %2\n\t"
                   "v_add_f32 %2, 1.0, %3\n\t"
                   "v_rcp_f32 %1, %2\n\t"
                   "v_cvt_pk_f16_f32 %2, %4, %5\n\t"
                   "s_nop 1\n\t"
                   "v_mul_f32 %0, %0, %4\n\t"
                   "v_mul_f32 %1, %1, %5\n\t" NOPS
                   : "=&v"(r0), "=&v"(r1), "=&v"(t), "=&v"(u) : "v"(a), "v"(b));
    }
    n += (__float_as_uint(r0) != __float_as_uint(q0)) + (__float_as_uint(r1) != __float_as_uint(q1));
  }
  if (n) atomicAdd(bad, n);
}

template <int V>
static void run(const char* name, unsigned long long* bad) {
  for (int k : {1, 2, 4, 8}) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(probe<V>, dim3(256 * k), dim3(256), 0, 0, bad, 20000);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-5s %d wave(s) per SIMD: %llu mismatches of %llu\n", name, k, h, 2ull * 20000 * 256 * 256 * k);
  }
}

int main() {
  unsigned long long* bad;
  hipMalloc(&bad, 8);
  run<0>("WAR", bad);
  run<1>("WAW", bad);
  run<2>("RAW0", bad);
  run<3>("RAW1", bad);
  run<4>("SILU", bad);
  return 0;
}
