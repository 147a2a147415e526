// Which clock does the chip run, by instruction mix?  (round 4, HISTORY.md §3 "clock")  A counter-free measurement:
// a kernel whose instruction count is known exactly runs for seconds; achieved rate / (work per cycle) = shader clock.
//   mode mfma : every wave issues independent v_mfma_f32_32x32x16_f16 (4 accumulators) back to back: 32 cycles each per SIMD
//               (8 passes x 4), i.e. 1024 FLOP / cycle / SIMD  ->  clock = FLOP/s / (1024 x SIMDs)
//   mode valu : every wave issues dependent-free v_fma_f32 (16 chains): one wave64 fma = 4 cycles on the SIMD's 16 lanes
//               -> clock = instructions/s per SIMD x 4     (1 wave per SIMD: nothing else competes for the port)
// Both also stamp s_memtime at the start and end of wave 0 of every workgroup: ticks / event time = what DESIGN §3 calls the
// s_memtime clock.  Operands are random (fill = 1) or zero (fill = 0): the DVFS give-back.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/clock_probe.hip -o tools/probe/clock_probe
//   tools/probe/clock_probe mfma 6 1      (mode, seconds, fill)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_kernel(const h8* __restrict__ in, float* __restrict__ sink, unsigned long long* ticks, int iters) {
  const int lane = threadIdx.x & 63;
  h8 a = in[lane], b = in[64 + lane];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 1.2345e30f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void valu_kernel(const float* __restrict__ in, float* __restrict__ sink, unsigned long long* ticks, int iters) {
  float x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = in[(threadIdx.x + 64 * k) & 1023];
  const float m = -1.0f + in[threadIdx.x & 63] * 1e-7f;  // x <- b - (1 - eps) x: sign flips every instruction (bits keep toggling)
  const float b = in[(threadIdx.x + 7) & 63];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += x[k];
  if (s == 1.2345e30f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "mfma";
  const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
  const int fill = argc > 3 ? atoi(argv[3]) : 1;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, simds = cus * 4;
  const int nwg = cus;  // one 256-thread workgroup per CU = one wave per SIMD
  h8* in;
  hipMalloc(&in, 4096 * 4);
  _Float16* hh = (_Float16*)malloc(4096 * 4);
  srand(1);
  for (int i = 0; i < 2048 * 4 / 2; ++i) hh[i] = fill ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)0.f;
  float* hf = (float*)hh;
  if (!strcmp(mode, "valu"))
    for (int i = 0; i < 1024; ++i) hf[i] = fill ? (rand() % 2001 - 1000) * 1e-3f : 0.f;
  hipMemcpy(in, hh, 4096 * 4, hipMemcpyHostToDevice);
  float* sink;
  hipMalloc(&sink, 4096);
  unsigned long long *dt, *ht = (unsigned long long*)malloc(8 * nwg);
  hipMalloc(&dt, 8 * nwg);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const bool mfma = !strcmp(mode, "mfma");
  const int iters = mfma ? 40000 : 120000;  // ~20-40 ms per launch
  double total = 0;
  int n = 0;
  printf("%s fill=%d: %d CUs, one wave per SIMD, %d iterations per launch\n", mode, fill, cus, iters);
  while (total < seconds) {
    hipEventRecord(e0);
    if (mfma)
      hipLaunchKernelGGL(mfma_kernel, dim3(nwg), dim3(256), 0, 0, in, sink, dt, iters);
    else
      hipLaunchKernelGGL(valu_kernel, dim3(nwg), dim3(256), 0, 0, (const float*)in, sink, dt, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(ht, dt, 8 * nwg, hipMemcpyDeviceToHost);
    double tk = 0;
    for (int i = 0; i < nwg; ++i) tk += (double)ht[i];
    tk /= nwg;
    const double insts = (double)iters * (mfma ? 32 : 64);  // per wave
    const double cyc = insts * (mfma ? 32.0 : 4.0);         // issue cycles per wave = per SIMD
    const double ghz = cyc / (ms * 1e6);
    total += ms * 1e-3;
    if (n % 16 == 0 || total >= seconds)
      printf("t=%6.2fs launch %7.3f ms  throughput-derived clock %.3f GHz%s  s_memtime %.3f ticks/ns (%.2f ticks per instruction)\n", total, ms, ghz,
             mfma ? "" : " (if 4 cycles per fma)", tk / (ms * 1e6), tk / insts);
    if (mfma && (n % 16 == 0 || total >= seconds))
      printf("          = %.1f TFLOP/s dense f16 (1024 FLOP/cycle/SIMD x %d SIMDs)\n", insts * 32768.0 * simds / (ms * 1e-3) / 1e12, simds);
    ++n;
  }
  return 0;
}
