// Hardware probe (development aid): HBM read bandwidth of (a) register loads, 16 x 16 B per lane in flight per wave,
// vs (b) LDS-DMA (global_load_lds) into a wave-private ring, two 8 KiB pieces ahead -- with and without a concurrent
// write stream of half the bytes (the 1x1 skip projection's 2 : 1 mix).
// Build: hipcc --offload-arch=gfx950 -O3 read_bw_probe.hip -o read_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// piece = 8 KiB = 512 units of 16 B; a wave walks pieces [p0, p1)
template <bool WRITE>
__global__ __launch_bounds__(256) void reg_read(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int npieces, float* sink) {
  const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
  const int per = (npieces + nw - 1) / nw, p0 = gw * per, p1 = min(npieces, p0 + per);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int p = p0; p + 1 < p1; p += 2) {  // 2 pieces = 16 loads per lane in flight
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = src[(size_t)p * 512 + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (WRITE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[(size_t)(p >> 1) * 512 + i * 64 + lane] = v[i] + v[i + 8];
    }
  }
  if (s[0] + s[1] + s[2] + s[3] == 1.2345f) sink[0] = 1.f;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void dma_read(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int npieces, float* sink) {
  extern __shared__ f32x4 ring[];  // [4 waves][3][512]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
  const int per = (npieces + nw - 1) / nw, p0 = gw * per, p1 = min(npieces, p0 + per);
  f32x4* my = ring + wave * 3 * 512;
  auto issue = [&](int p, int slot) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 512 + i * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(my + slot * 512 + i * 64), 16, 0, 0);
  };
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const int n = p1 - p0;
  if (n <= 0) return;
  for (int t = 0; t < 3 && t < n; ++t) issue(p0 + t, t);
  for (int t = 0; t < n; ++t) {
    // ops issued after DMA(t): [stores(t-2)] DMA(t+1) [stores(t-1)] DMA(t+2); a short tail waits for everything
    if (t + 2 < n) {
      if (WRITE) { if (t == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else if (t == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int slot = t % 3;
    f32x4 a, b;
    // raw LDS reads (the compiler must not tie them to the DMA it cannot see through)
    asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n s_waitcnt lgkmcnt(0)"
                 : "=v"(a), "=v"(b) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) f32x4*)(my + slot * 512 + lane)) : "memory");
    s += a + b;
    if (WRITE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(size_t)(p0 + t) * 256 + i * 64 + lane] = a + (float)i;
    }
    if (t + 3 < n) issue(p0 + t + 3, slot);
  }
  if (s[0] + s[1] + s[2] + s[3] == 1.2345f) sink[0] = 1.f;
}

int main() {
  const size_t bytes = 512ull << 20;  // 512 MiB read
  const int npieces = (int)(bytes / 8192);
  f32x4 *src, *dst; float* sink;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes / 2); hipMalloc(&sink, 4);
  hipMemset(src, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, const char* name, double total_bytes) {
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s %8.1f us  %6.2f TB/s (%s)\n", name, ms * 1e3, total_bytes / ms / 1e9, hipGetErrorString(hipGetLastError()));
  };
  for (int wgs : {256, 512, 1024}) {
    printf("-- %d workgroups of 256\n", wgs);
    time([&] { reg_read<false><<<wgs, 256>>>(src, dst, npieces, sink); }, "reg loads, read only", (double)bytes);
    time([&] { reg_read<true><<<wgs, 256>>>(src, dst, npieces, sink); }, "reg loads, read + write/2", bytes * 1.5);
    hipFuncSetAttribute((const void*)dma_read<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 8192);
    hipFuncSetAttribute((const void*)dma_read<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 8192);
    time([&] { dma_read<false><<<wgs, 256, 4 * 3 * 8192>>>(src, dst, npieces, sink); }, "LDS-DMA ring, read only", (double)bytes);
    time([&] { dma_read<true><<<wgs, 256, 4 * 3 * 8192>>>(src, dst, npieces, sink); }, "LDS-DMA ring, read + write/2", bytes * 1.5);
  }
  return 0;
}
