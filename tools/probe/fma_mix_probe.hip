// Does v_fma_mixlo/hi_f16 (x * 1.0 - h, h read as fp16) give bit-for-bit fp16(x - float(h)), subnormal results included?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/fma_mix_probe.hip -o tools/probe/fma_mix_probe && tools/probe/fma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* out_mix, unsigned* out_ref, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float x0 = x[2 * i], x1 = x[2 * i + 1];
  h2 h = {(_Float16)x0, (_Float16)x1};
  unsigned h01 = __builtin_bit_cast(unsigned, h), l01;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(l01) : "v"(x0), "v"(x1), "v"(h01));
  h2 lr = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
  out_mix[i] = l01;
  out_ref[i] = __builtin_bit_cast(unsigned, lr);
}
int main() {
  const int n = 1 << 22;
  float* hx = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float m = (float)rand() / RAND_MAX * 2.f - 1.f;
    int e = rand() % 44 - 30;  // 2^-30 .. 2^13
    hx[i] = ldexpf(m, e);
  }
  float* dx; unsigned *da, *db;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 2); hipMalloc(&db, n * 2);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, da, db, n);
  unsigned* ha = (unsigned*)malloc(n * 2); unsigned* hb = (unsigned*)malloc(n * 2);
  hipMemcpy(ha, da, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n / 2; ++i)
    if (ha[i] != hb[i]) { if (bad < 10) printf("x = %g %g: mix %08x ref %08x\n", hx[2 * i], hx[2 * i + 1], ha[i], hb[i]); ++bad; }
  printf("fma_mix probe: %d of %d pairs differ\n", bad, n / 2);
  return 0;
}
