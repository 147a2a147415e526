// Hardware probe (development aid): does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs, and what is the
// operand/result lane layout?  Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_probe.hip -o mfma_f16_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const _Float16* A, const _Float16* B, float* D) {  // A[32][16], B[16][32] row-major, D[32][32]
  const int l = threadIdx.x;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)];
  }
  f16v c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = c[r];
  }
}

int main() {
  _Float16 hA[32 * 16], hB[16 * 32];
  float hD[32 * 32], ref[32 * 32];
  // test 1: random-ish asymmetric values -> layout check
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (_Float16)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (_Float16)((k * 5 + j * 2) % 13 - 6);
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float s = 0; for (int k = 0; k < 16; ++k) s += (float)hA[i * 16 + k] * (float)hB[k * 32 + j];
    ref[i * 32 + j] = s; if (s != hD[i * 32 + j]) ++bad;
  }
  printf("layout check: %d mismatches of 1024\n", bad);
  // test 2: subnormal A (2^-20) x 1.0, subnormal x subnormal-ish, subnormal B
  for (int i = 0; i < 32 * 16; ++i) hA[i] = (_Float16)0.f;
  for (int i = 0; i < 16 * 32; ++i) hB[i] = (_Float16)0.f;
  const float sub = 9.5367431640625e-07f;  // 2^-20, subnormal in fp16 (min normal 2^-14)
  hA[0 * 16 + 0] = (_Float16)sub; hB[0 * 32 + 0] = (_Float16)1.0f;       // D[0][0] = 2^-20 if not flushed
  hA[1 * 16 + 0] = (_Float16)1.0f; hB[0 * 32 + 1] = (_Float16)sub;       // D[1][1] = 2^-20 (subnormal B)
  hA[2 * 16 + 0] = (_Float16)sub; hB[0 * 32 + 2] = (_Float16)1024.0f;    // D[2][2] = 2^-10
  hA[3 * 16 + 0] = (_Float16)5.9604644775390625e-08f; hB[0 * 32 + 3] = (_Float16)1.0f;  // smallest subnormal 2^-24
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  printf("subnormal A*1: %g (want %g)\n", hD[0], sub);
  printf("1*subnormal B: %g (want %g)\n", hD[1 * 32 + 1], sub);
  printf("subnormal A*1024: %g (want %g)\n", hD[2 * 32 + 2], sub * 1024);
  printf("min subnormal A*1: %g (want %g)\n", hD[3 * 32 + 3], 5.9604644775390625e-08);
  // test 3: accumulation precision: sum of 16 products with wide dynamic range in one MFMA
  for (int i = 0; i < 32 * 16; ++i) hA[i] = (_Float16)0.f;
  for (int i = 0; i < 16 * 32; ++i) hB[i] = (_Float16)0.f;
  hA[0] = (_Float16)2048.0f; hB[0] = (_Float16)2048.0f;                       // 2^22
  for (int k = 1; k < 16; ++k) { hA[k] = (_Float16)1.0f; hB[k * 32] = (_Float16)0.25f; }  // 15 * 0.25 = 3.75
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  printf("2^22 + 15*0.25 = %.3f (exact 4194307.75; fp32 chain gives 4194307.5/4194308)\n", hD[0]);
  return 0;
}
