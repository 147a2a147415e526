// Hardware probe (development aid): semantics of __builtin_amdgcn_global_load_lds with 16-byte elements on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const u32x4* src, u32x4* out) {
  __shared__ u32x4 lds[512];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 512; i += 256) lds[i] = (u32x4){0xdead, 0xdead, 0xdead, 0xdead};
  __syncthreads();
  // each wave: LDS destination base is wave-uniform; lane l is expected to land at base + 16 * l
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid),
                                   (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 256 + tid),
                                   (__attribute__((address_space(3))) void*)(lds + 256 + wave * 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) etc.
  __syncthreads();
  for (int i = tid; i < 512; i += 256) out[i] = lds[i];
}

int main() {
  u32x4 h[512], o[512];
  for (int i = 0; i < 512; ++i) h[i] = (u32x4){(unsigned)i, (unsigned)(i * 3 + 1), (unsigned)(i ^ 0x55), 7u};
  u32x4 *d, *dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 256>>>(d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; ++i) if (o[i][0] != h[i][0] || o[i][1] != h[i][1] || o[i][2] != h[i][2] || o[i][3] != h[i][3]) { if (bad < 8) printf("mismatch at %d: got %u %u %u %u\n", i, o[i][0], o[i][1], o[i][2], o[i][3]); ++bad; }
  printf("lds dma probe: %d mismatches of 512 (%s)\n", bad, hipGetErrorString(hipGetLastError()));
  return 0;
}
