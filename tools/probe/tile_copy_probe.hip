// Hardware probe (development aid): HBM copy bandwidth of an NHWC fp32 tensor (N=256, 64x64, 64 channels = 268 MB)
// as a function of the ACCESS PATTERN: linear float4 streaming vs one workgroup per 16x16-pixel tile (rows of 4 KiB
// at a 16 KiB stride) vs one workgroup per 4x64-pixel strip (64 KiB contiguous).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void copy_linear(const f4* a, f4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// tile = TH x TW pixels of one image, 64 channels (16 float4 per pixel); one WG (256 threads) per tile
template <int TH, int TW>
__global__ void copy_tiled(const f4* a, f4* b, int H, int W) {
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int t = blockIdx.x;
  const int n = t / (tiles_x * tiles_y), r = t % (tiles_x * tiles_y);
  const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
  for (int i = threadIdx.x; i < TH * TW * 16; i += 256) {
    const int q = i & 15, p = i >> 4;
    const int py = p / TW, px = p % TW;
    const size_t off = (((size_t)n * H + y0 + py) * W + x0 + px) * 16 + q;
    b[off] = a[off];
  }
}
// read-only variants (sum into a sink) to separate read from write behaviour
template <int TH, int TW>
__global__ void read_tiled(const f4* a, float* sink, int H, int W) {
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int t = blockIdx.x;
  const int n = t / (tiles_x * tiles_y), r = t % (tiles_x * tiles_y);
  const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
  f4 s = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < TH * TW * 16; i += 256) {
    const int q = i & 15, p = i >> 4;
    const int py = p / TW, px = p % TW;
    s += a[(((size_t)n * H + y0 + py) * W + x0 + px) * 16 + q];
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

template <class F>
static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main() {
  const int N = 256, H = 64, W = 64;
  const size_t n4 = (size_t)N * H * W * 16;  // float4 count
  f4 *a, *b; float* sink;
  hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&sink, 64);
  hipMemset(a, 1, n4 * 16);
  const double bytes = (double)n4 * 16;
  float ms;
  ms = timeit([&] { copy_linear<<<4096, 256>>>(a, b, n4); });
  printf("copy linear            : %7.1f us  %6.0f GB/s (read+write)\n", ms * 1e3, 2 * bytes / ms / 1e6);
  ms = timeit([&] { copy_tiled<16, 16><<<N * 16, 256>>>(a, b, H, W); });
  printf("copy 16x16 tiles       : %7.1f us  %6.0f GB/s\n", ms * 1e3, 2 * bytes / ms / 1e6);
  ms = timeit([&] { copy_tiled<4, 64><<<N * 16, 256>>>(a, b, H, W); });
  printf("copy 4x64 strips       : %7.1f us  %6.0f GB/s\n", ms * 1e3, 2 * bytes / ms / 1e6);
  ms = timeit([&] { copy_tiled<8, 16><<<N * 32, 256>>>(a, b, H, W); });
  printf("copy 8x16 tiles        : %7.1f us  %6.0f GB/s\n", ms * 1e3, 2 * bytes / ms / 1e6);
  ms = timeit([&] { read_tiled<16, 16><<<N * 16, 256>>>(a, sink, H, W); });
  printf("read 16x16 tiles       : %7.1f us  %6.0f GB/s (read only)\n", ms * 1e3, bytes / ms / 1e6);
  ms = timeit([&] { read_tiled<4, 64><<<N * 16, 256>>>(a, sink, H, W); });
  printf("read 4x64 strips       : %7.1f us  %6.0f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  return 0;
}
