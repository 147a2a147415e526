// dmd_pack_jobs -- every kernel-layout copy of a model's parameters in ONE launch.
//
// The kernels read convolution weights in packed layouts (dmd_pack_conv_weight: [CinPad/16][taps][CoutPad][16] fp32;
// dmd_pack_conv_weight_f16x2: [CinPad/16][taps][h|l][2][Cout][8] split fp16 pieces) and, for the data gradient, the packed
// layout of the TRANSPOSED convolution restricted to a slice of input channels:
//     W'[ci - c0][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx]   (autograd of F.conv2d, /root/reference/src/trainer.py:366)
// A training step changes every parameter, so every copy is rebuilt every step: per convolution that was flip + transpose +
// contiguous (+ zero-pad) in torch and two pack launches -- ~640 launches of 3-4 us per denoiser step, 2.7 ms of a 16 ms
// step once the step itself is a replayed hipGraph.  Here the copies are described once by a job table in device memory
// and rebuilt by one launch: grid (element blocks, jobs).
#include "dmd_common.h"

__global__ __launch_bounds__(256) void pack_jobs_kernel(const dmd_pack_job* __restrict__ jobs) {
  const dmd_pack_job j = jobs[blockIdx.y];
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int taps = j.k * j.k;
  // logical weight W'(co, c, tap) of the job: the parameter itself, or the transposed slice
  auto value = [&](int co, int c, int tap) -> float {
    if (j.kind == DMD_PACK_BIAS) return 0.f;
    if (!j.transposed) return (co < j.Cout && c < j.Cin) ? j.src[((size_t)co * j.Cin + c) * taps + tap] : 0.f;
    // W'[co][c][tap] = W[c][c0 + co][taps - 1 - tap]; rows c >= Cout of the parameter are zero padding
    return (co < j.c1 - j.c0 && c < j.Cout) ? j.src[((size_t)c * j.Cin + (j.c0 + co)) * taps + (taps - 1 - tap)] : 0.f;
  };
  if (j.kind == DMD_PACK_F32) {
    const size_t total = (size_t)(j.CinPad / 16) * taps * j.CoutPad * 16;
    if (idx >= total) return;
    const int ci = idx % 16;
    const int co = (idx / 16) % j.CoutPad;
    const int tap = (idx / (16 * (size_t)j.CoutPad)) % taps;
    const int chunk = idx / (16 * (size_t)j.CoutPad * taps);
    ((float*)j.dst)[idx] = value(co, chunk * 16 + ci, tap);
  } else if (j.kind == DMD_PACK_F16X2) {
    const int Cout = j.CoutPad;  // (the split layout has no separate padding: CoutPad == its Cout, 32 or 64)
    const size_t total = (size_t)(j.CinPad / 16) * taps * Cout * 16;
    if (idx >= total) return;
    const int e = idx % 8;
    const int co = (idx / 8) % Cout;
    const int g = (idx / (8 * (size_t)Cout)) % 2;
    const int tap = (idx / (8 * (size_t)Cout * 2)) % taps;
    const int chunk = idx / (8 * (size_t)Cout * 2 * taps);
    const float v = value(co, chunk * 16 + g * 8 + e, tap);
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const size_t base = ((((size_t)chunk * taps + tap) * 2 + 0) * 2 + g) * ((size_t)Cout * 8) + (size_t)co * 8 + e;
    ((_Float16*)j.dst)[base] = h;
    ((_Float16*)j.dst)[base + 2 * (size_t)Cout * 8] = l;
  } else {  // DMD_PACK_BIAS: [CoutPad] <- bias[0..Cout), zero beyond
    if (idx >= (size_t)j.CoutPad) return;
    ((float*)j.dst)[idx] = idx < (size_t)j.Cout ? j.src[idx] : 0.f;
  }
}

extern "C" int dmd_pack_jobs(const dmd_pack_job* jobs_device, int njobs, int64_t max_elems, dmd_stream_t stream) {
  DMD_CHECK_ARG(jobs_device && njobs > 0 && max_elems > 0, "pack_jobs: %d jobs, %lld elements", njobs, (long long)max_elems);
  DMD_CHECK_ARG(njobs <= 65535, "pack_jobs: more than 65535 jobs in one table");
  hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)((max_elems + 255) / 256), (unsigned)njobs), dim3(256), 0, (hipStream_t)stream,
                     jobs_device);
  DMD_LAUNCH_CHECK();
  return 0;
}


// dmd_checksums -- the audit of the packed copies (ABI v9).  A copy is rebuilt when its parameter's `Tensor._version` / storage
// pointer changes or an optimizer hook says so; a write that shows in neither (`p.data.copy_`, a collective on `.data`) would
// leave the kernels running on the old weights, silently.  So the host records, right behind every (re)build, an exact
// fingerprint of the SOURCE it was built from, and every so often compares it with the live parameter: one launch over all
// sources.  Fingerprint of a source = DMD_CHECKSUM_PARTS 64-bit sums of its 32-bit words taken as signed integers, part b over the
// words w with (w / 256) % PARTS == b: exact, no atomics, the same value for the same bits whatever the launch looks like.
__global__ __launch_bounds__(256) void checksums_kernel(const dmd_checksum_job* __restrict__ jobs, long long* __restrict__ out) {
  __shared__ long long red[256];
  const dmd_checksum_job j = jobs[blockIdx.y];
  const int32_t* w = (const int32_t*)j.src;
  long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.words; i += (int64_t)DMD_CHECKSUM_PARTS * 256) s += (long long)w[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)blockIdx.y * DMD_CHECKSUM_PARTS + blockIdx.x] = red[0];
}

extern "C" int dmd_checksums(const dmd_checksum_job* jobs_device, int njobs, long long* out_device, dmd_stream_t stream) {
  DMD_CHECK_ARG(jobs_device && out_device && njobs > 0 && njobs <= 65535, "checksums: %d jobs", njobs);
  hipLaunchKernelGGL(checksums_kernel, dim3(DMD_CHECKSUM_PARTS, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, out_device);
  DMD_LAUNCH_CHECK();
  return 0;
}
