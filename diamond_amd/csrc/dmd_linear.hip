// dmd_linear -- C[M,N] (+)= A[M,K] . W[N,K]^T + bias, optional SiLU, on v_mfma_f32_16x16x4_f32.
//
// nn.Linear call sites of the hot path: the 44 AdaGroupNorm linears of a denoiser forward
// batched into ONE (B,256)x(256,7168) call (blocks.py:39,44 share `cond`, blocks.py:171-177),
// cond_proj (inner_model.py:31-35), LSTM gate GEMMs (actor_critic.py:46,72;
// rew_end_model.py:34,53), heads (actor_critic.py:47-48; rew_end_model.py:35-39).
//
// Both operands are K-contiguous ("NT" GEMM), so MFMA fragments are plain 16-byte global
// loads: lane (i = lane & 15, kg = lane >> 4) reads A[m0 + i][k0 + 4 kg .. +3] and
// W[n0 + i][k0 + 4 kg .. +3]; the four components feed four MFMAs (k-remap, same trick as
// dmd_conv).  W is the MFMA A operand (rows = n), activations the B operand (cols = m), so a
// lane owns 4 consecutive n of one m -> 16-byte stores of C rows.
// Workgroup = 4 waves = 64 (m) x 64 (n) outputs; wave = 32 x 32 (2 x 2 MFMA blocks).
// These GEMMs are < 2 % of the step FLOPs and are L2-resident; no LDS staging.
#include "dmd_common.h"

__global__ __launch_bounds__(256) void linear_mfma_kernel(const dmd_linear_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int m0 = blockIdx.x * 64 + (wave >> 1) * 32;
  const int n0 = blockIdx.y * 64 + (wave & 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* arow[2];
  const float* wrow[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int m = m0 + 16 * b + i;
    m = m < p.M ? m : p.M - 1;  // clamp: duplicates are never stored
    arow[b] = p.A + (size_t)m * p.lda + 4 * kg;
    int n = n0 + 16 * b + i;
    n = n < p.N ? n : p.N - 1;
    wrow[b] = p.W + (size_t)n * p.ldw + 4 * kg;
  }
  // K loop, software-pipelined: the fragments of step k0 + 16 are in flight while the 16 MFMAs of step k0 issue
  // (these GEMMs are L2-resident and latency-bound: without the prefetch every step waits a full L2 round trip)
  f32x4 af[2], wf[2], an[2], wn[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    af[b] = *(const f32x4*)(arow[b]);
    wf[b] = *(const f32x4*)(wrow[b]);
  }
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    const int kn = k0 + 16 < p.K ? k0 + 16 : k0;  // last step: harmless reload
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      an[b] = *(const f32x4*)(arow[b] + kn);
      wn[b] = *(const f32x4*)(wrow[b] + kn);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb][t], af[mb][t], acc[nb][mb], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      af[b] = an[b];
      wf[b] = wn[b];
    }
  }
  // D rows = n (4*kg + r), cols = m (i)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int m = m0 + 16 * mb + i;
      const int n = n0 + 16 * nb + 4 * kg;
      if (m >= p.M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= p.N) continue;
        float v = acc[nb][mb][r];
        if (p.bias) v += p.bias[n + r];
        float* c = p.C + (size_t)m * p.ldc + n + r;
        if (p.accumulate) v += *c;
        if (p.silu) v = dmd_silu(v);
        *c = v;
      }
    }
}

extern "C" int dmd_linear(const dmd_linear_params* p, dmd_stream_t stream) {
  DMD_CHECK_ARG(p && p->A && p->W && p->C, "linear: null");
  DMD_CHECK_ARG(p->M > 0 && p->N > 0 && p->K > 0 && p->K % 16 == 0, "linear: bad M/N/K %d %d %d", p->M, p->N, p->K);
  DMD_CHECK_ARG(p->lda % 4 == 0 && p->ldw % 4 == 0, "linear: lda/ldw must be multiples of 4");
  dim3 grid((p->M + 63) / 64, (p->N + 63) / 64);
  hipLaunchKernelGGL(linear_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p);
  DMD_LAUNCH_CHECK();
  return 0;
}
