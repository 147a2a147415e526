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
// These GEMMs are < 2 % of the step FLOPs and are L2-resident; no LDS staging.
#include "dmd_common.h"

// SPLITK = false: workgroup = 4 waves = 64 (m) x 64 (n) outputs, wave = 32 x 32 (2 x 2 MFMA blocks) over the whole K.
// SPLITK = true (deep-K GEMMs with few output tiles: the LSTM gate GEMMs, (256, 2048) x (2048, 2048)^T is only 128 tiles
// of 64 x 64 on 256 CUs, each a serial chain of 128 L2 round trips): workgroup = ONE 32 x 32 tile, wave w accumulates
// the K quarter [w K/4, (w + 1) K/4); the four partial tiles are added through LDS in the fixed order
// ((w0 + w1) + w2) + w3 -- deterministic, no atomics -- and wave 0 runs the epilogue.
// K loop: DEPTH fragment sets in flight (these GEMMs are L2-resident and latency-bound: a set is reloaded right after
// the MFMAs that consumed it have issued, DEPTH - 1 steps before it is needed again).
#define LIN_DEPTH 4
template <bool SPLITK>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const dmd_linear_params p) {
  __shared__ f32x4 part[SPLITK ? 3 : 1][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int m0 = SPLITK ? blockIdx.x * 32 : blockIdx.x * 64 + (wave >> 1) * 32;
  const int n0 = SPLITK ? blockIdx.y * 32 : blockIdx.y * 64 + (wave & 1) * 32;
  const int kq = SPLITK ? p.K >> 2 : p.K;  // K range of this wave
  const int kbase = SPLITK ? wave * kq : 0;
  const int nsteps = kq >> 4;
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
    arow[b] = p.A + (size_t)m * p.lda + kbase + 4 * kg;
    int n = n0 + 16 * b + i;
    n = n < p.N ? n : p.N - 1;
    wrow[b] = p.W + (size_t)n * p.ldw + kbase + 4 * kg;
  }
  f32x4 af[LIN_DEPTH][2], wf[LIN_DEPTH][2];
#pragma unroll
  for (int d = 0; d < LIN_DEPTH; ++d) {
    const int ks = (d < nsteps ? d : nsteps - 1) * 16;  // short K: harmless reload of the last step
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      af[d][b] = *(const f32x4*)(arow[b] + ks);
      wf[d][b] = *(const f32x4*)(wrow[b] + ks);
    }
  }
  for (int s0 = 0; s0 < nsteps; s0 += LIN_DEPTH) {
#pragma unroll
    for (int d = 0; d < LIN_DEPTH; ++d) {
      if (s0 + d < nsteps) {  // wave-uniform
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[d][nb][t], af[d][mb][t], acc[nb][mb], 0, 0, 0);
        const int sn = s0 + d + LIN_DEPTH;
        const int ks = (sn < nsteps ? sn : nsteps - 1) * 16;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          af[d][b] = *(const f32x4*)(arow[b] + ks);
          wf[d][b] = *(const f32x4*)(wrow[b] + ks);
        }
      }
    }
  }
  if (SPLITK) {
    if (wave) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) part[wave - 1][nb * 2 + mb][lane] = acc[nb][mb];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[nb][mb] += part[w][nb * 2 + mb][lane];
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
  // the choice depends on K alone -- not on M (the batch), the data or the device: an output row is summed in the same
  // order whatever the batch size it is computed in
  if (p->K >= 512 && p->K % 64 == 0)
    hipLaunchKernelGGL(linear_mfma_kernel<true>, dim3((p->M + 31) / 32, (p->N + 31) / 32), dim3(256), 0, (hipStream_t)stream, *p);
  else
    hipLaunchKernelGGL(linear_mfma_kernel<false>, dim3((p->M + 63) / 64, (p->N + 63) / 64), dim3(256), 0, (hipStream_t)stream, *p);
  DMD_LAUNCH_CHECK();
  return 0;
}
