// Backward kernels of the actor-critic conv encoder (autograd of models/actor_critic.py:101-113,
// models/blocks.py:116-123: Conv3x3 -> 4 x [skip(x) + Conv3x3(SiLU(GroupNorm(x))), MaxPool2]).
//
// The reference gets these from ATen autograd (loss.backward(), trainer.py:366).  Here:
//   * dgrad  = dmd_conv2d on the flipped/transposed weight (host repacks it) -- no new kernel;
//   * wgrad  = dmd_conv2d_wgrad below: dW[co][ci][tap] = sum_pixels dy[pix][co] * a[pix + tap][ci]
//     as a GEMM whose contraction runs over PIXELS, on v_mfma_f32_16x16x4_f32 (exact fp32);
//     the conv input a = SiLU(GroupNorm(x)) is RECOMPUTED from x + its statistics while staging
//     (nothing but x and the pooling argmax is saved by the forward);
//   * GroupNorm+SiLU backward = two passes (group reductions, then elementwise apply);
//   * MaxPool backward = scatter by the saved argmax.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "dmd_common.h"

// ------------------------------------------------------------------------------------------------
// max pool 2x2 backward: dx[n, 2oy + k/2, 2ox + k%2, c] = (k == argmax) ? dpooled : 0
// ------------------------------------------------------------------------------------------------
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dp, const uint8_t* __restrict__ argmax, float* __restrict__ dx,
                                    int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, Cq = C / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * Ho * Wo * Cq) return;
  const int cq = idx % Cq;
  const size_t op = idx / Cq;
  const int ox = op % Wo;
  const int oy = (op / Wo) % Ho;
  const int n = op / ((size_t)Wo * Ho);
  const f32x4 g = *(const f32x4*)(dp + op * C + cq * 4);
  const uint32_t am = *(const uint32_t*)(argmax + op * C + cq * 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (((am >> (8 * e)) & 0xff) == (uint32_t)k) ? g[e] : 0.f;
    const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
    *(f32x4*)(dx + (((size_t)n * H + iy) * W + ix) * C + cq * 4) = v;
  }
}

extern "C" int dmd_maxpool2_bwd(const float* dpooled, const uint8_t* argmax, float* dx, int N, int H, int W, int C,
                                dmd_stream_t stream) {
  DMD_CHECK_ARG(dpooled && argmax && dx, "maxpool2_bwd: null");
  DMD_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2_bwd: H, W even, C %% 4 == 0");
  const size_t n = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dpooled,
                     argmax, dx, N, H, W, C);
  DMD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(+affine/FiLM) + SiLU backward.
//   forward: xh = (x - mean) * rstd ; u = xh * mul' + add ; a = silu(u)       (mul' = mul or 1 + mul)
//   given da:  du = da * silu'(u) ; dxh = du * mul'
//   dx = rstd * (dxh - mean_g(dxh) - xh * mean_g(dxh * xh)) [+ dskip]          (means over the group)
//   dmul[n][c] = sum_hw du * xh ; dadd[n][c] = sum_hw du                       (host sums over n for
//   batch-shared affine parameters)
// Pass A: grid (T, N), T = ceil(HW / 256): partial group sums (fp64) and per-channel sums.
// Pass B: elementwise.   Pass C: per-(n, c) reduction of the T channel partials.
// ------------------------------------------------------------------------------------------------
struct GnBwdElem {
  float xh, du, dxh;
};

__device__ __forceinline__ GnBwdElem gn_bwd_elem(float x, float da, float mean, float rstd, float mul, float add, bool silu) {
  GnBwdElem r;
  r.xh = (x - mean) * rstd;
  const float u = r.xh * mul + add;
  const float sg = dmd_sigmoid(u);
  r.du = silu ? da * (sg * (1.0f + u * (1.0f - sg))) : da;  // identity activation: attention pre-norm (blocks.py:64)
  r.dxh = r.du * mul;
  return r;
}

// valid extent (dmd_gn_bwd_params.W / valid_h / valid_w; W == 0: everything exists)
__device__ __forceinline__ bool gn_bwd_exists(const dmd_gn_bwd_params& p, int pix) {
  if (p.W == 0) return true;
  const int y = pix / p.W;
  return y < p.valid_h && pix - y * p.W < p.valid_w;
}
__device__ __forceinline__ double gn_bwd_count(const dmd_gn_bwd_params& p) { return p.W == 0 ? (double)p.HW : (double)p.valid_h * p.valid_w; }

#define GN_BWD_PIX 256  // pixels per workgroup in passes A and B
#define GN_BWD_MAXG 8   // C <= 256

template <int GN_BWD_U>  // pixels requested per thread and trip (dmd_gn_silu_bwd picks by launch size)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const dmd_gn_bwd_params p, int T, double* __restrict__ group_partial,
                                                            float* __restrict__ chan_partial) {
  __shared__ float g_mean[GN_BWD_MAXG], g_rstd[GN_BWD_MAXG];
  __shared__ double red[4][GN_BWD_MAXG][2];
  __shared__ float cred[256][8];
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int C = p.C, CQ = C / 4, G = C / DMD_GN_GROUP > 0 ? C / DMD_GN_GROUP : 1;
  const int gsz = C / G;
  if (tid < G) {
    float m, r;
    dmd_finalize_stats(p.norm.stats + ((size_t)(n * G + tid) * p.norm.stat_tiles) * 2, p.norm.stat_tiles,
                       (double)gsz * gn_bwd_count(p), &m, &r);
    g_mean[tid] = m;
    g_rstd[tid] = r;
  }
  __syncthreads();
  const int q = tid % CQ;
  const int c0 = 4 * q;
  const int g = c0 / gsz;
  const float mean = g_mean[g], rstd = g_rstd[g];
  float mul[4], add[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m = p.norm.mul ? p.norm.mul[(size_t)n * p.norm.mul_stride + c0 + e] : 1.0f;
    if (p.norm.mul_plus_one) m = 1.0f + m;
    mul[e] = m;
    add[e] = p.norm.add ? p.norm.add[(size_t)n * p.norm.add_stride + c0 + e] : 0.0f;
  }
  double s1 = 0.0, s2 = 0.0;
  float dm[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  // GN_BWD_U pixels per trip, every request in front of the first use (a thread's pixels in the same order as one at a time: the
  // same sums, the same bits).  A launch of the denoiser's training step (batch 32: 512 workgroups at 64 x 64, 128 at 32 x 32) has
  // two waves per SIMD or fewer, one request pair each: 4 requests deep the two passes take 26.8 instead of 28.3 us on average,
  // the step 7.80 instead of 7.89 ms; the actor-critic's launches (3,840 images: 61k workgroups, the SIMDs full) are 3 % FASTER one
  // deep -- occupancy already supplies the requests, and the deeper trip's registers cost a wave (profiles/r06n_ab_gn_bwd.txt).
  // A pixel that does not exist -- beyond the tile, outside the valid extent -- requests the trip's first pixel instead (a
  // predicated load is a branch, and behind a branch hipcc waits for everything: see wgrad_ps_kernel).
  const int pix_end = min(p.HW, (t + 1) * GN_BWD_PIX);
  const int pstride = 256 / CQ;
  for (int pix0 = t * GN_BWD_PIX + tid / CQ; pix0 < pix_end; pix0 += GN_BWD_U * pstride) {
    f32x4 xv[GN_BWD_U], dv[GN_BWD_U];
    bool ok[GN_BWD_U];
#pragma unroll
    for (int u = 0; u < GN_BWD_U; ++u) {
      const int pix = pix0 + u * pstride;
      ok[u] = pix < pix_end && gn_bwd_exists(p, pix);  // outside the valid extent: not part of any sum
      const size_t off = ((size_t)n * p.HW + (ok[u] ? pix : pix0)) * C + c0;
      xv[u] = *(const f32x4*)(p.x + off);
      dv[u] = *(const f32x4*)(p.da + off);
    }
#pragma unroll
    for (int u = 0; u < GN_BWD_U; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const GnBwdElem r = gn_bwd_elem(xv[u][e], dv[u][e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
        s1 += (double)r.dxh;
        s2 += (double)r.dxh * (double)r.xh;
        dm[e] += r.du * r.xh;
        db[e] += r.du;
      }
    }
  }
  // group sums: threads of a wave may belong to different groups (C = 64: quads 0-7 / 8-15)
  for (int gg = 0; gg < G; ++gg) {
    const double a = dmd_wave_sum(g == gg ? s1 : 0.0);
    const double b = dmd_wave_sum(g == gg ? s2 : 0.0);
    if ((tid & 63) == 0) {
      red[tid >> 6][gg][0] = a;
      red[tid >> 6][gg][1] = b;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cred[tid][e] = dm[e];
    cred[tid][4 + e] = db[e];
  }
  __syncthreads();
  if (tid < G) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) {
      a += red[w][tid][0];
      b += red[w][tid][1];
    }
    double* o = group_partial + ((size_t)(n * G + tid) * T + t) * 2;
    o[0] = a;
    o[1] = b;
  }
  if (tid < C) {  // channel tid: quad tid / 4, element tid % 4; fixed-order sum over the pixel lanes
    const int qq = tid >> 2, e = tid & 3;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < 256 / CQ; ++l) {
      a += cred[l * CQ + qq][e];
      b += cred[l * CQ + qq][4 + e];
    }
    float* o = chan_partial + (((size_t)n * T + t) * C + tid) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// Workgroup (0, n) also sums the T per-tile channel partials of image n into dmul / dadd, in tile order (round 4: a separate
// launch until then; bitwise the same sums, one launch less per normalisation: 12.05 -> 11.90 ms per denoiser training step)
template <int GN_BWD_U>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const dmd_gn_bwd_params p, int T, const double* __restrict__ group_partial,
                                                           const float* __restrict__ chan_partial) {
  __shared__ float g_mean[GN_BWD_MAXG], g_rstd[GN_BWD_MAXG], g_m1[GN_BWD_MAXG], g_m2[GN_BWD_MAXG];
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int C = p.C, CQ = C / 4, G = C / DMD_GN_GROUP > 0 ? C / DMD_GN_GROUP : 1;
  const int gsz = C / G;
  if (t == 0) {
    for (int c = tid; c < C; c += 256) {
      float a = 0.f, b = 0.f;
      for (int tt = 0; tt < T; ++tt) {
        const float* o = chan_partial + (((size_t)n * T + tt) * C + c) * 2;
        a += o[0];
        b += o[1];
      }
      p.dmul[(size_t)n * C + c] = a;
      p.dadd[(size_t)n * C + c] = b;
    }
  }
  if (tid < G) {
    float m, r;
    const double cnt = (double)gsz * gn_bwd_count(p);
    dmd_finalize_stats(p.norm.stats + ((size_t)(n * G + tid) * p.norm.stat_tiles) * 2, p.norm.stat_tiles, cnt, &m, &r);
    g_mean[tid] = m;
    g_rstd[tid] = r;
    double a = 0.0, b = 0.0;
    const double* gp = group_partial + ((size_t)(n * G + tid) * T) * 2;
    for (int k = 0; k < T; ++k) {
      a += gp[2 * k];
      b += gp[2 * k + 1];
    }
    g_m1[tid] = (float)(a / cnt);
    g_m2[tid] = (float)(b / cnt);
  }
  __syncthreads();
  const int q = tid % CQ;
  const int c0 = 4 * q;
  const int g = c0 / gsz;
  const float mean = g_mean[g], rstd = g_rstd[g], m1 = g_m1[g], m2 = g_m2[g];
  float mul[4], add[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m = p.norm.mul ? p.norm.mul[(size_t)n * p.norm.mul_stride + c0 + e] : 1.0f;
    if (p.norm.mul_plus_one) m = 1.0f + m;
    mul[e] = m;
    add[e] = p.norm.add ? p.norm.add[(size_t)n * p.norm.add_stride + c0 + e] : 0.0f;
  }
  const int pix_end = min(p.HW, (t + 1) * GN_BWD_PIX);
  const int pstride = 256 / CQ;
  // (two copies of the loop, with and without the skip gradient's request: a request under `if (p.dskip)` inside the loop is a branch)
  auto pass = [&](auto skip_c) __attribute__((always_inline)) {
    constexpr bool SKIP = decltype(skip_c)::value;
    for (int pix0 = t * GN_BWD_PIX + tid / CQ; pix0 < pix_end; pix0 += GN_BWD_U * pstride) {  // (GN_BWD_U requests deep: see pass A)
      f32x4 xv[GN_BWD_U], dv[GN_BWD_U], sv[SKIP ? GN_BWD_U : 1];
      bool ok[GN_BWD_U];
#pragma unroll
      for (int u = 0; u < GN_BWD_U; ++u) {
        const int pix = pix0 + u * pstride;
        ok[u] = pix < pix_end && gn_bwd_exists(p, pix);
        const size_t off = ((size_t)n * p.HW + (ok[u] ? pix : pix0)) * C + c0;
        xv[u] = *(const f32x4*)(p.x + off);
        dv[u] = *(const f32x4*)(p.da + off);
        if (SKIP) sv[SKIP ? u : 0] = *(const f32x4*)(p.dskip + off);
      }
#pragma unroll
      for (int u = 0; u < GN_BWD_U; ++u) {
        const int pix = pix0 + u * pstride;
        if (pix >= pix_end) continue;
        const size_t off = ((size_t)n * p.HW + pix) * C + c0;
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok[u]) {  // (outside the valid extent: a zero gradient)
          if (SKIP) o = sv[SKIP ? u : 0];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const GnBwdElem r = gn_bwd_elem(xv[u][e], dv[u][e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
            o[e] += rstd * (r.dxh - m1 - r.xh * m2);
          }
        }
        *(f32x4*)(p.dx + off) = o;
      }
    }
  };
  if (p.dskip)
    pass(std::true_type{});
  else
    pass(std::false_type{});
}

// HW <= 256 (the 16x16 and 8x8 levels: half of a training step's normalisations): ONE workgroup holds a whole image, so both
// passes are one launch -- pass A's sums stay in LDS, nothing goes through the workspace.  Every sum is formed as the two kernels
// above form it (T = 1: their "sums over tiles" are one term), every element is computed by the same expression: the results
// are bit-identical to the two launches (DIAMOND_GN_BWD_FUSED=0 takes those: test hook).
__global__ __launch_bounds__(256) void gn_bwd_fused_kernel(const dmd_gn_bwd_params p) {
  __shared__ float g_mean[GN_BWD_MAXG], g_rstd[GN_BWD_MAXG], g_m1[GN_BWD_MAXG], g_m2[GN_BWD_MAXG];
  __shared__ double red[4][GN_BWD_MAXG][2];
  __shared__ float cred[256][8];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, CQ = C / 4, G = C / DMD_GN_GROUP > 0 ? C / DMD_GN_GROUP : 1;
  const int gsz = C / G;
  const double cnt = (double)gsz * gn_bwd_count(p);
  if (tid < G) {
    float m, r;
    dmd_finalize_stats(p.norm.stats + ((size_t)(n * G + tid) * p.norm.stat_tiles) * 2, p.norm.stat_tiles, cnt, &m, &r);
    g_mean[tid] = m;
    g_rstd[tid] = r;
  }
  __syncthreads();
  const int q = tid % CQ;
  const int c0 = 4 * q;
  const int g = c0 / gsz;
  const float mean = g_mean[g], rstd = g_rstd[g];
  float mul[4], add[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m = p.norm.mul ? p.norm.mul[(size_t)n * p.norm.mul_stride + c0 + e] : 1.0f;
    if (p.norm.mul_plus_one) m = 1.0f + m;
    mul[e] = m;
    add[e] = p.norm.add ? p.norm.add[(size_t)n * p.norm.add_stride + c0 + e] : 0.0f;
  }
  double s1 = 0.0, s2 = 0.0;
  float dm[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pix = tid / CQ; pix < p.HW; pix += 256 / CQ) {
    if (!gn_bwd_exists(p, pix)) continue;
    const size_t off = ((size_t)n * p.HW + pix) * C + c0;
    const f32x4 xv = *(const f32x4*)(p.x + off);
    const f32x4 dv = *(const f32x4*)(p.da + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const GnBwdElem r = gn_bwd_elem(xv[e], dv[e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
      s1 += (double)r.dxh;
      s2 += (double)r.dxh * (double)r.xh;
      dm[e] += r.du * r.xh;
      db[e] += r.du;
    }
  }
  for (int gg = 0; gg < G; ++gg) {
    const double a = dmd_wave_sum(g == gg ? s1 : 0.0);
    const double b = dmd_wave_sum(g == gg ? s2 : 0.0);
    if ((tid & 63) == 0) {
      red[tid >> 6][gg][0] = a;
      red[tid >> 6][gg][1] = b;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cred[tid][e] = dm[e];
    cred[tid][4 + e] = db[e];
  }
  __syncthreads();
  if (tid < G) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) {
      a += red[w][tid][0];
      b += red[w][tid][1];
    }
    g_m1[tid] = (float)(a / cnt);
    g_m2[tid] = (float)(b / cnt);
  }
  if (tid < C) {
    const int qq = tid >> 2, e = tid & 3;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < 256 / CQ; ++l) {
      a += cred[l * CQ + qq][e];
      b += cred[l * CQ + qq][4 + e];
    }
    p.dmul[(size_t)n * C + tid] = a;
    p.dadd[(size_t)n * C + tid] = b;
  }
  __syncthreads();
  const float m1 = g_m1[g], m2 = g_m2[g];
  for (int pix = tid / CQ; pix < p.HW; pix += 256 / CQ) {
    const size_t off = ((size_t)n * p.HW + pix) * C + c0;
    if (!gn_bwd_exists(p, pix)) {
      *(f32x4*)(p.dx + off) = (f32x4){0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const f32x4 xv = *(const f32x4*)(p.x + off);
    const f32x4 dv = *(const f32x4*)(p.da + off);
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.dskip) o = *(const f32x4*)(p.dskip + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const GnBwdElem r = gn_bwd_elem(xv[e], dv[e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
      o[e] += rstd * (r.dxh - m1 - r.xh * m2);
    }
    *(f32x4*)(p.dx + off) = o;
  }
}

// Round 6, for launches of at most one workgroup per CU (the denoiser training step's batch of 32: a launch is 32 workgroups and
// their latency, 14.7 us): the image stays in REGISTERS between the two passes (at most GN_FUSED_ITEMS float4 pairs per thread, requested up
// front: a workgroup per image is one wave of loads, not 16 dependent round trips, and pass B neither reloads nor recomputes
// sigmoid(u)); images with more items per thread (C = 128 at 16 x 16) take the two loops of round 4.  Same sums in the same order.  It
// needs every register of a SIMD (one workgroup per CU): launches of more images (the actor-critic backward: 3,840) stay on
// gn_bwd_fused_kernel above, which measured 259 us where this one takes 352 (tools/wgrad_bench.py gn, 16 x 16 x 64 channels).
#define GN_FUSED_ITEMS 16
__global__ __launch_bounds__(256) void gn_bwd_fused_regs_kernel(const dmd_gn_bwd_params p) {
  __shared__ float g_mean[GN_BWD_MAXG], g_rstd[GN_BWD_MAXG], g_m1[GN_BWD_MAXG], g_m2[GN_BWD_MAXG];
  __shared__ double red[4][GN_BWD_MAXG][2];
  __shared__ float cred[256][8];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, CQ = C / 4, G = C / DMD_GN_GROUP > 0 ? C / DMD_GN_GROUP : 1;
  const int gsz = C / G;
  const double cnt = (double)gsz * gn_bwd_count(p);
  const int q = tid % CQ;
  const int c0 = 4 * q;
  const int lanes = 256 / CQ;  // pixels per round
  const bool in_regs = p.HW <= GN_FUSED_ITEMS * lanes;  // uniform
  // ---- every request of the image first (unconditional: an item behind the image re-reads its last pixel, unused) ----
  f32x4 xr[GN_FUSED_ITEMS], dr[GN_FUSED_ITEMS];
  if (in_regs) {
#pragma unroll
    for (int it = 0; it < GN_FUSED_ITEMS; ++it) {
      const int pix = min(tid / CQ + it * lanes, p.HW - 1);
      const size_t off = ((size_t)n * p.HW + pix) * C + c0;
      xr[it] = *(const f32x4*)(p.x + off);
      dr[it] = *(const f32x4*)(p.da + off);
    }
  }
  if (tid < G) {
    float m, r;
    dmd_finalize_stats(p.norm.stats + ((size_t)(n * G + tid) * p.norm.stat_tiles) * 2, p.norm.stat_tiles, cnt, &m, &r);
    g_mean[tid] = m;
    g_rstd[tid] = r;
  }
  __syncthreads();
  const int g = c0 / gsz;
  const float mean = g_mean[g], rstd = g_rstd[g];
  float mul[4], add[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m = p.norm.mul ? p.norm.mul[(size_t)n * p.norm.mul_stride + c0 + e] : 1.0f;
    if (p.norm.mul_plus_one) m = 1.0f + m;
    mul[e] = m;
    add[e] = p.norm.add ? p.norm.add[(size_t)n * p.norm.add_stride + c0 + e] : 0.0f;
  }
  double s1 = 0.0, s2 = 0.0;
  float dm[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  if (in_regs) {
#pragma unroll
    for (int it = 0; it < GN_FUSED_ITEMS; ++it) {
      const int pix = tid / CQ + it * lanes;
      if (pix >= p.HW || !gn_bwd_exists(p, pix)) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const GnBwdElem r = gn_bwd_elem(xr[it][e], dr[it][e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
        s1 += (double)r.dxh;
        s2 += (double)r.dxh * (double)r.xh;
        dm[e] += r.du * r.xh;
        db[e] += r.du;
        xr[it][e] = r.xh;   // (kept for pass B: the normalised value and the gradient through the activation)
        dr[it][e] = r.dxh;
      }
    }
  } else {
    for (int pix = tid / CQ; pix < p.HW; pix += lanes) {
      if (!gn_bwd_exists(p, pix)) continue;
      const size_t off = ((size_t)n * p.HW + pix) * C + c0;
      const f32x4 xv = *(const f32x4*)(p.x + off);
      const f32x4 dv = *(const f32x4*)(p.da + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const GnBwdElem r = gn_bwd_elem(xv[e], dv[e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
        s1 += (double)r.dxh;
        s2 += (double)r.dxh * (double)r.xh;
        dm[e] += r.du * r.xh;
        db[e] += r.du;
      }
    }
  }
  for (int gg = 0; gg < G; ++gg) {
    const double a = dmd_wave_sum(g == gg ? s1 : 0.0);
    const double b = dmd_wave_sum(g == gg ? s2 : 0.0);
    if ((tid & 63) == 0) {
      red[tid >> 6][gg][0] = a;
      red[tid >> 6][gg][1] = b;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cred[tid][e] = dm[e];
    cred[tid][4 + e] = db[e];
  }
  __syncthreads();
  if (tid < G) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) {
      a += red[w][tid][0];
      b += red[w][tid][1];
    }
    g_m1[tid] = (float)(a / cnt);
    g_m2[tid] = (float)(b / cnt);
  }
  if (tid < C) {
    const int qq = tid >> 2, e = tid & 3;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < 256 / CQ; ++l) {
      a += cred[l * CQ + qq][e];
      b += cred[l * CQ + qq][4 + e];
    }
    p.dmul[(size_t)n * C + tid] = a;
    p.dadd[(size_t)n * C + tid] = b;
  }
  __syncthreads();
  const float m1 = g_m1[g], m2 = g_m2[g];
  if (in_regs) {
#pragma unroll
    for (int it = 0; it < GN_FUSED_ITEMS; ++it) {
      const int pix = tid / CQ + it * lanes;
      if (pix >= p.HW) continue;
      const size_t off = ((size_t)n * p.HW + pix) * C + c0;
      f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (gn_bwd_exists(p, pix)) {
        if (p.dskip) o = *(const f32x4*)(p.dskip + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += rstd * (dr[it][e] - m1 - xr[it][e] * m2);
      }
      *(f32x4*)(p.dx + off) = o;
    }
    return;
  }
  for (int pix = tid / CQ; pix < p.HW; pix += lanes) {
    const size_t off = ((size_t)n * p.HW + pix) * C + c0;
    if (!gn_bwd_exists(p, pix)) {
      *(f32x4*)(p.dx + off) = (f32x4){0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const f32x4 xv = *(const f32x4*)(p.x + off);
    const f32x4 dv = *(const f32x4*)(p.da + off);
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.dskip) o = *(const f32x4*)(p.dskip + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const GnBwdElem r = gn_bwd_elem(xv[e], dv[e], mean, rstd, mul[e], add[e], p.identity_activation == 0);
      o[e] += rstd * (r.dxh - m1 - r.xh * m2);
    }
    *(f32x4*)(p.dx + off) = o;
  }
}

static inline int gn_bwd_tiles(int HW) { return (HW + GN_BWD_PIX - 1) / GN_BWD_PIX; }

extern "C" int64_t dmd_gn_bwd_workspace_bytes(int N, int HW, int C) {
  const int T = gn_bwd_tiles(HW);
  const int G = C / DMD_GN_GROUP > 0 ? C / DMD_GN_GROUP : 1;
  return (int64_t)N * G * T * 2 * 8 + (int64_t)N * T * C * 2 * 4;
}

extern "C" int dmd_gn_silu_bwd(const dmd_gn_bwd_params* pp, dmd_stream_t stream) {
  DMD_CHECK_ARG(pp && pp->x && pp->da && pp->dx && pp->workspace && pp->dmul && pp->dadd, "gn_silu_bwd: null");
  DMD_CHECK_ARG(pp->norm.stats && pp->norm.stat_tiles > 0, "gn_silu_bwd: statistics missing");
  DMD_CHECK_ARG((pp->W == 0 && pp->valid_h == 0 && pp->valid_w == 0) ||
                (pp->W > 0 && pp->HW % pp->W == 0 && pp->valid_h > 0 && pp->valid_h <= pp->HW / pp->W && pp->valid_w > 0 && pp->valid_w <= pp->W),
                "gn_silu_bwd: valid extent %d x %d of a (%d / %d) x %d tensor", pp->valid_h, pp->valid_w, pp->HW, pp->W, pp->W);
  DMD_CHECK_ARG(pp->C % 4 == 0 && pp->C <= 256 && 256 % (pp->C / 4) == 0 && (pp->C % DMD_GN_GROUP == 0 || pp->C < DMD_GN_GROUP),
                "gn_silu_bwd: unsupported C %d", pp->C);
  dmd_gn_bwd_params p = *pp;
  const int T = gn_bwd_tiles(p.HW);
  const int G = p.C / DMD_GN_GROUP > 0 ? p.C / DMD_GN_GROUP : 1;
  double* group_partial = (double*)p.workspace;
  float* chan_partial = (float*)((char*)p.workspace + (size_t)p.N * G * T * 2 * 8);
  hipStream_t st = (hipStream_t)stream;
  static DmdEnvInt fused_env{"DIAMOND_GN_BWD_FUSED", 1};
  if (T == 1 && fused_env.get() != 0) {
    // (DIAMOND_GN_BWD_FUSED: 0 two launches, 1 by launch size, 2 / 3 always the one / the other single launch -- test hooks:
    //  all four are bit-identical)
    if (fused_env.get() == 3 || (fused_env.get() != 2 && p.N <= 256))
      hipLaunchKernelGGL(gn_bwd_fused_regs_kernel, dim3(p.N), dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL(gn_bwd_fused_kernel, dim3(p.N), dim3(256), 0, st, p);
    DMD_LAUNCH_CHECK();
    return 0;
  }
  if ((long long)T * p.N <= 1024) {  // (at most four workgroups per CU: the requests have to come from inside a thread)
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<4>, dim3(T, p.N), dim3(256), 0, st, p, T, group_partial, chan_partial);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<4>, dim3(T, p.N), dim3(256), 0, st, p, T, (const double*)group_partial, (const float*)chan_partial);
  } else {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<1>, dim3(T, p.N), dim3(256), 0, st, p, T, group_partial, chan_partial);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, dim3(T, p.N), dim3(256), 0, st, p, T, (const double*)group_partial, (const float*)chan_partial);
  }
  DMD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// wgrad:  dW[co][ci][tap] = sum_{n,y,x} dy[n,y,x,co] * a[n, y + ty - 1, x + tx - 1, ci]
//
// GEMM view: rows = co (NCO blocks of 16), columns = (tap, ci) (NB = TAPS * NCI blocks of 16),
// contraction = pixels, 4 per v_mfma_f32_16x16x4_f32:
//   A operand (lane i = lane & 15, k = lane >> 4) = dy[pixel k][co i]
//   B operand (lane j = lane & 15, k = lane >> 4) = a[pixel k shifted by the tap][ci j]
// A workgroup (4 waves) walks a contiguous range of tiles (tile = two 8x8 pixel patches of
// possibly different images); per tile the activated input patches (+halo, ALL input channels)
// and the dy tile live in LDS with a row stride == 16 (mod 32) floats, which makes the
// ds_read_b32 operand fetches (16 consecutive channels x 2 pixels per 32-lane group) conflict
// free.  Wave w owns column blocks {w, w + 4, ...} for every row block: NCO + CB LDS reads feed
// NCO * CB MFMAs per 4 pixels.  Accumulators stay in registers across the workgroup's tiles
// and are written ONCE as a partial; a second kernel sums the partials in a fixed order
// (deterministic) and scatters into the OIHW gradient.
// ------------------------------------------------------------------------------------------------
template <int NCO_, int NCI_, int TAPS_>
struct WgradGeom {
  static constexpr int NCO = NCO_, NCI = NCI_, TAPS = TAPS_;
  static constexpr int NB = TAPS_ * NCI_;
  static constexpr int CB = (NB + 3) / 4;
  static constexpr int CIN = 16 * NCI_, COUT = 16 * NCO_;
  static constexpr int PAD = TAPS_ == 9 ? 1 : 0;
  static constexpr int PW = 8 + 2 * PAD;
  static constexpr int PP = PW * PW;                       // patch pixels per subtile
  static constexpr int SB = (CIN + 31) / 32 * 32 + 16;     // patch row stride (floats)
  static constexpr int SA = (COUT + 31) / 32 * 32 + 16;    // dy row stride
  static constexpr int PATCH_FLOATS = 2 * PP * SB;
  static constexpr int DY_FLOATS = 128 * SA;
  static constexpr int TAB_FLOATS = 2 * 3 * CIN;
  static constexpr int SMEM_BYTES = (PATCH_FLOATS + DY_FLOATS + TAB_FLOATS) * 4;
};

struct SubTile {
  int n, y0, x0;
  bool valid;
};

__device__ __forceinline__ SubTile wgrad_subtile(int N, int H, int W, int gs) {
  const int tx = W / 8, per_img = tx * (H / 8);
  SubTile t;
  t.valid = gs < N * per_img;
  const int g2 = t.valid ? gs : 0;
  t.n = g2 / per_img;
  const int r = g2 - t.n * per_img;
  t.y0 = (r / tx) * 8;
  t.x0 = (r % tx) * 8;
  return t;
}

// fp32 -> its split-fp16 pieces packed into the same 4 bytes: h = fp16(x) in the low half, l = fp16(x - h) in the high half
__device__ __forceinline__ float wg_pack_hl(float x) {
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  const unsigned u = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
  return __builtin_bit_cast(float, u);
}
typedef _Float16 wg_h8 __attribute__((ext_vector_type(8)));
// eight packed values (k = 0..7) -> the h and the l operand of v_mfma_f32_16x16x32_f16
__device__ __forceinline__ void wg_unpack8(const float (&r)[8], wg_h8& h, wg_h8& l) {
  unsigned u[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) u[e] = __builtin_bit_cast(unsigned, r[e]);
  const uint4 hh = {__builtin_amdgcn_perm(u[1], u[0], 0x05040100u), __builtin_amdgcn_perm(u[3], u[2], 0x05040100u),
                    __builtin_amdgcn_perm(u[5], u[4], 0x05040100u), __builtin_amdgcn_perm(u[7], u[6], 0x05040100u)};
  const uint4 ll = {__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u),
                    __builtin_amdgcn_perm(u[5], u[4], 0x07060302u), __builtin_amdgcn_perm(u[7], u[6], 0x07060302u)};
  h = __builtin_bit_cast(wg_h8, hh);
  l = __builtin_bit_cast(wg_h8, ll);
}

// SPLIT (template parameter of wgrad_kernel; dmd_wgrad_params.precision == DMD_PRECISION_F16X2):
//   false  exact: v_mfma_f32_16x16x4_f32, an fp32 fma chain over the pixels, each tile strictly load -> LDS -> barrier -> MFMA
//   true   the staged activations and dy are kept as packed split-fp16 pairs (same 4 bytes per value, same LDS layout, the same
//          two-lanes-per-bank ds_read_b32 pattern: lane (i, kg) supplies pixels (row r, column kg + 4 h), r < 4, h < 2) and 32
//          pixels are contracted per v_mfma_f32_16x16x32_f16 x 3 -- 12 instead of 128 matrix-pipe cycles per 16 pixels -- with
//          the NEXT tile's global loads issued before the MFMA phase of the current one and held in registers (one workgroup
//          per CU, 512 registers per lane: ~84 of them carry the tile in flight).  The bias gradient sums the raw dy.
// Round 4, measured (profiles/r04_staged_wgrad_ab.txt, r04_ab_train.txt; denoiser training step at batch 32, one hipGraph):
// 16-pixel MFMAs without prefetch on up to 1024 workgroups 13.99 ms -> 32-pixel MFMAs 13.53 -> + prefetch 13.11 -> at most 256
// workgroups (each writes one 147 KB partial of the whole gradient: a quarter of the reduction traffic) 12.01; a one-pass
// reduction of up to 256 partials instead of the two-pass one was slower (14.1) and is gone.
// the 32-pixel contraction over the staged tile
template <class G>
__device__ __forceinline__ void wgrad_mfma_k32(f32x4 (&acc)[G::NCO][G::CB], const float* dyt, const float* patch, const int (&boff)[G::CB],
                                               int aoff) {
#pragma unroll 1
  for (int kq = 0; kq < 4; ++kq) {
    const int s = kq >> 1, j = kq & 1;
    const float* ap = dyt + (size_t)(s * 64 + j * 32) * G::SA + aoff;
    const float* bp = patch + (size_t)(s * G::PP + 4 * j * G::PW) * G::SB;
    wg_h8 ah[G::NCO], al[G::NCO];
#pragma unroll
    for (int a = 0; a < G::NCO; ++a) {
      float r[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) r[v] = ap[a * 16 + ((v >> 1) * 8 + 4 * (v & 1)) * G::SA];
      wg_unpack8(r, ah[a], al[a]);
    }
#pragma unroll
    for (int b = 0; b < G::CB; ++b) {
      const float* q0 = bp + boff[b];
      float r[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) r[v] = q0[((v >> 1) * G::PW + 4 * (v & 1)) * G::SB];
      wg_h8 bh, bl;
      wg_unpack8(r, bh, bl);
#pragma unroll
      for (int a = 0; a < G::NCO; ++a) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh, acc[a][b], 0, 0, 0);
      }
    }
  }
}

template <class G, bool SPLIT>
__global__ __launch_bounds__(256) void wgrad_kernel(const dmd_wgrad_params p, int tiles_total, int tiles_per_wg) {
  DMD_DYNAMIC_LDS(float, smem);
  float* patch = smem;                        // [2][PP][SB]
  float* dyt = smem + G::PATCH_FLOATS;        // [128][SA]
  float* tab = dyt + G::DY_FLOATS;            // [2][3][CIN]: mean, a, add
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;

  f32x4 acc[G::NCO][G::CB];
#pragma unroll
  for (int a = 0; a < G::NCO; ++a)
#pragma unroll
    for (int s = 0; s < G::CB; ++s) acc[a][s] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // per-slot B offset: column block b = wave + 4 s -> (tap, ci block)
  int boff[G::CB];
#pragma unroll
  for (int s = 0; s < G::CB; ++s) {
    int b = wave + 4 * s;
    b = b < G::NB ? b : 0;  // surplus slots recompute block 0 and are never stored
    const int tap = b / G::NCI, cib = b - tap * G::NCI;
    const int ty = G::TAPS == 9 ? tap / 3 : 0, tx = G::TAPS == 9 ? tap % 3 : 0;
    boff[s] = (ty * G::PW + tx + kg) * G::SB + cib * 16 + i;
  }
  const int aoff = kg * G::SA + i;

  constexpr int CQO = G::COUT / 4;  // dy channel quads
  constexpr int CQI = G::CIN / 4;
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};  // bias gradient of channel quad (tid % CQO)
  const int Cx = p.src.C;                    // == CIN (checked on the host)
  const int Hv = p.valid_h ? p.valid_h : p.H, Wv = p.valid_w ? p.valid_w : p.W;  // valid extent (include/diamond_hip.h)
  const int tile_begin = blockIdx.x * tiles_per_wg;
  const int tile_end = min(tiles_total, tile_begin + tiles_per_wg);
  int tab_n0 = -1, tab_n1 = -1;

  if constexpr (SPLIT) {
    // ---- software-pipelined tile loop: raw loads of tile t + 1 fly under the MFMA phase of tile t ----
    constexpr int NPQ = 2 * G::PP * CQI;           // patch quads of a tile
    constexpr int NP = (NPQ + 255) / 256;          // ... per thread
    constexpr int ND = (128 * CQO) / 256;
    f32x4 px[NP], pd[ND];
    auto fetch = [&](int tile) {
      const SubTile f0 = wgrad_subtile(p.N, p.H, p.W, 2 * tile), f1 = wgrad_subtile(p.N, p.H, p.W, 2 * tile + 1);
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int id = it * 256 + tid;
        const int q = id % CQI, pp2 = id / CQI;
        const int s = pp2 >= G::PP ? 1 : 0;
        const int pp = pp2 - s * G::PP;
        const int py = pp / G::PW, pxx = pp - py * G::PW;
        const SubTile t = s ? f1 : f0;
        const int iy = t.y0 - G::PAD + py, ix = t.x0 - G::PAD + pxx;
        px[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (id < NPQ && t.valid && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv)
          px[it] = *(const f32x4*)(p.src.x + (((size_t)t.n * p.H + iy) * p.W + ix) * Cx + 4 * q);
      }
#pragma unroll
      for (int it = 0; it < ND; ++it) {
        const int id = it * 256 + tid;
        const int q = id % CQO, pix = id / CQO;
        const int s = pix >> 6, r = pix & 63;
        const SubTile t = s ? f1 : f0;
        pd[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t.valid && t.y0 + (r >> 3) < Hv && t.x0 + (r & 7) < Wv)
          pd[it] = *(const f32x4*)(p.dy + (((size_t)t.n * p.H + t.y0 + (r >> 3)) * p.W + t.x0 + (r & 7)) * G::COUT + 4 * q);
      }
    };
    if (tile_begin < tile_end) fetch(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
      SubTile st[2];
      st[0] = wgrad_subtile(p.N, p.H, p.W, 2 * tile);
      st[1] = wgrad_subtile(p.N, p.H, p.W, 2 * tile + 1);
      __syncthreads();  // previous tile's MFMA reads are done
      if (p.src.prologue != DMD_PROLOGUE_NONE && (st[0].n != tab_n0 || st[1].n != tab_n1)) {
        for (int c = tid; c < 2 * G::CIN; c += 256) {
          const int s = c / G::CIN, cc = c - s * G::CIN;
          float m, a, ad;
          norm_entry(p.src.norm, s ? st[1].n : st[0].n, cc, Cx, (double)(Cx < DMD_GN_GROUP ? Cx : DMD_GN_GROUP) * Hv * Wv, &m, &a, &ad);
          tab[(s * 3 + 0) * G::CIN + cc] = m;
          tab[(s * 3 + 1) * G::CIN + cc] = a;
          tab[(s * 3 + 2) * G::CIN + cc] = ad;
        }
        tab_n0 = st[0].n;
        tab_n1 = st[1].n;
        __syncthreads();
      }
      // ---- the tile in registers -> activated, split, into LDS ----
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int id = it * 256 + tid;
        if (id >= NPQ) continue;
        const int q = id % CQI, pp2 = id / CQI;
        const int s = pp2 >= G::PP ? 1 : 0;
        const int pp = pp2 - s * G::PP;
        const int py = pp / G::PW, pxx = pp - py * G::PW;
        const SubTile t = s ? st[1] : st[0];
        const int iy = t.y0 - G::PAD + py, ix = t.x0 - G::PAD + pxx;
        f32x4 v = px[it];
        if (p.src.prologue != DMD_PROLOGUE_NONE && t.valid && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cc = 4 * q + e;
            float u = (v[e] - tab[(s * 3 + 0) * G::CIN + cc]) * tab[(s * 3 + 1) * G::CIN + cc] + tab[(s * 3 + 2) * G::CIN + cc];
            if (p.src.prologue == DMD_PROLOGUE_NORM_SILU) u = dmd_silu(u);
            v[e] = u;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wg_pack_hl(v[e]);
        *(f32x4*)(patch + (size_t)pp2 * G::SB + 4 * q) = v;
      }
#pragma unroll
      for (int it = 0; it < ND; ++it) {
        const int id = it * 256 + tid;
        const int q = id % CQO, pix = id / CQO;
        f32x4 v = pd[it];
        bsum += v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wg_pack_hl(v[e]);
        *(f32x4*)(dyt + (size_t)pix * G::SA + 4 * q) = v;
      }
      __syncthreads();
      if (tile + 1 < tile_end) fetch(tile + 1);
      wgrad_mfma_k32<G>(acc, dyt, patch, boff, aoff);
    }
  } else
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    SubTile st[2];
    st[0] = wgrad_subtile(p.N, p.H, p.W, 2 * tile);
    st[1] = wgrad_subtile(p.N, p.H, p.W, 2 * tile + 1);
    __syncthreads();  // previous tile's MFMA reads are done
    if (p.src.prologue != DMD_PROLOGUE_NONE && (st[0].n != tab_n0 || st[1].n != tab_n1)) {
      for (int c = tid; c < 2 * G::CIN; c += 256) {
        const int s = c / G::CIN, cc = c - s * G::CIN;
        float m, a, ad;
        norm_entry(p.src.norm, s ? st[1].n : st[0].n, cc, Cx,  // (select, not st[s]: a per-lane index would put st[] in scratch)
                   (double)(Cx < DMD_GN_GROUP ? Cx : DMD_GN_GROUP) * Hv * Wv, &m, &a, &ad);
        tab[(s * 3 + 0) * G::CIN + cc] = m;
        tab[(s * 3 + 1) * G::CIN + cc] = a;
        tab[(s * 3 + 2) * G::CIN + cc] = ad;
      }
      tab_n0 = st[0].n;
      tab_n1 = st[1].n;
      __syncthreads();
    }
    // ---- stage the two activated input patches ----
    for (int id = tid; id < 2 * G::PP * CQI; id += 256) {
      const int q = id % CQI, pp2 = id / CQI;
      const int s = pp2 >= G::PP ? 1 : 0;
      const int pp = pp2 - s * G::PP;
      const int py = pp / G::PW, px = pp - py * G::PW;
      const SubTile t = s ? st[1] : st[0];
      const int iy = t.y0 - G::PAD + py, ix = t.x0 - G::PAD + px;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t.valid && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
        v = *(const f32x4*)(p.src.x + (((size_t)t.n * p.H + iy) * p.W + ix) * Cx + 4 * q);
        if (p.src.prologue != DMD_PROLOGUE_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cc = 4 * q + e;
            float u = (v[e] - tab[(s * 3 + 0) * G::CIN + cc]) * tab[(s * 3 + 1) * G::CIN + cc] + tab[(s * 3 + 2) * G::CIN + cc];
            if (p.src.prologue == DMD_PROLOGUE_NORM_SILU) u = dmd_silu(u);
            v[e] = u;
          }
        }
      }
      *(f32x4*)(patch + (size_t)pp2 * G::SB + 4 * q) = v;
    }
    // ---- stage dy (zero for a missing second subtile) ----
#pragma unroll
    for (int it = 0; it < (128 * CQO) / 256; ++it) {
      const int id = it * 256 + tid;
      const int q = id % CQO, pix = id / CQO;  // q == tid % CQO for every it
      const int s = pix >> 6, r = pix & 63;
      const SubTile t = s ? st[1] : st[0];
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t.valid && t.y0 + (r >> 3) < Hv && t.x0 + (r & 7) < Wv)
        v = *(const f32x4*)(p.dy + (((size_t)t.n * p.H + t.y0 + (r >> 3)) * p.W + t.x0 + (r & 7)) * G::COUT + 4 * q);
      bsum += v;
      *(f32x4*)(dyt + (size_t)pix * G::SA + 4 * q) = v;
    }
    __syncthreads();
    // ---- 32 k-groups of 4 pixels ----
#pragma unroll 2
    for (int kq = 0; kq < 32; ++kq) {
      const int s = kq >> 4, g = kq & 15;
      const int row = g >> 1, col0 = (g & 1) * 4;
      const float* ap = dyt + (size_t)(s * 64 + row * 8 + col0) * G::SA + aoff;
      const float* bp = patch + (size_t)(s * G::PP + row * G::PW + col0) * G::SB;
      float av[G::NCO], bv[G::CB];
#pragma unroll
      for (int a = 0; a < G::NCO; ++a) av[a] = ap[a * 16];
#pragma unroll
      for (int b = 0; b < G::CB; ++b) bv[b] = bp[boff[b]];
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int a = 0; a < G::NCO; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }

  // ---- partial results: [wg][NB][NCO][64 lanes][4] ----
  constexpr int PER_TOTAL = G::NB * G::NCO * 256 + G::COUT;  // [weights partial | bias partial] per workgroup
  float* part = p.workspace + (size_t)blockIdx.x * PER_TOTAL;
#pragma unroll
  for (int s = 0; s < G::CB; ++s) {
    const int b = wave + 4 * s;
    if (b < G::NB) {
#pragma unroll
      for (int a = 0; a < G::NCO; ++a) *(f32x4*)(part + ((size_t)(b * G::NCO + a) * 64 + lane) * 4) = acc[a][s];
    }
  }
  // ---- bias partial: sum over threads with the same channel quad ----
  __syncthreads();
  f32x4* red = (f32x4*)smem;
  red[tid] = bsum;
  __syncthreads();
  if (tid < CQO) {
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < 256 / CQO; ++l) a += red[l * CQO + tid];
    float* bpart = part + G::NB * G::NCO * 256;
    *(f32x4*)(bpart + 4 * tid) = a;
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad_ps_kernel -- the split-fp16 weight gradient as a producer / consumer pair (round 6).
//
// wgrad_kernel<G, true> above runs its two phases one after the other on four waves: a 128-pixel tile is staged (GroupNorm / FiLM /
// SiLU, h/l split, LDS writes: vector ALU), a barrier, then contracted (ds_read + MFMA: matrix pipe), a barrier -- one wave per
// SIMD, so nothing hides the other phase's latencies.  Measured (tools/wgrad_bench.py): 99.5 us for the 64 -> 64 weight gradient
// of 32 images of 64 x 64 (97 TFLOP/s algorithmic, 12 % of the f16 peak by executed MFMAs), ~21 us per tile and CU of which the
// MFMAs are 3.3.  Here a 512-thread workgroup splits the roles:
//   waves 0-3  CONSUME: accumulators in registers, operands from LDS, MFMAs only (wave w owns column blocks w, w + 4, ...);
//   waves 4-7  PRODUCE: the raw loads of sub-tile u + 2 fly while sub-tile u is activated, split and written to the OTHER
//              LDS buffer (two register sets, two LDS buffers: a sub-tile = one 8 x 8 pixel patch = half of the old tile, so
//              the pair of buffers is the old tile's LDS);
// one workgroup barrier per sub-tile.  A SIMD holds one wave of each role: the staging arithmetic issues under the MFMAs.
// Same plan (wgrad_plan: a workgroup = a contiguous range of 128-pixel tiles = sub-tiles 2 t, 2 t + 1), same partial layout,
// same order of accumulation per accumulator element (sub-tile by sub-tile, 32 pixels per MFMA, h.l + l.h + h.h) and per bias
// element: with DMD_PROLOGUE_NONE the partials are BITWISE those of wgrad_kernel<G, true>; a normalised source differs in the
// last bit of its activations (v_exp / v_rcp SiLU like the forward's split-fp16 kernels instead of expf and an IEEE division:
// the exact form is 45 instructions per element and would make the producers the critical path).
// The per-image table (mean, scale, shift per channel) is double-buffered in LDS and rebuilt by the producers one barrier
// ahead of the first sub-tile of a new image.
// ------------------------------------------------------------------------------------------------
// LDS operand layout of wgrad_ps_kernel: PIXEL PAIRS.  The k index of v_mfma_f32_16x16x32_f16 runs over pixels, a lane supplies
// k = 8 kg .. 8 kg + 7 = (row d, column kg) and (row d, column kg + 4), d = 0..3, of a 4-row slice -- one register per pair.  The
// single-role kernel stores one dword {h | l << 16} per (pixel, channel), so a consumer builds every operand register out of
// two loaded dwords (v_perm: 16 per operand, 1.5 vector instructions per MFMA -- on a SIMD whose vector issue the producer
// wave needs too: the first version of this kernel, same layout, ran staging + contraction additively).  Here the producers
// store what the MFMA reads: per pair (row, column c | c + 4) and channel one dword {h(c) | h(c + 4) << 16} in an H plane and
// one {l(c) | l(c + 4) << 16} in an L plane -- v_cvt_pk_f16_f32 + two v_fma_mix per two values instead of five instructions
// per value -- and an operand is four ds_read_b32, no arithmetic.  A patch row of 10 pixels is 6 pairs (columns 4, 5 are in two).
template <class G>
struct WgradPs {
  static constexpr int NPC = G::TAPS == 9 ? 6 : 4;         // pair columns of a patch row
  static constexpr int SPB = 2 * G::CIN + 16;               // dwords per patch pair: [H: CIN][L: CIN][pad]  (== 16 mod 32: the two
  static constexpr int SPA = 2 * G::COUT + 16;              //  k groups of a 32-lane read land in different bank halves)
  static constexpr int PATCH_DW = G::PW * NPC * SPB, DY_DW = 32 * SPA;
  static constexpr int BUF_FLOATS = PATCH_DW + DY_DW;       // one sub-tile
  static constexpr int TAB_FLOATS = 2 * 3 * G::CIN;         // two images' [mean | scale | shift][CIN]
  static constexpr int SMEM_BYTES = (2 * BUF_FLOATS + TAB_FLOATS) * 4;
  static constexpr int CQI = G::CIN / 4, CQO = G::COUT / 4;
  static constexpr int NPI = G::PW * NPC * CQI, NDI = 32 * CQO;  // items (pair x channel quad) of a sub-tile
  static constexpr int NP = (NPI + 255) / 256, ND = (NDI + 255) / 256;  // ... per producer thread
  static_assert(256 % CQI == 0 && 256 % CQO == 0, "a producer thread keeps one channel quad");
};

typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));
// -DDMD_LAB builds only (tools/wgrad_bench.py, WGRAD_LAB=bits in dmd_wgrad_params.precision >> 8): 1 no contraction, 2 no staging,
// 4 no prefetch -- how the three overlap (profiles/r06l_wgrad_bench_ps_*.txt).  The shipped library has no such switch.
#ifdef DMD_LAB
#define WG_LAB(bit) ((p.precision & (bit)) != 0)
#else
#define WG_LAB(bit) false
#endif
#ifdef WS_HOST_HELPERS
#define wg_low_pair ws_low_pair
#else
// {fp16(x0 - h0), fp16(x1 - h1)} of h01 = {h0, h1}: one mixed-precision fma per value (the difference is exact in fp32)
__device__ __forceinline__ unsigned wg_low_pair(float x0, float x1, unsigned h01) {
  unsigned l01;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(l01)
      : "v"(x0), "v"(x1), "v"(h01));
  return l01;
}
#endif
typedef _Float16 wg_h2 __attribute__((ext_vector_type(2)));

// NORMED = the source has a prologue (GroupNorm, with or without SiLU), a template parameter: as a run-time flag every element's
// arithmetic sat between two scalar branches (290 scalar instructions per sub-tile next to 380 vector ones).  WHICH prologue stays
// a run-time flag, on purpose: with the SiLU compiled as straight-line code (three instantiations, the whole prologue a constant)
// the 32 -> 32 launches of the actor-critic -- the one shape with TWO staging waves per SIMD, two workgroups per CU -- returned
// element 1 of every channel quad wrong now and then, run to run (1e-3 relative; only NORM_SILU sources, only above 256
// workgroups: tools/debug/wgrad_race.py, profiles/r06n_wgrad_race.txt).  Two idle cycles behind every v_exp / v_rcp made the errors
// rarer, not zero; the register reuse hipcc chose there (`v_rcp_f32 v98, v92` / `v_add_f32 v92, 1.0, v93`) is safe on its own at
// every occupancy (tools/probe/trans_war_probe.hip, profiles/r06o_trans_war_probe.txt): the mechanism is NOT established.  Behind a
// scalar branch per element the code is the second version's, which the run-to-run test has never caught.  (One kernel with three copies of the pipeline: 154 registers instead of 111 for that shape,
// its second workgroup per CU no longer fits, 1,630 -> 1,740 us: profiles/r06n_ab_wgrad_v3.txt.)
template <class G, bool NORMED>
__global__ __launch_bounds__(512) void wgrad_ps_kernel(const dmd_wgrad_params p, int tiles_total, int tiles_per_wg) {
  using P = WgradPs<G>;
  DMD_DYNAMIC_LDS(float, smem);
  float* tab = smem + 2 * P::BUF_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int Cx = p.src.C;  // == CIN (checked on the host)
  const int Hv = p.valid_h ? p.valid_h : p.H, Wv = p.valid_w ? p.valid_w : p.W;
  const int txs = p.W / 8, per_img = txs * (p.H / 8);
  const int sub_total = p.N * per_img;
  const int sub_begin = 2 * blockIdx.x * tiles_per_wg;
  const int sub_end = min(sub_total, 2 * min(tiles_total, (int)(blockIdx.x + 1) * tiles_per_wg));
  const int n = sub_end - sub_begin;  // > 0 (wgrad_plan), uniform
  constexpr int PER_TOTAL = G::NB * G::NCO * 256 + G::COUT;
  float* part = p.workspace + (size_t)blockIdx.x * PER_TOTAL;

  if (producer) {
    // ================================ producers ================================
    const int ptid = tid - 256;
    const int qi = ptid % P::CQI, qo = ptid % P::CQO;  // this thread's channel quads (the same for every item)
    f32x4 pxa[2 * P::NP], pda[2 * P::ND], pxb[2 * P::NP], pdb[2 * P::ND];  // (both pixels of a pair)
    f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};  // bias gradient of dy channel quad qo
    constexpr bool normed = NORMED;
    const bool silu = p.src.prologue == DMD_PROLOGUE_NORM_SILU;
    // Every request is UNCONDITIONAL (a pixel outside the image, the valid extent or the item count reads pixel (0, 0) of its
    // image and is zeroed when it is staged): a predicated load is a branch to hipcc, and behind a branch its wait insertion
    // stops counting -- `s_waitcnt vmcnt(0)` right behind the requests, i.e. no prefetch at all (the first version of this
    // kernel: loads, staging and MFMAs of the 32-channel launches added up to the launch time, 1,133 + 470 + 693 us of 2,064).
    // What an item is never changes -- its pair's row / column inside the sub-tile, its place in LDS: computed once.  Per
    // sub-tile a request is then the wave-uniform image base + a 32-bit byte offset, and whether it exists one bit of a mask
    // that travels with the register set (SQ counters of the second version: 400 vector instructions per producer wave and
    // sub-tile of the 32-channel shape, 150 of them arithmetic on values, vector issue 0.58 of the kernel's cycles).
    int prow_[P::NP], pcol_[P::NP], pdst_[P::NP], drow_[P::ND], dcol_[P::ND], ddst_[P::ND];
#pragma unroll
    for (int it = 0; it < P::NP; ++it) {
      const int pair = it * (256 / P::CQI) + ptid / P::CQI;
      const int prow = pair / P::NPC;
      prow_[it] = pair < G::PW * P::NPC ? prow - G::PAD : -(1 << 20);  // (an item beyond the patch: never inside any image)
      pcol_[it] = pair - prow * P::NPC - G::PAD;
      pdst_[it] = pair * P::SPB + 4 * qi;
    }
#pragma unroll
    for (int it = 0; it < P::ND; ++it) {
      const int pair = it * (256 / P::CQO) + ptid / P::CQO;
      drow_[it] = pair < 32 ? (pair >> 2) : (1 << 20);
      dcol_[it] = pair & 3;
      ddst_[it] = pair * P::SPA + 4 * qo;
    }
    unsigned oka = 0, okb = 0;  // bit 2 it + h: patch pixel h of item it exists; bit 16 + 2 it + h: dy pixel
    auto fetch = [&](f32x4 (&px)[2 * P::NP], f32x4 (&pd)[2 * P::ND], unsigned& okm, int u) __attribute__((always_inline)) {
      const int img = u / per_img, r = u - img * per_img;
      const int y0 = (r / txs) * 8, x0 = (r % txs) * 8;
      const char* xs = (const char*)(p.src.x + (size_t)img * p.H * p.W * Cx);  // wave-uniform
      const char* ds = (const char*)(p.dy + (size_t)img * p.H * p.W * G::COUT);
      unsigned m = 0;
#pragma unroll
      for (int it = 0; it < P::NP; ++it) {
        const int iy = y0 + prow_[it];
        const bool oky = (unsigned)iy < (unsigned)Hv;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ix = x0 + pcol_[it] + 4 * h;
          const bool ok = oky && (unsigned)ix < (unsigned)Wv;
          m |= ok ? 1u << (2 * it + h) : 0u;
          const unsigned off = ok ? ((unsigned)(iy * p.W + ix) * (unsigned)Cx + 4u * qi) * 4u : 16u * qi;  // (bytes inside the image: < 2^32)
          px[2 * it + h] = *(const f32x4*)(xs + off);
        }
      }
#pragma unroll
      for (int it = 0; it < P::ND; ++it) {
        const int oy = y0 + drow_[it];
        const bool oky = (unsigned)oy < (unsigned)Hv;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ox = x0 + dcol_[it] + 4 * h;
          const bool ok = oky && ox < Wv;
          m |= ok ? 1u << (16 + 2 * it + h) : 0u;
          const unsigned off = ok ? ((unsigned)(oy * p.W + ox) * (unsigned)G::COUT + 4u * qo) * 4u : 16u * qo;
          pd[2 * it + h] = *(const f32x4*)(ds + off);
        }
      }
      okm = m;
    };
    auto build_table = [&](int img, int which) __attribute__((always_inline)) {
      if (ptid < G::CIN) {
        float m, a, ad;
        norm_entry(p.src.norm, img, ptid, Cx, (double)(Cx < DMD_GN_GROUP ? Cx : DMD_GN_GROUP) * Hv * Wv, &m, &a, &ad);
        float* t = tab + which * 3 * G::CIN;
        t[ptid] = m;
        t[G::CIN + ptid] = a;
        t[2 * G::CIN + ptid] = ad;
      }
    };
    int which = 0;  // the table of the image being staged
    auto stage = [&](const f32x4 (&px)[2 * P::NP], const f32x4 (&pd)[2 * P::ND], unsigned okm, float* buf) __attribute__((always_inline)) {
      float tm[4], ta[4], tad[4];
      if (normed) {
        const float* t = tab + which * 3 * G::CIN + 4 * qi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          tm[e] = t[e];
          ta[e] = t[G::CIN + e];
          tad[e] = t[2 * G::CIN + e];
        }
      }
#pragma unroll
      for (int it = 0; it < P::NP; ++it) {
        if ((it + 1) * (256 / P::CQI) > G::PW * P::NPC && prow_[it] < -G::PAD) continue;  // (only the last round can lie beyond the patch)
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = (okm >> (2 * it + h)) & 1;  // (outside: the convolution's zero padding, after the prologue)
          v[h] = px[2 * it + h];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float w = v[h][e];
            if (normed) {
              w = (w - tm[e]) * ta[e] + tad[e];
              if (silu) w = dmd_silu_fast(w);
            }
            v[h][e] = ok ? w : 0.f;
          }
        }
        wg_u4 hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hh[e] = __builtin_bit_cast(unsigned, (wg_h2){(_Float16)v[0][e], (_Float16)v[1][e]});
          ll[e] = wg_low_pair(v[0][e], v[1][e], hh[e]);
        }
        float* dst = buf + pdst_[it];
        *(wg_u4*)dst = hh;
        *(wg_u4*)(dst + G::CIN) = ll;
      }
      float* dyt = buf + P::PATCH_DW;
#pragma unroll
      for (int it = 0; it < P::ND; ++it) {
        if (P::NDI < 256 && drow_[it] >= 8) continue;  // (COUT = 16: half of the producer threads have no dy item)
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = (okm >> (16 + 2 * it + h)) & 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[h][e] = ok ? pd[2 * it + h][e] : 0.f;
          bsum += v[h];
        }
        wg_u4 hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hh[e] = __builtin_bit_cast(unsigned, (wg_h2){(_Float16)v[0][e], (_Float16)v[1][e]});
          ll[e] = wg_low_pair(v[0][e], v[1][e], hh[e]);
        }
        float* dst = dyt + ddst_[it];
        *(wg_u4*)dst = hh;
        *(wg_u4*)(dst + G::COUT) = ll;
      }
    };
    // (the requests first: the table's own round trips -- partial sums, scale, shift -- then run under them; a launch of the
    //  16 x 16 or 8 x 8 level is two sub-tiles per workgroup and little else than this prologue)
    fetch(pxa, pda, oka, sub_begin);
    if (n > 1) fetch(pxb, pdb, okb, sub_begin + 1);
    if (normed) build_table(sub_begin / per_img, 0);
    __syncthreads();  // the first table
    // one step: sub-tile sub_begin + k out of register set (px, pd) into buffer k & 1, its successor-but-one requested into the
    // same registers, the next image's table if sub-tile k + 1 starts one
    auto step = [&](f32x4 (&px)[2 * P::NP], f32x4 (&pd)[2 * P::ND], unsigned& okm, int k) __attribute__((always_inline)) {
      if (k < n) {
        const int u = sub_begin + k;
        if (!WG_LAB(0x200)) stage(px, pd, okm, smem + (k & 1) * P::BUF_FLOATS);
        if (k + 2 < n && !WG_LAB(0x400)) fetch(px, pd, okm, u + 2);
        if (normed && k + 1 < n && (u + 1) / per_img != u / per_img) {
          which ^= 1;
          build_table((u + 1) / per_img, which);
        }
      }
    };
    for (int k = 0; k <= n; k += 2) {
      step(pxa, pda, oka, k);
      __syncthreads();
      if (k + 1 <= n) {
        step(pxb, pdb, okb, k + 1);
        __syncthreads();
      }
    }
    // ---- bias partial: sum over the producer threads with the same channel quad, in thread order ----
    f32x4* red = (f32x4*)smem;  // (every consumer read of the buffers is behind the last barrier)
    red[ptid] = bsum;
    __syncthreads();
    if (ptid < P::CQO) {
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < 256 / P::CQO; ++l) a += red[l * P::CQO + ptid];
      *(f32x4*)(part + G::NB * G::NCO * 256 + 4 * ptid) = a;
    }
    return;
  }

  // ================================ consumers ================================
  const int i = lane & 15, kg = lane >> 4;
  f32x4 acc[G::NCO][G::CB];
#pragma unroll
  for (int a = 0; a < G::NCO; ++a)
#pragma unroll
    for (int s = 0; s < G::CB; ++s) acc[a][s] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int boff[G::CB];  // per-slot B offset (dwords): column block b = wave + 4 s -> (tap, ci block)
#pragma unroll
  for (int s = 0; s < G::CB; ++s) {
    int b = wave + 4 * s;
    b = b < G::NB ? b : 0;  // surplus slots recompute block 0 and are never stored
    const int tap = b / G::NCI, cib = b - tap * G::NCI;
    const int ty = G::TAPS == 9 ? tap / 3 : 0, tx = G::TAPS == 9 ? tap % 3 : 0;
    boff[s] = (ty * P::NPC + tx + kg) * P::SPB + cib * 16 + i;
  }
  const int aoff = kg * P::SPA + i;
  auto contract = [&](const float* buf) __attribute__((always_inline)) {
    const unsigned* dyt = (const unsigned*)buf + P::PATCH_DW;
    const unsigned* pat = (const unsigned*)buf;
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {  // 32 pixels (rows 4 j .. 4 j + 3 of the sub-tile) per MFMA
      const unsigned* ap = dyt + (4 * j * 4) * P::SPA + aoff;
      const unsigned* bp = pat + (4 * j * P::NPC) * P::SPB;
      wg_h8 ah[G::NCO], al[G::NCO];
#pragma unroll
      for (int a = 0; a < G::NCO; ++a) {
        wg_u4 h, l;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          h[d] = ap[(d * 4) * P::SPA + a * 16];
          l[d] = ap[(d * 4) * P::SPA + a * 16 + G::COUT];
        }
        ah[a] = __builtin_bit_cast(wg_h8, h);
        al[a] = __builtin_bit_cast(wg_h8, l);
      }
#pragma unroll
      for (int b = 0; b < G::CB; ++b) {
        const unsigned* q0 = bp + boff[b];
        wg_u4 h, l;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          h[d] = q0[(d * P::NPC) * P::SPB];
          l[d] = q0[(d * P::NPC) * P::SPB + G::CIN];
        }
        const wg_h8 bh = __builtin_bit_cast(wg_h8, h), bl = __builtin_bit_cast(wg_h8, l);
#pragma unroll
        for (int a = 0; a < G::NCO; ++a) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh, acc[a][b], 0, 0, 0);
        }
      }
    }
  };
  __syncthreads();  // (the producers' first table)
  for (int k = 0; k <= n; k += 2) {
    if (k >= 1 && !WG_LAB(0x100)) contract(smem + ((k - 1) & 1) * P::BUF_FLOATS);
    __syncthreads();
    if (k + 1 <= n) {
      if (!WG_LAB(0x100)) contract(smem + (k & 1) * P::BUF_FLOATS);
      __syncthreads();
    }
  }
  // ---- partial results: [wg][NB][NCO][64 lanes][4] ----
#pragma unroll
  for (int s = 0; s < G::CB; ++s) {
    const int b = wave + 4 * s;
    if (b < G::NB) {
#pragma unroll
      for (int a = 0; a < G::NCO; ++a) *(f32x4*)(part + ((size_t)(b * G::NCO + a) * 64 + lane) * 4) = acc[a][s];
    }
  }
  __syncthreads();  // (the producers' bias reduction)
}

// Reduction of the per-workgroup partials, two deterministic passes:
//   pass 1: grid (elements / 256, WGRAD_SLICES): slice s sums workgroups {s, s + S, ...} -> ws2[s][e]
//   pass 2: element e of [NB][NCO][64][4] (+ bias) sums the S slices in order -> OIHW gradient
#define WGRAD_SLICES 16

__global__ void wgrad_reduce1_kernel(const float* __restrict__ ws, int num_wg, int per_total, float* __restrict__ ws2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int sl = blockIdx.y;
  if (idx >= per_total) return;
  float s = 0.f;
  for (int w = sl; w < num_wg; w += WGRAD_SLICES) s += ws[(size_t)w * per_total + idx];
  ws2[(size_t)sl * per_total + idx] = s;
}

__global__ void wgrad_reduce2_kernel(const float* __restrict__ ws2, int nslices, int NB, int NCO, int NCI, int taps, int cin_real,
                                     float* __restrict__ dw, float* __restrict__ dbias) {
  const int per = NB * NCO * 256;
  const int per_total = per + NCO * 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_total) return;
  double s = 0.0;
  for (int sl = 0; sl < nslices; ++sl) s += (double)ws2[(size_t)sl * per_total + idx];
  if (idx < per) {
    const int r = idx & 3, lane = (idx >> 2) & 63;
    const int blk = idx >> 8;
    const int cob = blk % NCO, b = blk / NCO;
    const int tap = b / NCI, cib = b - tap * NCI;
    const int co = cob * 16 + 4 * (lane >> 4) + r;
    const int ci = cib * 16 + (lane & 15);
    if (ci < cin_real) dw[((size_t)co * cin_real + ci) * taps + tap] = (float)s;
  } else if (dbias) {
    dbias[idx - per] = (float)s;
  }
}

// The same two reductions for MANY weight gradients in one launch (dmd_wgrad_reduce_jobs): a training step's backward holds ~60
// weight gradients whose reductions are ~140 launches of a few microseconds of work each, and nobody reads a weight gradient
// before the backward is over.  blockIdx.y = job (the table travels BY VALUE in the kernel arguments: no device table to
// upload, nothing a hipGraph replay could find stale), one thread per element, the additions of an element in exactly the
// order of the two kernels above (slice sums in fp32 in workgroup order, the slices in fp64 in slice order; few partials:
// directly in fp64) -- the gradients are bit-identical to the undeferred launches.  ld_cin / c0: the element lands in
// dw[(co * ld_cin + c0 + ci) * taps + tap], so the sources of a convolution over concatenated inputs write the slices of ONE
// OIHW tensor (no torch.cat afterwards).
#define WGRAD_JOBS_PER_LAUNCH 32
struct wgrad_job_batch {
  dmd_wgrad_reduce_job job[WGRAD_JOBS_PER_LAUNCH];
};

__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const wgrad_job_batch b) {
  const dmd_wgrad_reduce_job& j = b.job[blockIdx.y];
  const int per = j.NB * j.NCO * 256;
  const int per_total = per + j.NCO * 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_total) return;
  const float* __restrict__ ws = j.partials;
  double s = 0.0;
  if (j.num_wg <= 4 * WGRAD_SLICES) {
#pragma unroll 4
    for (int w = 0; w < j.num_wg; ++w) s += (double)ws[(size_t)w * per_total + idx];
  } else {
    for (int sl = 0; sl < WGRAD_SLICES; ++sl) {
      float f = 0.f;
#pragma unroll 4
      for (int w = sl; w < j.num_wg; w += WGRAD_SLICES) f += ws[(size_t)w * per_total + idx];  // (loads ahead, additions in order)
      s += (double)f;
    }
  }
  if (idx < per) {
    const int r = idx & 3, lane = (idx >> 2) & 63;
    const int blk = idx >> 8;
    const int cob = blk % j.NCO, bb = blk / j.NCO;
    const int tap = bb / j.NCI, cib = bb - tap * j.NCI;
    const int co = cob * 16 + 4 * (lane >> 4) + r;
    const int ci = cib * 16 + (lane & 15);
    if (ci < j.cin_real) j.dw[((size_t)co * j.ld_cin + j.c0 + ci) * j.taps + tap] = (float)s;
  } else if (j.dbias) {
    j.dbias[idx - per] = (float)s;
  }
}

static int wgrad_plan(const dmd_wgrad_params* p, int* tiles, int* num_wg, int* tpw) {
  const int sub = p->N * (p->H / 8) * (p->W / 8);
  *tiles = (sub + 1) / 2;
  // at most `cap` workgroups, each walking a contiguous range of tiles with its accumulators in registers: every workgroup
  // writes one partial of the whole gradient, so fewer of them is less reduction traffic.  256 = one workgroup per CU (round 4:
  // measured on the 64-output-channel instances, whose 80-105 KB of LDS allow one or barely two per CU); the 32-output-channel
  // 3x3 instances of the actor-critic encoder use 64 KB, two fit a CU and hide each other's load latency: 512 (round 6, same
  // box: wgrad<2,2,9> 6.45 -> 4.44 ms and <2,1,9> 2.42 -> 1.62 ms per window against +0.4 ms of reduction,
  // profiles/r06g_ab_wgrad_cap.txt).  The choice depends on the shape only.  DIAMOND_WGRAD_MAX_WG (1..1024; a test hook for the
  // multi-tile paths, and the A/B's knob) overrides it; the workspace is always sized for 1024.
  static DmdEnvInt cap_env{"DIAMOND_WGRAD_MAX_WG", -1};
  const int shape_cap = (p->Cout / 16 <= 2 && p->taps == 9) ? 512 : 256;
  const int cap = cap_env.get() >= 1 && cap_env.get() <= 1024 ? cap_env.get() : shape_cap;
  int n = *tiles < cap ? *tiles : cap;
  *tpw = (*tiles + n - 1) / n;
  *num_wg = (*tiles + *tpw - 1) / *tpw;
  return 0;
}

extern "C" int64_t dmd_wgrad_workspace_floats(const dmd_wgrad_params* p) {
  if (!p) return -1;
  const int sub = p->N * (p->H / 8) * (p->W / 8);
  const int tiles = (sub + 1) / 2;
  const int max_wg = tiles < 1024 ? tiles : 1024;  // the largest plan wgrad_plan can make (see DIAMOND_WGRAD_MAX_WG)
  const int64_t NB = (int64_t)p->taps * (p->src.C / 16), NCO = p->Cout / 16;
  return (int64_t)(max_wg + WGRAD_SLICES) * (NB * NCO * 256 + p->Cout);
}

template <int NCO, int NCI, int TAPS>
static int launch_wgrad(const dmd_wgrad_params& p, hipStream_t st) {
  using G = WgradGeom<NCO, NCI, TAPS>;
  int tiles, num_wg, tpw;
  wgrad_plan(&p, &tiles, &num_wg, &tpw);
  static bool attr_set[DMD_MAX_DEVICES] = {};  // per instantiation AND per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DMD_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<G, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::SMEM_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ps_kernel<G, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              WgradPs<G>::SMEM_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ps_kernel<G, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              WgradPs<G>::SMEM_BYTES);
    DMD_CHECK_ARG(e == hipSuccess, "wgrad: hipFuncSetAttribute(%d bytes): %s", G::SMEM_BYTES, hipGetErrorString(e));
    attr_set[dev] = true;
  }
  // DIAMOND_WGRAD_PS=0: the split-fp16 gradient on the single-role kernel (A/B, and the tests' bitwise comparison of the two)
  static DmdEnvInt ps_env{"DIAMOND_WGRAD_PS", 1};
  if ((p.precision & 0xff) == DMD_PRECISION_F16X2 && ps_env.get() != 0) {
    if (p.src.prologue != DMD_PROLOGUE_NONE)
      hipLaunchKernelGGL((wgrad_ps_kernel<G, true>), dim3(num_wg), dim3(512), WgradPs<G>::SMEM_BYTES, st, p, tiles, tpw);
    else
      hipLaunchKernelGGL((wgrad_ps_kernel<G, false>), dim3(num_wg), dim3(512), WgradPs<G>::SMEM_BYTES, st, p, tiles, tpw);
  } else if ((p.precision & 0xff) == DMD_PRECISION_F16X2)
    hipLaunchKernelGGL((wgrad_kernel<G, true>), dim3(num_wg), dim3(256), G::SMEM_BYTES, st, p, tiles, tpw);
  else
    hipLaunchKernelGGL((wgrad_kernel<G, false>), dim3(num_wg), dim3(256), G::SMEM_BYTES, st, p, tiles, tpw);
  const int per_total = G::NB * NCO * 256 + NCO * 16;
  if (p.defer_reduce) return 0;  // (the partials stay in the workspace: dmd_wgrad_job describes them, dmd_wgrad_reduce_jobs sums them)
  float* ws2 = p.workspace + (size_t)num_wg * per_total;
  const int single = 4 * WGRAD_SLICES;
  if (num_wg <= single) {
    // few partials (the low-resolution levels at the training batch): summed directly, in fp64, in workgroup order -- one
    // launch less per weight gradient (the training step is a chain of ~600 small kernels)
    hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((per_total + 255) / 256), dim3(256), 0, st, (const float*)p.workspace, num_wg, G::NB,
                       NCO, NCI, TAPS, p.cin_real, p.dw, p.dbias);
    return 0;
  }
  hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((per_total + 255) / 256, WGRAD_SLICES), dim3(256), 0, st, p.workspace, num_wg,
                     per_total, ws2);
  hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((per_total + 255) / 256), dim3(256), 0, st, (const float*)ws2, WGRAD_SLICES, G::NB,
                     NCO, NCI, TAPS, p.cin_real, p.dw, p.dbias);
  return 0;
}

extern "C" int dmd_wgrad_job(const dmd_wgrad_params* p, dmd_wgrad_reduce_job* job) {
  DMD_CHECK_ARG(p && job, "wgrad job: null");  // (workspace / dw / dbias may still be null: the caller fills the job's pointers)
  DMD_CHECK_ARG(p->taps == 9 || p->taps == 1, "wgrad job: taps");
  DMD_CHECK_ARG(p->Cout % 16 == 0 && p->src.C % 16 == 0 && p->cin_real > 0 && p->cin_real <= p->src.C, "wgrad job: channels");
  int tiles, num_wg, tpw;
  wgrad_plan(p, &tiles, &num_wg, &tpw);
  job->partials = p->workspace;
  job->dw = p->dw;
  job->dbias = p->dbias;
  job->num_wg = num_wg;
  job->NCI = p->src.C / 16;
  job->NCO = p->Cout / 16;
  job->NB = p->taps * job->NCI;
  job->taps = p->taps;
  job->cin_real = p->cin_real;
  job->ld_cin = p->cin_real;  // (the caller widens these two for a gradient that is a slice of a larger OIHW tensor)
  job->c0 = 0;
  return 0;
}

extern "C" int dmd_wgrad_reduce_jobs(const dmd_wgrad_reduce_job* jobs, int njobs, dmd_stream_t stream) {
  DMD_CHECK_ARG(njobs >= 0 && (jobs || njobs == 0), "wgrad reduce jobs: null");
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += WGRAD_JOBS_PER_LAUNCH) {
    wgrad_job_batch b;
    memset(&b, 0, sizeof(b));
    const int n = njobs - j0 < WGRAD_JOBS_PER_LAUNCH ? njobs - j0 : WGRAD_JOBS_PER_LAUNCH;
    int max_total = 0;
    for (int i = 0; i < n; ++i) {
      const dmd_wgrad_reduce_job& j = jobs[j0 + i];
      DMD_CHECK_ARG(j.partials && j.dw && j.num_wg > 0 && j.num_wg <= 1024 && j.NCO > 0 && j.NCI > 0 && (j.taps == 9 || j.taps == 1) &&
                        j.NB == j.taps * j.NCI,
                    "wgrad reduce jobs: job %d malformed", j0 + i);
      DMD_CHECK_ARG(j.cin_real > 0 && j.cin_real <= 16 * j.NCI && j.c0 >= 0 && j.c0 + j.cin_real <= j.ld_cin,
                    "wgrad reduce jobs: job %d: channels [%d, %d) outside a row of %d", j0 + i, j.c0, j.c0 + j.cin_real, j.ld_cin);
      b.job[i] = j;
      const int per_total = j.NB * j.NCO * 256 + j.NCO * 16;
      if (per_total > max_total) max_total = per_total;
    }
    hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3((max_total + 255) / 256, n), dim3(256), 0, st, b);
  }
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_conv2d_wgrad(const dmd_wgrad_params* p, dmd_stream_t stream) {
  DMD_CHECK_ARG(p && p->src.x && p->dy && p->workspace && p->dw, "wgrad: null");
  DMD_CHECK_ARG(p->defer_reduce == 0 || p->defer_reduce == 1, "wgrad: defer_reduce is 0 or 1");
  DMD_CHECK_ARG(p->N > 0 && p->H % 8 == 0 && p->W % 8 == 0, "wgrad: H, W must be multiples of 8 (%d x %d)", p->H, p->W);
  DMD_CHECK_ARG(p->taps == 9 || p->taps == 1, "wgrad: taps");
  DMD_CHECK_ARG(p->valid_h >= 0 && p->valid_h <= p->H && p->valid_w >= 0 && p->valid_w <= p->W && (p->valid_h == 0) == (p->valid_w == 0),
                "wgrad: valid extent %d x %d outside the %d x %d buffer", p->valid_h, p->valid_w, p->H, p->W);
  DMD_CHECK_ARG(p->cin_real > 0 && p->cin_real <= p->src.C, "wgrad: cin_real");
  if (p->src.prologue != DMD_PROLOGUE_NONE)
    DMD_CHECK_ARG(p->src.norm.stats && p->src.norm.stat_tiles > 0, "wgrad: prologue without statistics");
  hipStream_t st = (hipStream_t)stream;
  const int nco = p->Cout / 16, nci = p->src.C / 16;
  DMD_CHECK_ARG(p->Cout % 16 == 0 && p->src.C % 16 == 0, "wgrad: channels must be multiples of 16");
  int rc = -1;
  if (p->taps == 9) {
    if (nco == 2 && nci == 1) rc = launch_wgrad<2, 1, 9>(*p, st);
    else if (nco == 4 && nci == 1) rc = launch_wgrad<4, 1, 9>(*p, st);  // denoiser conv_in (15 -> 64)
    else if (nco == 1 && nci == 4) rc = launch_wgrad<1, 4, 9>(*p, st);  // denoiser conv_out (64 -> 3, dy padded to 16)
    else if (nco == 2 && nci == 2) rc = launch_wgrad<2, 2, 9>(*p, st);
    else if (nco == 4 && nci == 2) rc = launch_wgrad<4, 2, 9>(*p, st);
    else if (nco == 4 && nci == 4) rc = launch_wgrad<4, 4, 9>(*p, st);
  } else {
    if (nco == 4 && nci == 2) rc = launch_wgrad<4, 2, 1>(*p, st);
    else if (nco == 2 && nci == 2) rc = launch_wgrad<2, 2, 1>(*p, st);
    else if (nco == 4 && nci == 4) rc = launch_wgrad<4, 4, 1>(*p, st);
  }
  DMD_CHECK_ARG(rc >= 0, "wgrad: no kernel instance for Cin %d -> Cout %d, taps %d", p->src.C, p->Cout, p->taps);
  if (rc) return rc;
  DMD_LAUNCH_CHECK();
  return 0;
}
