// conv_lat_kernel -- the split-fp16 3x3 convolution for launches of a FEW tiles (play.py's interactive B = 1 sampler,
// reference src/play.py:105-109, src/game/play_env.py:113-124).
//
// STAGED: reached only with DIAMOND_CONV_LATENCY_TILES > 0 (dmd_conv2d routes launches of at most that many
// conv_f16ws tiles here); written after the round's GPU budget was spent, checked on the SIMT interpreter
// (tests/test_simt_kernels.py) and by tests/test_gpu_staged.py, NOT yet measured.  tools/gpu/staged_latency.sh measures it.
//
// Why another kernel.  conv_f16ws_kernel is a throughput design: one 768-thread workgroup per CU walks 256-pixel tiles
// through a producer / consumer pipeline of four 16-channel chunk steps of ~5,600 cycles each.  At B = 1 a 64x64 level is 16
// such tiles (6 % of the CUs), a 32x32 level 4, a 16x16 level ONE, and a launch costs the full pipeline depth of one tile:
// 17.5 us per launch on average, 102 launches per imagined frame = 1.9 of the 3.1 ms of kernel time per frame
// (profiles/r03_latency_b1_kernel_stats.csv).  Here the same arithmetic is cut the other way:
//   * a workgroup (4 waves) = one 8 x 16 pixel tile x ONE 32-channel half of the 64 output channels: 64 workgroups at
//     64x64, 16 at 32x32, 4 at 16x16 -- 4x the CUs per launch, a quarter of the work on each;
//   * no pipeline: all 256 threads stage the whole activated, split input patch (10 x 18 pixels x all input channels,
//     GroupNorm / FiLM / SiLU applied on the way, h and l planes) into LDS once, one barrier, then each wave runs its
//     32 pixels x 32 couts through K = Cin x 9 with the weight fragments read straight from L2 in MFMA A-operand order
//     (the dmd_pack_conv_weight_f16x2 layout IS that order: 16 bytes per lane, 512 contiguous bytes per half-wave) and
//     double-buffered in registers one 16-channel chunk ahead (18 x 16 bytes per lane in flight under 27 MFMAs);
//   * the first chunk's weights are requested BEFORE the staging phase, the statistics of a normalised source are
//     finalised by one wave per group (lanes over the partial sums + butterfly), not by a serial loop per channel;
//   * optional fused skip projection (dmd_conv_params.proj_*): the 128 raw channels of the tile are staged next to the
//     patch and contracted as eight more K steps into the same accumulators.
// Forms: conv_lat_kernel<PROJ, CQ, COUT, HEAD> (stride 1, W % 16 == 0: 16 / 32 / 64 / 128 input and 32 / 64 output channels,
// two sources, upsampling, the few-channel NCHW head), conv_lat_s2_kernel (stride 2: the 17 x 33 patch staged 32 channels at a
// time) and conv_lat_b8_kernel (W % 16 != 0: two 8 x 8 blocks per workgroup -- the 8 x 8 level of a recorded forward and its
// data gradients).  dmd_conv2d_latency_eligible() is the list.
// Same C ABI, same results to rounding (different summation order than conv_f16ws: a launch routed here is NOT bitwise
// the large-batch launch, which is why the route is by tile count and off by default), same statistics layout
// (dmd_conv_stat_tiles: one partial per 8 x 16 tile and 32-channel group = one workgroup).
#include <stdlib.h>

#include "dmd_common.h"

typedef _Float16 lt_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 lt_h4 __attribute__((ext_vector_type(4)));
typedef float lt_f16v __attribute__((ext_vector_type(16)));

#define LT_PW 18                    // patch columns (16 + halo)
#define LT_NPP 180                  // patch pixels (10 x 18)
#define LT_PROJ_C 128               // channels of the fused projection's input (two 64-channel sources)
#define LT_PROJ_RS (2 * LT_PROJ_C + 16)

// weights of one 16-channel chunk in A-operand order: [tap][h | l], lane (cout i = lane & 31, k group g = lane >> 5)
template <int TAPS, int COUT>
__device__ __forceinline__ void lt_load_w(lt_h8 (&w)[2 * TAPS], const lt_h8* __restrict__ wp, int chunk, int unit_lane) {
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) w[2 * t + hl] = wp[(((size_t)chunk * TAPS + t) * 2 + hl) * (2 * COUT) + unit_lane];
}

// one chunk of the 3x3 contraction: 9 taps x 3 MFMAs (w_h x_l + w_l x_h + w_h x_h); PITCH = patch pixels per row
template <int PITCH = LT_PW>
__device__ __forceinline__ void lt_chunk9(lt_f16v& acc, const lt_h8 (&w)[18], const unsigned char* ph, const unsigned char* pl, int pixbyte,
                                          int rs, int kbyte) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int off = pixbyte + ((t / 3) * PITCH + (t % 3)) * rs + kbyte;
    const lt_h8 bh = *(const lt_h8*)(ph + off);
    const lt_h8 bl = *(const lt_h8*)(pl + off);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * t], bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * t + 1], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * t], bh, acc, 0, 0, 0);
  }
}

// the same with the weight fragments read tap by tap (TP flavour: no chunk-sized register buffer)
template <int COUT>
__device__ __forceinline__ void lt_chunk9_stream(lt_f16v& acc, const lt_h8* __restrict__ wp, int chunk, int unit_lane, const unsigned char* ph,
                                                 const unsigned char* pl, int pixbyte, int rs, int kbyte) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const lt_h8 wh = wp[(((size_t)chunk * 9 + t) * 2 + 0) * (2 * COUT) + unit_lane];
    const lt_h8 wl = wp[(((size_t)chunk * 9 + t) * 2 + 1) * (2 * COUT) + unit_lane];
    const int off = pixbyte + ((t / 3) * LT_PW + (t % 3)) * rs + kbyte;
    const lt_h8 bh = *(const lt_h8*)(ph + off);
    const lt_h8 bl = *(const lt_h8*)(pl + off);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void lt_split_store(unsigned char* ph, unsigned char* pl, int byte, const f32x4 v) {
  lt_h4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (_Float16)v[e];
    l[e] = (_Float16)(v[e] - (float)h[e]);
  }
  *(lt_h4*)(ph + byte) = h;
  *(lt_h4*)(pl + byte) = l;
}

// CQ = input channel quads (4 / 8 / 16 / 32: 16 / 32 / 64 / 128 input channels), COUT = 32 | 64 output channels (of the packed
// weights), HEAD: the few-channel NCHW head (conv_out: p.Cout <= 4 real channels of 32 packed ones, no statistics / residual)
// TP: the THROUGHPUT flavour for the 32-channel layers at any batch (DIAMOND_CONV_LATENCY_TP=1): those launches are memory-bound,
// so the registers that buy latency inside ONE workgroup (chunk-sized weight buffers; bias and residual requested up front) are given
// up for occupancy -- 4 workgroups per CU instead of 2 hide each other's round trips.
template <bool PROJ, int CQ, int COUT, bool HEAD = false, bool TP = false>
__global__ __launch_bounds__(256, TP ? 4 : 1) void conv_lat_kernel(const dmd_conv_params p) {
  DMD_DYNAMIC_LDS(unsigned char, lt_smem);
  __shared__ float g_mean[4], g_rstd[4];
  __shared__ double red[4][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = blockIdx.y;  // which 32 of the COUT output channels
  const int tx_n = p.W / 16, per_img = tx_n * (p.H / 8);
  const int n = blockIdx.x / per_img, timg = blockIdx.x - n * per_img;
  const int y0 = (timg / tx_n) * 8, x0 = (timg % tx_n) * 16;

  const int up = p.upsample;
  const int Hs = p.H >> up, Ws = p.W >> up;
  constexpr int Cin = 4 * CQ;    // CQ divides 256 -> a thread always stages the same four channels
  const int C0 = p.src[0].C;
  constexpr int rs = 2 * Cin + 16;  // patch row stride in bytes: +16 spreads the 32 pixels of a B-operand read over the banks
  unsigned char* ph = lt_smem;
  unsigned char* pl = ph + LT_NPP * rs;
  unsigned char* jh = pl + LT_NPP * rs;            // PROJ: the tile's 128 pixels x 128 raw channels, h plane
  unsigned char* jl = jh + 128 * LT_PROJ_RS;       // ... l plane

  // ---- weights of chunk 0: requested before anything else, needed after the staging phase ----
  const int g = lane >> 5, ci = lane & 31;
  const int unit_lane = g * COUT + half * 32 + ci;  // [k group][cout] inside a (chunk, tap, piece) block of 2 COUT 16-byte units
  const lt_h8* wp = (const lt_h8*)p.w_f16;
  lt_h8 wa[18], wb[18];
  if constexpr (!TP) lt_load_w<9, COUT>(wa, wp, 0, unit_lane);

  // ---- everything that does not depend on anything else is REQUESTED now, in one batch: this thread's share of the patch,
  //      of the projection sources, its prologue parameters, the bias row and the residual of its output pixel.  (Fields of
  //      p.src[] are picked by selects between kernel arguments: indexing the argument struct with a per-thread index would
  //      turn every field into a dependent memory round trip.) ----
  const int q = tid % CQ;
  const int c4 = 4 * q;
  const bool s1 = c4 >= C0;  // this thread's four channels belong to the second source
  const int cl = c4 - (s1 ? C0 : 0);
  const float* sx = (s1 ? p.src[1].x : p.src[0].x) + cl;
  const int Cs = s1 ? p.src[1].C : C0;
  const int prol = s1 ? p.src[1].prologue : p.src[0].prologue;
  const float* mulp = s1 ? p.src[1].norm.mul : p.src[0].norm.mul;
  const float* addp = s1 ? p.src[1].norm.add : p.src[0].norm.add;
  const int64_t mul_stride = s1 ? p.src[1].norm.mul_stride : p.src[0].norm.mul_stride;
  const int64_t add_stride = s1 ? p.src[1].norm.add_stride : p.src[0].norm.add_stride;
  const int plus_one = s1 ? p.src[1].norm.mul_plus_one : p.src[0].norm.mul_plus_one;
  constexpr int PPSTEP = 256 / CQ, NIT = (LT_NPP + PPSTEP - 1) / PPSTEP;
  f32x4 sv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int pp = it * PPSTEP + tid / CQ;
    const int py = pp / LT_PW, px = pp - py * LT_PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    sv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pp < LT_NPP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
      sv[it] = *(const f32x4*)(sx + (((size_t)n * Hs + (iy >> up)) * Ws + (ix >> up)) * Cs);
  }
  f32x4 jv[PROJ ? 16 : 1];
  const int pq = tid & 31;  // PROJ: the tile's own 128 pixels of the two raw 64-channel sources (thread: channel quad tid % 32)
  if (PROJ) {
    const float* jx = (pq < 16 ? p.proj_x[0] : p.proj_x[1]) + 4 * (pq & 15);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int tp = it * 8 + (tid >> 5);
      jv[it] = *(const f32x4*)(jx + (((size_t)n * p.H + y0 + (tp >> 4)) * p.W + x0 + (tp & 15)) * 64);
    }
  }
  float mul[4] = {1.f, 1.f, 1.f, 1.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
  if (prol != DMD_PROLOGUE_NONE) {
    if (mulp) {
#pragma unroll
      for (int e = 0; e < 4; ++e) mul[e] = mulp[(size_t)n * mul_stride + cl + e];
    }
    if (addp) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ad[e] = addp[(size_t)n * add_stride + cl + e];
    }
  }
  // accumulators start from the bias row: register r holds cout 8 (r / 4) + 4 g + r % 4 of this half, pixel ci
  const int oy = y0 + 2 * wave + (ci >> 4), ox = x0 + (ci & 15);
  const size_t obase = (((size_t)n * p.H + oy) * p.W + ox) * COUT + half * 32 + 4 * g;
  float bias[16], pbias[PROJ ? 16 : 1], res[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = res[r] = 0.f;
#pragma unroll
  for (int r = 0; r < (PROJ ? 16 : 1); ++r) pbias[r] = 0.f;
  if (!TP && p.bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = p.bias[half * 32 + 8 * (r >> 2) + 4 * g + (r & 3)];
  }
  if (PROJ && p.proj_bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pbias[r] = p.proj_bias[half * 32 + 8 * (r >> 2) + 4 * g + (r & 3)];
  }
  if (!TP && p.residual) {
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = p.residual[obase + 8 * (r >> 2) + (r & 3)];
  }

  // ---- GroupNorm statistics: wave k finalises the k-th 32-channel group of the concatenated input ----
  constexpr int ngroups = Cin >> 5;
  if (wave < ngroups) {
    const bool w1 = wave * 32 >= C0;  // wave-uniform
    const int wprol = w1 ? p.src[1].prologue : p.src[0].prologue;
    if (wprol != DMD_PROLOGUE_NONE) {
      const int Cw = w1 ? p.src[1].C : C0;
      const int gi = (wave * 32 - (w1 ? C0 : 0)) >> 5, G = Cw >> 5;
      const int T = w1 ? p.src[1].norm.stat_tiles : p.src[0].norm.stat_tiles;
      const double* st = (w1 ? p.src[1].norm.stats : p.src[0].norm.stats) + ((size_t)(n * G + gi) * T) * 2;
      double s = 0.0, ss = 0.0;
      for (int t = lane; t < T; t += 64) {
        s += st[2 * t];
        ss += st[2 * t + 1];
      }
      s = dmd_wave_sum(s);
      ss = dmd_wave_sum(ss);
      if (lane == 0) {
        const double cnt = (double)DMD_GN_GROUP * Hs * Ws;
        const double m = s / cnt;
        double var = ss / cnt - m * m;
        var = var < 0.0 ? 0.0 : var;
        g_mean[wave] = (float)m;
        g_rstd[wave] = (float)(1.0 / sqrt(var + (double)DMD_GN_EPS));
      }
    }
  }
  __syncthreads();

  // ---- the patch: activated, split into h and l planes, into LDS ----
  float mean = 0.f, a[4] = {1.f, 1.f, 1.f, 1.f};
  if (prol != DMD_PROLOGUE_NONE) {
    mean = g_mean[c4 >> 5];
    const float rstd = g_rstd[c4 >> 5];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = rstd * (plus_one ? 1.0f + mul[e] : mul[e]);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int pp = it * PPSTEP + tid / CQ;
    if (pp >= LT_NPP) continue;
    const int py = pp / LT_PW, px = pp - py * LT_PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    f32x4 v = sv[it];
    if (prol != DMD_PROLOGUE_NONE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = (v[e] - mean) * a[e] + ad[e];
        v[e] = prol == DMD_PROLOGUE_NORM_SILU ? dmd_silu_fast(u) : u;
      }
    }
    lt_split_store(ph, pl, pp * rs + 8 * q, v);
  }
  if (PROJ) {
#pragma unroll
    for (int it = 0; it < 16; ++it) lt_split_store(jh, jl, (it * 8 + (tid >> 5)) * LT_PROJ_RS + 8 * pq, jv[it]);
  }
  lt_f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = PROJ ? bias[r] + pbias[r] : bias[r];
  __syncthreads();

  // ---- K loop: wave = pixel block (tile rows 2 wave, 2 wave + 1), lane pixel ci -> (row ci / 16, column ci % 16) ----
  const int pixbyte = ((2 * wave + (ci >> 4)) * LT_PW + (ci & 15)) * rs;  // top-left tap of this lane's pixel
  constexpr int nch = Cin >> 4;
  lt_h8 wj[16];
  if constexpr (nch == 1) lt_chunk9(acc, wa, ph, pl, pixbyte, rs, g * 16);  // conv_in: 16 (15 real) input channels
  if constexpr (TP) {  // weights tap by tap: their round trips are other workgroups' compute time
    for (int c = 0; c < nch; ++c) lt_chunk9_stream<COUT>(acc, wp, c, unit_lane, ph, pl, pixbyte, rs, c * 32 + g * 16);
  } else
  for (int c = 0; c + 1 < nch; c += 2) {
    lt_load_w<9, COUT>(wb, wp, c + 1, unit_lane);
    lt_chunk9(acc, wa, ph, pl, pixbyte, rs, c * 32 + g * 16);
    if (c + 2 < nch) lt_load_w<9, COUT>(wa, wp, c + 2, unit_lane);
    if (PROJ && c + 2 >= nch) {
      const lt_h8* wq = (const lt_h8*)p.proj_w_f16;
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) wj[2 * k + hl] = wq[((size_t)k * 2 + hl) * 128 + unit_lane];
    }
    lt_chunk9(acc, wb, ph, pl, pixbyte, rs, (c + 1) * 32 + g * 16);
  }
  if (PROJ) {
    const int tp = (2 * wave + (ci >> 4)) * 16 + (ci & 15);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int off = tp * LT_PROJ_RS + k * 32 + g * 16;
      const lt_h8 bh = *(const lt_h8*)(jh + off);
      const lt_h8 bl = *(const lt_h8*)(jl + off);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wj[2 * k], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wj[2 * k + 1], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wj[2 * k], bh, acc, 0, 0, 0);
    }
  }

  if constexpr (HEAD) {  // NCHW planes of the real channels: they are registers 0..Cout-1 of the lanes with k group 0
    if (g == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < p.Cout) p.out[(((size_t)n * p.Cout + r) * p.H + oy) * p.W + ox] = acc[r];
    }
    return;
  }
  // ---- write-out: residual, NHWC store, partial statistics of this (tile, 32-channel group) ----
  if constexpr (TP) {  // (bias: the accumulators started from zero here)
    if (p.bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) res[r] = p.bias[half * 32 + 8 * (r >> 2) + 4 * g + (r & 3)];
    }
    if (p.residual) {
#pragma unroll
      for (int r = 0; r < 16; ++r) res[r] += p.residual[obase + 8 * (r >> 2) + (r & 3)];
    }
  }
  double s = 0.0, ss = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = acc[4 * k + e] + res[4 * k + e];
      s += (double)o[e];
      ss += (double)o[e] * (double)o[e];
    }
    *(f32x4*)(p.out + obase + 8 * k) = o;
  }
  if (p.out_stats) {
    s = dmd_wave_sum(s);
    ss = dmd_wave_sum(ss);
    if (lane == 0) {
      red[wave][0] = s;
      red[wave][1] = ss;
    }
    __syncthreads();
    if (tid == 0) {
      double* o = p.out_stats + ((size_t)(n * (COUT / 32) + half) * per_img + timg) * 2;
      o[0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
      o[1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
  }
}

// ---- stride 2 (Downsample, blocks.py:93-100): the 8 x 16 output tile reads a 17 x 33 input patch, staged 32 channels at a
// time (two rounds for 64 input channels: 90 KB of LDS instead of 160); the next round's loads fly under this round's MFMAs ----
#define LT2_PW 33
#define LT2_NPP 561  // 17 x 33
#define LT2_RS 80    // 32 channels x 2 bytes + 16

template <int CQ, int COUT>
__global__ __launch_bounds__(256) void conv_lat_s2_kernel(const dmd_conv_params p) {
  DMD_DYNAMIC_LDS(unsigned char, lt_smem);
  __shared__ double red[4][2];
  constexpr int Cin = 4 * CQ, ROUNDS = Cin / 32, NIT = (LT2_NPP + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = blockIdx.y;
  const int tx_n = p.W / 16, per_img = tx_n * (p.H / 8);
  const int n = blockIdx.x / per_img, timg = blockIdx.x - n * per_img;
  const int y0 = (timg / tx_n) * 8, x0 = (timg % tx_n) * 16;
  const int Hi = 2 * p.H, Wi = 2 * p.W;  // input extent
  unsigned char* ph = lt_smem;
  unsigned char* pl = ph + LT2_NPP * LT2_RS;
  const int g = lane >> 5, ci = lane & 31;
  const int unit_lane = g * COUT + half * 32 + ci;
  const lt_h8* wp = (const lt_h8*)p.w_f16;
  lt_h8 wa[18], wb[18];
  lt_load_w<9, COUT>(wa, wp, 0, unit_lane);
  lt_load_w<9, COUT>(wb, wp, 1, unit_lane);

  // this thread stages channel quad q (of the round's 8) of patch pixels tid / 8 + 32 it
  const int q = tid & 7;
  f32x4 sv[NIT];
  auto fetch = [&](int round) {
    const float* sx = p.src[0].x + round * 32 + 4 * q;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pp = it * 32 + (tid >> 3);
      const int py = pp / LT2_PW, px = pp - py * LT2_PW;
      const int iy = 2 * y0 - 1 + py, ix = 2 * x0 - 1 + px;
      sv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (pp < LT2_NPP && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) sv[it] = *(const f32x4*)(sx + (((size_t)n * Hi + iy) * Wi + ix) * Cin);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pp = it * 32 + (tid >> 3);
      if (pp < LT2_NPP) lt_split_store(ph, pl, pp * LT2_RS + 8 * q, sv[it]);
    }
  };
  fetch(0);
  const int oy = y0 + 2 * wave + (ci >> 4), ox = x0 + (ci & 15);
  const size_t obase = (((size_t)n * p.H + oy) * p.W + ox) * COUT + half * 32 + 4 * g;
  float bias[16], res[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = res[r] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = p.bias[half * 32 + 8 * (r >> 2) + 4 * g + (r & 3)];
  }
  if (p.residual) {
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = p.residual[obase + 8 * (r >> 2) + (r & 3)];
  }
  commit();
  lt_f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bias[r];
  // top-left tap of this lane's output pixel (row 2 wave + ci / 16, column ci % 16) in the stride-2 patch
  const int pixbyte = ((2 * (2 * wave + (ci >> 4))) * LT2_PW + 2 * (ci & 15)) * LT2_RS;
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    if (round + 1 < ROUNDS) fetch(round + 1);
    __syncthreads();
    lt_chunk9<LT2_PW>(acc, wa, ph, pl, pixbyte, LT2_RS, g * 16);
    if (round + 1 < ROUNDS) lt_load_w<9, COUT>(wa, wp, 2 * round + 2, unit_lane);
    lt_chunk9<LT2_PW>(acc, wb, ph, pl, pixbyte, LT2_RS, 32 + g * 16);
    if (round + 1 < ROUNDS) {
      lt_load_w<9, COUT>(wb, wp, 2 * round + 3, unit_lane);
      __syncthreads();  // every wave has read this round's patch
      commit();
    }
  }
  double s = 0.0, ss = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = acc[4 * k + e] + res[4 * k + e];
      s += (double)o[e];
      ss += (double)o[e] * (double)o[e];
    }
    *(f32x4*)(p.out + obase + 8 * k) = o;
  }
  if (p.out_stats) {
    s = dmd_wave_sum(s);
    ss = dmd_wave_sum(ss);
    if (lane == 0) {
      red[wave][0] = s;
      red[wave][1] = ss;
    }
    __syncthreads();
    if (tid == 0) {
      double* o = p.out_stats + ((size_t)(n * (COUT / 32) + half) * per_img + timg) * 2;
      o[0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
      o[1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
  }
}

// ---- images whose width is not a multiple of 16 (the 8 x 8 level: W = 8; 24, 40, ...): a workgroup = TWO 8 x 8 pixel blocks
// (consecutive in (image, row, column) order, possibly of different images) x one 32-channel half; waves 0, 1 finish the
// rows 0-3 / 4-7 of block 0, waves 2, 3 those of block 1.  Statistics: one partial per 8 x 8 block (dmd_conv_stat_tiles). ----
template <int CQ, int COUT>
__global__ __launch_bounds__(256) void conv_lat_b8_kernel(const dmd_conv_params p, int nsub) {
  DMD_DYNAMIC_LDS(unsigned char, lt_smem);
  __shared__ float g_mean[2][4], g_rstd[2][4];
  __shared__ double red[4][2];
  constexpr int Cin = 4 * CQ, rs = 2 * Cin + 16, NPP = 200, PPSTEP = 256 / CQ, NIT = (NPP + PPSTEP - 1) / PPSTEP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = blockIdx.y;
  const int bx = p.W / 8, per_img = bx * (p.H / 8);
  int sn[2], sy[2], sx0[2], st[2];
  bool sval[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int gs = 2 * blockIdx.x + s;
    sval[s] = gs < nsub;
    const int g2 = sval[s] ? gs : 0;
    sn[s] = g2 / per_img;
    st[s] = g2 - sn[s] * per_img;
    sy[s] = (st[s] / bx) * 8;
    sx0[s] = (st[s] % bx) * 8;
  }
  const int C0 = p.src[0].C;
  unsigned char* ph = lt_smem;
  unsigned char* pl = ph + NPP * rs;
  const int g = lane >> 5, ci = lane & 31;
  const int unit_lane = g * COUT + half * 32 + ci;
  const lt_h8* wp = (const lt_h8*)p.w_f16;
  lt_h8 wa[18], wb[18];
  lt_load_w<9, COUT>(wa, wp, 0, unit_lane);

  const int q = tid % CQ, c4 = 4 * q;
  const bool s1 = c4 >= C0;
  const int cl = c4 - (s1 ? C0 : 0);
  const float* sx = (s1 ? p.src[1].x : p.src[0].x) + cl;
  const int Cs = s1 ? p.src[1].C : C0;
  const int prol = s1 ? p.src[1].prologue : p.src[0].prologue;
  const float* mulp = s1 ? p.src[1].norm.mul : p.src[0].norm.mul;
  const float* addp = s1 ? p.src[1].norm.add : p.src[0].norm.add;
  const int64_t mul_stride = s1 ? p.src[1].norm.mul_stride : p.src[0].norm.mul_stride;
  const int64_t add_stride = s1 ? p.src[1].norm.add_stride : p.src[0].norm.add_stride;
  const int plus_one = s1 ? p.src[1].norm.mul_plus_one : p.src[0].norm.mul_plus_one;
  f32x4 sv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int pp = it * PPSTEP + tid / CQ;
    const int sb = pp >= 100 ? 1 : 0, ppl = pp - 100 * sb;
    const int py = ppl / 10, px = ppl - py * 10;
    const int iy = (sb ? sy[1] : sy[0]) - 1 + py, ix = (sb ? sx0[1] : sx0[0]) - 1 + px;
    sv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pp < NPP && (sb ? sval[1] : sval[0]) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
      sv[it] = *(const f32x4*)(sx + (((size_t)(sb ? sn[1] : sn[0]) * p.H + iy) * p.W + ix) * Cs);
  }
  float mul[2][4], ad[2][4];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mul[sb][e] = 1.f;
      ad[sb][e] = 0.f;
    }
  if (prol != DMD_PROLOGUE_NONE) {
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      if (mulp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mul[sb][e] = mulp[(size_t)sn[sb] * mul_stride + cl + e];
      }
      if (addp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ad[sb][e] = addp[(size_t)sn[sb] * add_stride + cl + e];
      }
    }
  }
  // this lane's output pixel: block wave / 2, row 4 (wave & 1) + ci / 8, column ci % 8
  const int ob = wave >> 1, orow = 4 * (wave & 1) + (ci >> 3), ocol = ci & 7;
  const bool oval = ob ? sval[1] : sval[0];
  const size_t obase = (((size_t)(ob ? sn[1] : sn[0]) * p.H + (ob ? sy[1] : sy[0]) + orow) * p.W + (ob ? sx0[1] : sx0[0]) + ocol) * COUT + half * 32 + 4 * g;
  float bias[16], res[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = res[r] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = p.bias[half * 32 + 8 * (r >> 2) + 4 * g + (r & 3)];
  }
  if (p.residual && oval) {
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = p.residual[obase + 8 * (r >> 2) + (r & 3)];
  }

  // ---- GroupNorm statistics: (block, 32-channel group) combinations over the four waves ----
  constexpr int ngroups = Cin >> 5;
  for (int c = wave; c < 2 * ngroups; c += 4) {
    const int sb = c & 1, k = c >> 1;
    const bool w1 = k * 32 >= C0;
    const int wprol = w1 ? p.src[1].prologue : p.src[0].prologue;
    if (wprol != DMD_PROLOGUE_NONE) {
      const int Cw = w1 ? p.src[1].C : C0;
      const int gi = (k * 32 - (w1 ? C0 : 0)) >> 5, G = Cw >> 5;
      const int T = w1 ? p.src[1].norm.stat_tiles : p.src[0].norm.stat_tiles;
      const double* stp = (w1 ? p.src[1].norm.stats : p.src[0].norm.stats) + ((size_t)((sb ? sn[1] : sn[0]) * G + gi) * T) * 2;
      double s = 0.0, ss = 0.0;
      for (int t = lane; t < T; t += 64) {
        s += stp[2 * t];
        ss += stp[2 * t + 1];
      }
      s = dmd_wave_sum(s);
      ss = dmd_wave_sum(ss);
      if (lane == 0) {
        const double cnt = (double)DMD_GN_GROUP * p.H * p.W;
        const double m = s / cnt;
        double var = ss / cnt - m * m;
        var = var < 0.0 ? 0.0 : var;
        g_mean[sb][k] = (float)m;
        g_rstd[sb][k] = (float)(1.0 / sqrt(var + (double)DMD_GN_EPS));
      }
    }
  }
  __syncthreads();

  float mean[2] = {0.f, 0.f}, a[2][4];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb)
#pragma unroll
    for (int e = 0; e < 4; ++e) a[sb][e] = 1.f;
  if (prol != DMD_PROLOGUE_NONE) {
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      mean[sb] = g_mean[sb][c4 >> 5];
      const float rstd = g_rstd[sb][c4 >> 5];
#pragma unroll
      for (int e = 0; e < 4; ++e) a[sb][e] = rstd * (plus_one ? 1.0f + mul[sb][e] : mul[sb][e]);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int pp = it * PPSTEP + tid / CQ;
    if (pp >= NPP) continue;
    const int sb = pp >= 100 ? 1 : 0, ppl = pp - 100 * sb;
    const int py = ppl / 10, px = ppl - py * 10;
    const int iy = (sb ? sy[1] : sy[0]) - 1 + py, ix = (sb ? sx0[1] : sx0[0]) - 1 + px;
    f32x4 v = sv[it];
    if (prol != DMD_PROLOGUE_NONE && (sb ? sval[1] : sval[0]) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = (v[e] - (sb ? mean[1] : mean[0])) * (sb ? a[1][e] : a[0][e]) + (sb ? ad[1][e] : ad[0][e]);
        v[e] = prol == DMD_PROLOGUE_NORM_SILU ? dmd_silu_fast(u) : u;
      }
    }
    lt_split_store(ph, pl, pp * rs + 8 * q, v);
  }
  lt_f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bias[r];
  __syncthreads();

  const int pixbyte = (ob * 100 + orow * 10 + ocol) * rs;
  constexpr int nch = Cin >> 4;
  for (int c = 0; c + 1 < nch; c += 2) {
    lt_load_w<9, COUT>(wb, wp, c + 1, unit_lane);
    lt_chunk9<10>(acc, wa, ph, pl, pixbyte, rs, c * 32 + g * 16);
    if (c + 2 < nch) lt_load_w<9, COUT>(wa, wp, c + 2, unit_lane);
    lt_chunk9<10>(acc, wb, ph, pl, pixbyte, rs, (c + 1) * 32 + g * 16);
  }

  double s = 0.0, ss = 0.0;
  if (oval) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = acc[4 * k + e] + res[4 * k + e];
        s += (double)o[e];
        ss += (double)o[e] * (double)o[e];
      }
      *(f32x4*)(p.out + obase + 8 * k) = o;
    }
  }
  if (p.out_stats) {
    s = dmd_wave_sum(s);
    ss = dmd_wave_sum(ss);
    if (lane == 0) {
      red[wave][0] = s;
      red[wave][1] = ss;
    }
    __syncthreads();
    if (tid < 2 && (tid ? sval[1] : sval[0])) {  // thread b: the partial of block b = its two waves
      double* o = p.out_stats + ((size_t)((tid ? sn[1] : sn[0]) * (COUT / 32) + half) * per_img + (tid ? st[1] : st[0])) * 2;
      o[0] = red[2 * tid][0] + red[2 * tid + 1][0];
      o[1] = red[2 * tid][1] + red[2 * tid + 1][1];
    }
  }
}

// 1: these parameters can run on conv_lat_kernel (a subset of what conv_f16ws_kernel takes)
extern "C" int dmd_conv2d_latency_eligible(const dmd_conv_params* p) {
  if (!p || (p->precision & 0xff) != DMD_PRECISION_F16X2 || !p->w_f16) return 0;
  if (p->taps != 9 || p->residual_norm.stats) return 0;
  if (p->H % 8 != 0 || p->W % 8 != 0 || p->valid_h || p->valid_w) return 0;
  if (p->W % 16 != 0) {  // 8 x 8 blocks (conv_lat_b8_kernel): the plain stride-1 layers only
    if (p->stride != 1 || p->upsample || p->proj_nsrc || p->out_nchw || p->nsrc < 1 || p->nsrc > 2) return 0;
    int c8 = 0;
    for (int i = 0; i < p->nsrc; ++i) {
      if (!p->src[i].x || p->src[i].C % 32 != 0) return 0;
      c8 += p->src[i].C;
    }
    if (p->CoutPad != p->Cout) return 0;
    return ((p->Cout == 64 && (c8 == 64 || c8 == 128)) || (p->Cout == 32 && c8 == 32)) ? 1 : 0;
  }
  if (p->stride == 2)  // Downsample: one raw source, as many outputs as inputs
    return (p->nsrc == 1 && p->src[0].x && p->src[0].prologue == DMD_PROLOGUE_NONE && (p->src[0].C == 64 || p->src[0].C == 32) &&
            p->Cout == p->src[0].C && p->CoutPad == p->Cout && !p->out_nchw && !p->upsample && !p->proj_nsrc) ? 1 : 0;
  if (p->stride != 1) return 0;
  if (p->nsrc < 1 || p->nsrc > 2) return 0;
  int cin = 0;
  for (int i = 0; i < p->nsrc; ++i) {
    if (!p->src[i].x) return 0;
    cin += p->src[i].C;
  }
  // few-channel NCHW head (conv_out): Cout <= 4 zero-padded to 32 packed channels, 64 normalised input channels
  const bool head = p->out_nchw && p->Cout <= 4 && p->CoutPad == 32 && !p->residual && !p->out_stats && !p->proj_nsrc;
  if (head) return (p->nsrc == 1 && cin == 64 && !p->upsample) ? 1 : 0;
  if ((p->Cout != 64 && p->Cout != 32) || p->CoutPad != p->Cout || p->out_nchw) return 0;
  // conv_in: one 16-channel source (15 real), nothing to normalise
  if (cin == 16) return (p->nsrc == 1 && p->Cout == 64 && p->src[0].prologue == DMD_PROLOGUE_NONE && !p->upsample && !p->proj_nsrc) ? 1 : 0;
  for (int i = 0; i < p->nsrc; ++i)
    if (p->src[i].C % 32 != 0) return 0;  // whole GroupNorm groups per source, an even number of 16-channel chunks
  if (cin != 32 && cin != 64 && cin != 128) return 0;
  if (p->Cout == 32 && cin == 128) return 0;  // (no such layer; not instantiated)
  if ((long long)p->N * p->H * p->W * 128 * 4 >= (1ll << 40)) return 0;
  if (p->proj_nsrc) {
    if (p->proj_nsrc != 2 || !p->proj_w_f16 || !p->proj_x[0] || !p->proj_x[1] || p->proj_C[0] != 64 || p->proj_C[1] != 64) return 0;
    if (p->upsample || p->residual || cin != 64 || p->Cout != 64) return 0;
  }
  return 1;
}

// DIAMOND_CONV_LATENCY_TILES = the largest number of conv_f16ws tiles (256 pixels each) a launch may have to be routed here;
// 0 / unset: never (the kernel is staged, see the header of this file).  DIAMOND_CONV_LATENCY_TILES_C32 = the same cap for the
// 32-output-channel layers only (default: the general cap).  Those are memory-bound at ANY batch (288 MACs per input value:
// the 256-batch launch of the reward / end encoder runs at 2.5x its HBM time on the pipelined kernel), so a large value there
// asks whether plain occupancy -- two of these workgroups per CU, every load of a workgroup in flight at once -- streams
// better than the producer / consumer pipeline does; tools/gpu/staged_latency.sh measures that too.
// (Read per launch while the kernels are staged: tests and the A/B scripts flip the caps inside one process.)
int dmd_conv_lat_route(const dmd_conv_params& p) {
  const char* e = getenv("DIAMOND_CONV_LATENCY_TILES");
  long long cap = e ? atoll(e) : 0;
  if (p.Cout == 32)
    if (const char* e32 = getenv("DIAMOND_CONV_LATENCY_TILES_C32")) cap = atoll(e32);
  if (cap <= 0 || !dmd_conv2d_latency_eligible(&p)) return 0;
  const long long tiles16 = (long long)p.N * p.H * p.W / 256;  // 256-pixel tiles
  return tiles16 <= cap ? 1 : 0;
}

// the instantiation dmd_launch_conv_lat picks, spelled like rocprofv3's kernel trace (dmd_conv2d_kernel_name)
void dmd_conv_lat_kernel_name(const dmd_conv_params& p, char* buf, int buf_len) {
  const int cin = p.src[0].C + (p.nsrc > 1 ? p.src[1].C : 0);
  if (p.W % 16 != 0)
    snprintf(buf, buf_len, "conv_lat_b8_kernel<%d, %d>", cin / 4, p.Cout);
  else if (p.stride == 2)
    snprintf(buf, buf_len, "conv_lat_s2_kernel<%d, %d>", cin / 4, p.Cout);
  else if (p.out_nchw)
    snprintf(buf, buf_len, "conv_lat_kernel<false, 16, 32, true>");
  else
    snprintf(buf, buf_len, "conv_lat_kernel<%s, %d, %d, false>", p.proj_nsrc ? "true" : "false", cin / 4, p.Cout);
}

int dmd_launch_conv_lat(const dmd_conv_params& p, hipStream_t st) {
  const int cin = p.src[0].C + (p.nsrc > 1 ? p.src[1].C : 0);
  const int lds = 2 * LT_NPP * (2 * cin + 16) + (p.proj_nsrc ? 2 * 128 * LT_PROJ_RS : 0);
  static bool attr_set[DMD_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DMD_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {  // (the instances that can need more than 64 KiB of LDS)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lat_kernel<false, 32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lat_kernel<true, 16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    DMD_CHECK_ARG(e == hipSuccess, "conv_lat: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set[dev] = true;
  }
  if (p.W % 16 != 0) {
    static bool attr8[DMD_MAX_DEVICES] = {};
    if (!attr8[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lat_b8_kernel<32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      DMD_CHECK_ARG(e == hipSuccess, "conv_lat_b8: hipFuncSetAttribute: %s", hipGetErrorString(e));
      attr8[dev] = true;
    }
    const int nsub = p.N * (p.H / 8) * (p.W / 8);
    const dim3 grid8((unsigned)((nsub + 1) / 2), (unsigned)(p.Cout / 32));
    const int lds8 = 2 * 200 * (2 * cin + 16);
    if (p.Cout == 32)
      hipLaunchKernelGGL((conv_lat_b8_kernel<8, 32>), grid8, dim3(256), lds8, st, p, nsub);
    else if (cin == 64)
      hipLaunchKernelGGL((conv_lat_b8_kernel<16, 64>), grid8, dim3(256), lds8, st, p, nsub);
    else
      hipLaunchKernelGGL((conv_lat_b8_kernel<32, 64>), grid8, dim3(256), lds8, st, p, nsub);
    return 0;
  }
  const dim3 grid((unsigned)(p.N * (p.H / 8) * (p.W / 16)), (unsigned)(p.CoutPad / 32));
  if (p.stride == 2) {
    static bool attr2[DMD_MAX_DEVICES] = {};
    if (!attr2[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lat_s2_kernel<16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LT2_NPP * LT2_RS);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lat_s2_kernel<8, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LT2_NPP * LT2_RS);
      DMD_CHECK_ARG(e == hipSuccess, "conv_lat_s2: hipFuncSetAttribute: %s", hipGetErrorString(e));
      attr2[dev] = true;
    }
    if (p.Cout == 64)
      hipLaunchKernelGGL((conv_lat_s2_kernel<16, 64>), grid, dim3(256), 2 * LT2_NPP * LT2_RS, st, p);
    else
      hipLaunchKernelGGL((conv_lat_s2_kernel<8, 32>), grid, dim3(256), 2 * LT2_NPP * LT2_RS, st, p);
    return 0;
  }
  if (p.out_nchw)
    hipLaunchKernelGGL((conv_lat_kernel<false, 16, 32, true>), grid, dim3(256), lds, st, p);
  else if (cin == 16)
    hipLaunchKernelGGL((conv_lat_kernel<false, 4, 64>), grid, dim3(256), lds, st, p);
  else if (p.proj_nsrc)
    hipLaunchKernelGGL((conv_lat_kernel<true, 16, 64>), grid, dim3(256), lds, st, p);
  else if (p.Cout == 64 && cin == 128)
    hipLaunchKernelGGL((conv_lat_kernel<false, 32, 64>), grid, dim3(256), lds, st, p);
  else if (p.Cout == 64 && cin == 64)
    hipLaunchKernelGGL((conv_lat_kernel<false, 16, 64>), grid, dim3(256), lds, st, p);
  else if (p.Cout == 64)
    hipLaunchKernelGGL((conv_lat_kernel<false, 8, 64>), grid, dim3(256), lds, st, p);
  else if (cin == 64)
    hipLaunchKernelGGL((conv_lat_kernel<false, 16, 32>), grid, dim3(256), lds, st, p);
  else if (getenv("DIAMOND_CONV_LATENCY_TP") && atoi(getenv("DIAMOND_CONV_LATENCY_TP")) == 1)
    hipLaunchKernelGGL((conv_lat_kernel<false, 8, 32, false, true>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((conv_lat_kernel<false, 8, 32>), grid, dim3(256), lds, st, p);
  return 0;
}
