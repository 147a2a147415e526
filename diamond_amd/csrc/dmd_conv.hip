// dmd_conv2d -- 3x3 / 1x1 convolution as an implicit GEMM on v_mfma_f32_16x16x4_f32.
//
// Replaces the F.conv2d + F.group_norm + F.silu + torch.cat + F.interpolate chains of
// models/blocks.py:18-19,41-45,96,108-110,119,133-147 and inner_model.py:36,41,46,48.
//
// Work decomposition (one 256-thread workgroup = 4 wave64):
//   * output tile: 128 pixels (CFG A: one 8x16 patch of one image; CFG B: two 8x8 patches,
//     used when W is not a multiple of 16, e.g. the 8x8 level) x up to 64 output channels.
//   * GEMM view:  D[cout][pixel] += W[cout][cin,tap] * X[cin,tap][pixel]; the WEIGHTS are the
//     MFMA A operand (rows = cout) and the PIXELS the B operand (cols = pixel), so each lane
//     ends up with 4 consecutive couts of one pixel -> 16-byte NHWC stores.
//   * waves: WN along cout (16 couts each), WM = 4/WN along pixels, MB = 8/WM 16-pixel
//     m-blocks per wave; fp32 accumulators acc[MB] (f32x4).
//   * K loop: input channels in chunks of 16.  The (halo'd) input patch of a chunk lives in
//     LDS as [patch pixel][4 channel quads] float4, double buffered; the GroupNorm/FiLM
//     affine + SiLU is applied ONCE per input element while staging (zero padding is
//     applied after it, like the reference).  The 9 taps read shifted windows of the patch.
//     Quad index is rotated by 2*(pixel>>2) -> conflict-free ds_read_b128 for 16 consecutive
//     pixels (tools/lds_sim.py).
//   * weights stream from L2 straight into registers, one float4 per lane per (chunk, tap),
//     prefetched one step ahead (packed [chunk][tap][cout][16 cin], 1 KiB per wave-load).
//   * k-remap: lane (i, kg) holds cin quad kg as a float4; MFMA t of a step contracts
//     cin {4*k'+t}, k'=0..3 -- legal because A and B use the same map.
//   * epilogue: + bias, + (optionally GroupNorm'ed) residual, store, and per-(image, group)
//     fp64 partial (sum, sum^2) for the NEXT GroupNorm.
#include <stdlib.h>

#include "dmd_common.h"

#define DMD_CIN_MAX 256

// SPLIT_ (DMD_PRECISION_F16X2 launches the wave-specialised kernel does not cover: stride 2, 192-channel qkv, odd
// shapes): same staging and tiling, but the operands are kept as split-fp16 pieces -- a patch unit holds {h4, l4} of its
// four channels instead of four floats (same 16 bytes; the fragment layout of v_mfma_f32_16x16x16_f16 is the fp32 one,
// 4 consecutive k per lane), the weight fragment is split in registers once per tap, and a product is three f16 MFMAs
// (w_h x_l + w_l x_h + w_h x_h) into the fp32 accumulator: 24 matrix-pipe cycles per 16 channels instead of 128.
template <int WN_, bool CFGB_, int TAPS_, int STRIDE_, bool SPLIT_ = false>
struct ConvGeom {
  static constexpr bool SPLIT = SPLIT_;
  static constexpr int WN = WN_;
  static constexpr int WM = 4 / WN_;
  static constexpr int MB = 8 / WM;
  static constexpr bool CFGB = CFGB_;
  static constexpr int SUB = CFGB_ ? 2 : 1;
  static constexpr int TH = 8;
  static constexpr int TW = CFGB_ ? 8 : 16;
  static constexpr int TAPS = TAPS_;
  static constexpr int S = STRIDE_;
  static constexpr int PAD = TAPS_ == 9 ? 1 : 0;
  static constexpr int PH = (TH - 1) * STRIDE_ + 1 + 2 * PAD;
  static constexpr int PW = (TW - 1) * STRIDE_ + 1 + 2 * PAD;
  static constexpr int NPP = SUB * PH * PW;
  static constexpr int ITEMS = (NPP * 4 + 255) / 256;
  // stride-2 patches are ~4x larger: single-buffer them to stay under 64 KiB of static LDS
  static constexpr int NBUF = STRIDE_ == 2 ? 1 : 2;
  // minimum waves per SIMD the kernel is compiled for: the 64-cout 3x3 stride-1 instance on 8x8 patches (8 m-blocks per
  // wave + nine prefetched weight fragments) does not fit 256 registers -- one workgroup per SIMD set instead of spilling
  static constexpr int MIN_WAVES = (WN_ == 4 && CFGB_ && TAPS_ == 9 && STRIDE_ == 1) ? 1 : 2;
};

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// four fp32 values -> {h4 h, h4 l} in the same 16 bytes: h = fp16(x), l = fp16(x - h).  No clamp: see the range contract
// in dmd_conv_f16ws.hip (an operand beyond the fp16 range turns into NaNs, never into a silently saturated value).
__device__ __forceinline__ f32x4 split_h4l4(f32x4 v) {
  h4 hh, ll;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hh[e] = (_Float16)v[e];
    ll[e] = (_Float16)(v[e] - (float)hh[e]);
  }
  const uint2 a = __builtin_bit_cast(uint2, hh), b = __builtin_bit_cast(uint2, ll);
  return __builtin_bit_cast(f32x4, (uint4){a.x, a.y, b.x, b.y});
}

struct TileInfo {
  int n, y0, x0;
  bool valid;
};

template <class G>
__device__ __forceinline__ TileInfo decode_subtile(const dmd_conv_params& p, int tile, int s) {
  const int tiles_x = p.W / G::TW, tiles_y = p.H / G::TH;
  const int per_img = tiles_x * tiles_y;
  const int gs = tile * G::SUB + s;
  TileInfo t;
  t.valid = gs < p.N * per_img;
  const int g2 = t.valid ? gs : 0;
  t.n = g2 / per_img;
  const int r = g2 - t.n * per_img;
  const int ty = r / tiles_x;
  t.y0 = ty * G::TH;
  t.x0 = (r - ty * tiles_x) * G::TW;
  return t;
}

template <int SUB>
__device__ __forceinline__ TileInfo pick_tile(const TileInfo* ti, int s) {
  // select instead of ti[s]: a runtime-indexed register array would go to scratch
  return (SUB == 2 && s) ? ti[SUB - 1] : ti[0];
}

template <class G>
__global__ __launch_bounds__(256, G::MIN_WAVES) void conv_mfma_kernel(const dmd_conv_params p) {
  __shared__ f32x4 patch[G::NBUF][G::NPP * 4];
  __shared__ float tab_mean[G::SUB][DMD_CIN_MAX];
  __shared__ float tab_a[G::SUB][DMD_CIN_MAX];
  __shared__ float tab_add[G::SUB][DMD_CIN_MAX];
  __shared__ float rtab[G::SUB][3][64];
  __shared__ double red[G::SUB][4][2];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % G::WN;
  const int wm = wave / G::WN;
  const int j = lane & 15;   // pixel within m-block (B operand / D column)
  const int kg = lane >> 4;  // cin quad (operands) / cout quad (D rows)
  const int tile = blockIdx.x;
  const int cout_group0 = blockIdx.y * (16 * G::WN);

  TileInfo ti[G::SUB];
#pragma unroll
  for (int s = 0; s < G::SUB; ++s) ti[s] = decode_subtile<G>(p, tile, s);

  const int up = p.upsample;
  const int Hin = p.H * G::S, Win = p.W * G::S;  // conv-input extent
  const int Hs = Hin >> up, Ws = Win >> up;      // stored source extent
  // valid extent (include/diamond_hip.h): of the output, of the conv input, of the stored sources
  const int Hvo = p.valid_h ? p.valid_h : p.H, Wvo = p.valid_w ? p.valid_w : p.W;
  const int Hvi = Hvo * G::S, Wvi = Wvo * G::S;
  const int Hvs = Hvi >> up, Wvs = Wvi >> up;
  const int C0 = p.src[0].C;
  const int C1 = p.nsrc > 1 ? p.src[1].C : 0;
  const int nch0 = C0 >> 4;
  const int nchunks = (C0 + C1) >> 4;

  // ---- prologue tables (mean, a, add) per (subtile image, concatenated input channel) ----
  for (int c = tid; c < C0 + C1; c += 256) {
    const int si = c < C0 ? 0 : 1;
    const dmd_conv_src& sc = p.src[si];
    const int cl = si ? c - C0 : c;
#pragma unroll
    for (int s = 0; s < G::SUB; ++s) {
      float m = 0.f, a = 1.f, ad = 0.f;
      if (sc.prologue != DMD_PROLOGUE_NONE && ti[s].valid)
        norm_entry(sc.norm, ti[s].n, cl, sc.C, (double)DMD_GN_GROUP * Hvs * Wvs, &m, &a, &ad);
      tab_mean[s][c] = m;
      tab_a[s][c] = a;
      tab_add[s][c] = ad;
    }
  }
  if (p.residual_norm.stats) {
    for (int c = tid; c < 16 * G::WN; c += 256) {
#pragma unroll
      for (int s = 0; s < G::SUB; ++s) {
        float m = 0.f, a = 1.f, ad = 0.f;
        const int cc = cout_group0 + c;
        if (ti[s].valid && cc < p.Cout)
          norm_entry(p.residual_norm, ti[s].n, cc, p.Cout, (double)DMD_GN_GROUP * Hvo * Wvo, &m, &a, &ad);
        rtab[s][0][c] = m;
        rtab[s][1][c] = a;
        rtab[s][2][c] = ad;
      }
    }
  }

  // ---- staging items of this thread (chunk invariant) ----
  int goff[G::ITEMS];  // source pixel index, -1: zero (padding / outside / no item)
  int loff[G::ITEMS];  // float4 index in a patch buffer, -1: no item
  int isub[G::ITEMS];
  const int q = tid & 3;
#pragma unroll
  for (int it = 0; it < G::ITEMS; ++it) {
    const int id = it * 256 + tid;
    const int pp = id >> 2;
    const bool ok = pp < G::NPP;
    const int s = G::SUB == 1 ? 0 : (pp >= G::PH * G::PW ? 1 : 0);
    const TileInfo t = pick_tile<G::SUB>(ti, s);
    const int rem = pp - s * (G::PH * G::PW);
    const int py = rem / G::PW;
    const int px = rem - py * G::PW;
    const int iy = t.y0 * G::S - G::PAD + py;
    const int ix = t.x0 * G::S - G::PAD + px;
    const bool inb = ok && t.valid && iy >= 0 && iy < Hvi && ix >= 0 && ix < Wvi;
    goff[it] = inb ? ((t.n * Hs + (iy >> up)) * Ws + (ix >> up)) : -1;
    loff[it] = ok ? pp * 4 + ((q + 2 * (pp >> 2)) & 3) : -1;
    isub[it] = s;
  }

  // ---- per-lane pixel bases of the wave's m-blocks (top-left tap) ----
  int pixbase[G::MB];
#pragma unroll
  for (int mb = 0; mb < G::MB; ++mb) {
    const int gmb = wm * G::MB + mb;
    int s, y, x;
    if (G::CFGB) {
      s = gmb >> 2;
      y = (gmb & 3) * 2 + (j >> 3);
      x = j & 7;
    } else {
      s = 0;
      y = gmb;
      x = j;
    }
    pixbase[mb] = s * (G::PH * G::PW) + (y * G::S) * G::PW + x * G::S;
  }

  f32x4 acc[G::MB];
#pragma unroll
  for (int mb = 0; mb < G::MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // weight stream: packed [chunk][tap][CoutPad][16]; lane reads cout (16*wn + j'), quad kg
  // (A operand: row i = lane & 15 = cout within the block, k group = lane >> 4)
  const size_t wstep = (size_t)p.CoutPad * 16;
  const float* wlane = p.w + (size_t)(cout_group0 + 16 * wn + j) * 16 + 4 * kg;

  f32x4 stage[G::ITEMS];
  auto load_chunk = [&](int ck) {
    const int si = ck < nch0 ? 0 : 1;
    const dmd_conv_src& sc = p.src[si];
    const int c0 = (si ? ck - nch0 : ck) * 16 + 4 * q;
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
      stage[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (goff[it] >= 0) stage[it] = *(const f32x4*)(sc.x + (size_t)goff[it] * sc.C + c0);
    }
  };
  const bool fast_math = p.precision != DMD_PRECISION_F32;
  auto store_chunk = [&](int ck, int buf) {
    const int si = ck < nch0 ? 0 : 1;
    const int prologue = p.src[si].prologue;
    const int cc = ck * 16 + 4 * q;  // concatenated channel index of this thread's quad
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
      f32x4 v = stage[it];
      if (prologue != DMD_PROLOGUE_NONE && goff[it] >= 0) {
        const int s = isub[it];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (v[e] - tab_mean[s][cc + e]) * tab_a[s][cc + e] + tab_add[s][cc + e];
          // precision != F32 (no-grad world-model launches that fall back to this kernel): v_exp/v_rcp SiLU,
          // 1 ulp each, instead of IEEE expf + division -- the staging VALU work bounds the small-Cout layers
          if (prologue == DMD_PROLOGUE_NORM_SILU) t = fast_math ? dmd_silu_fast(t) : dmd_silu(t);
          v[e] = t;
        }
      }
      if (G::SPLIT) v = split_h4l4(v);
      if (loff[it] >= 0) patch[buf][loff[it]] = v;
    }
  };

  // A chunk's weight fragments (TAPS x 16 bytes per lane, L2-resident) are fetched all at once, right after the last MFMA
  // that reads the previous chunk's, so that ONE L2 round trip per chunk overlaps with the staging of the next patch:
  // fetched tap by tap, one step ahead, the latency (~1 us) was exposed at every tap once the MFMAs of a tap take
  // 0.1 us (SPLIT) -- a stride-2 launch with few workgroups was a chain of 36 such round trips.  (A second register set,
  // loaded a whole chunk ahead, spills in the stride-2 instances.)
  f32x4 wreg[G::TAPS];
  auto load_weights = [&](int ck, f32x4 (&dst)[G::TAPS]) {
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) dst[tap] = *(const f32x4*)(wlane + (size_t)(ck * G::TAPS + tap) * wstep);
  };

  __syncthreads();  // tables visible
  load_chunk(0);
  load_weights(0, wreg);
  store_chunk(0, 0);
  __syncthreads();

  for (int ck = 0; ck < nchunks; ++ck) {
    const int buf = G::NBUF == 2 ? (ck & 1) : 0;
    const bool more = ck + 1 < nchunks;
    if (more) load_chunk(ck + 1);
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int dy = G::TAPS == 9 ? tap / 3 : 0;
      const int dx = G::TAPS == 9 ? tap % 3 : 0;
      const f32x4 wf = wreg[tap];
      if (G::SPLIT) {
        const uint4 wb = __builtin_bit_cast(uint4, split_h4l4(wf));
        const h4 wh = __builtin_bit_cast(h4, (uint2){wb.x, wb.y}), wl = __builtin_bit_cast(h4, (uint2){wb.z, wb.w});
#pragma unroll
        for (int mb = 0; mb < G::MB; ++mb) {
          const int pix = pixbase[mb] + dy * G::PW + dx;
          const uint4 xb = __builtin_bit_cast(uint4, patch[buf][pix * 4 + ((kg + 2 * (pix >> 2)) & 3)]);
          const h4 xh = __builtin_bit_cast(h4, (uint2){xb.x, xb.y}), xl = __builtin_bit_cast(h4, (uint2){xb.z, xb.w});
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, xl, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl, xh, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, xh, acc[mb], 0, 0, 0);
        }
        // keep the fragment reads of a tap next to its MFMAs: left alone, the scheduler hoists the reads of all nine taps
        // (72 x 16 bytes per lane) above the first MFMA and the stride-2 instances spill
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int mb = 0; mb < G::MB; ++mb) {
        const int pix = pixbase[mb] + dy * G::PW + dx;
        const f32x4 xf = patch[buf][pix * 4 + ((kg + 2 * (pix >> 2)) & 3)];
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[0], acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[1], acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[2], acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[3], acc[mb], 0, 0, 0);
      }
    }
    if (more) load_weights(ck + 1, wreg);
    if (G::NBUF == 1) __syncthreads();  // everyone done reading the only buffer
    if (more) store_chunk(ck + 1, G::NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue: lane owns couts [cq, cq+4) of pixel j of each m-block ----
  const int cl = 16 * wn + 4 * kg;   // cout within the workgroup's group
  const int cq = cout_group0 + cl;   // global cout of acc[.][0]
  f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + cq);
  double ssum[G::SUB], ssq[G::SUB];
#pragma unroll
  for (int s = 0; s < G::SUB; ++s) ssum[s] = ssq[s] = 0.0;

#pragma unroll
  for (int mb = 0; mb < G::MB; ++mb) {
    const int gmb = wm * G::MB + mb;
    int s, y, x;
    if (G::CFGB) {
      s = gmb >> 2;
      y = (gmb & 3) * 2 + (j >> 3);
      x = j & 7;
    } else {
      s = 0;
      y = gmb;
      x = j;
    }
    const TileInfo t = pick_tile<G::SUB>(ti, s);
    if (!t.valid) continue;
    const int oy = t.y0 + y, ox = t.x0 + x;
    const size_t pixel = ((size_t)t.n * p.H + oy) * p.W + ox;
    f32x4 v = acc[mb] + bias;
    if (p.residual) {
      if (cq + 3 < p.Cout && (p.Cout & 3) == 0) {
        f32x4 r = *(const f32x4*)(p.residual + pixel * p.Cout + cq);
        if (p.residual_norm.stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = (r[e] - rtab[s][0][cl + e]) * rtab[s][1][cl + e] + rtab[s][2][cl + e];
        }
        v += r;
      } else {
        for (int e = 0; e < 4; ++e)
          if (cq + e < p.Cout) {
            float r = p.residual[pixel * p.Cout + cq + e];
            if (p.residual_norm.stats) r = (r - rtab[s][0][cl + e]) * rtab[s][1][cl + e] + rtab[s][2][cl + e];
            v[e] += r;
          }
      }
    }
    if (p.out_nchw) {
      for (int e = 0; e < 4; ++e)
        if (cq + e < p.Cout) p.out[(((size_t)t.n * p.Cout + cq + e) * p.H + oy) * p.W + ox] = v[e];
    } else if (cq + 3 < p.Cout && (p.Cout & 3) == 0) {
      *(f32x4*)(p.out + pixel * p.Cout + cq) = v;
    } else {
      for (int e = 0; e < 4; ++e)
        if (cq + e < p.Cout) p.out[pixel * p.Cout + cq + e] = v[e];
    }
    // a wave covers both subtiles only when WM == 1 (then s is a compile-time constant);
    // otherwise all of its m-blocks belong to ONE subtile and slot 0 is used.
    constexpr bool kBoth = G::CFGB && G::WM == 1;
    const bool counted = oy < Hvo && ox < Wvo;  // outside the valid extent: stored, not counted
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d = counted ? (double)v[e] : 0.0;
      if (kBoth && mb >= G::MB / 2) {
        ssum[G::SUB - 1] += d;
        ssq[G::SUB - 1] += d * d;
      } else {
        ssum[0] += d;
        ssq[0] += d * d;
      }
    }
  }

  if (p.out_stats) {  // uniform; requires Cout % 32 == 0 (checked on the host)
    constexpr bool kBoth = G::CFGB && G::WM == 1;
    const int swave = G::CFGB ? ((wm * G::MB) >> 2) : 0;  // subtile of a single-subtile wave
#pragma unroll
    for (int s = 0; s < G::SUB; ++s) {
      const double a = dmd_wave_sum(ssum[kBoth ? s : 0]);
      const double b = dmd_wave_sum(ssq[kBoth ? s : 0]);
      if (lane == 0) {
        const bool mine = kBoth || s == swave;
        red[s][wave][0] = mine ? a : 0.0;
        red[s][wave][1] = mine ? b : 0.0;
      }
    }
    __syncthreads();
    // group g of this workgroup = couts [32g, 32g+32) = waves with wn in {2g, 2g+1}
    constexpr int NG = G::WN / 2 > 0 ? G::WN / 2 : 1;
    if (tid < G::SUB * NG) {
      const int s = tid / NG, g = tid % NG;
      const TileInfo t = pick_tile<G::SUB>(ti, s);
      if (t.valid) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 4; ++w) {
          // waves that hold no pixel of subtile s contributed exact zeros to red[s][w]
          if (((w % G::WN) >> 1) == g) {
            a += red[s][w][0];
            b += red[s][w][1];
          }
        }
        const int tiles_x = p.W / G::TW, tiles_y = p.H / G::TH;
        const int T = tiles_x * tiles_y;
        const int tt = (t.y0 / G::TH) * tiles_x + t.x0 / G::TW;
        const int Gt = p.Cout / DMD_GN_GROUP;
        const int gg = cout_group0 / DMD_GN_GROUP + g;
        double* o = p.out_stats + ((size_t)(t.n * Gt + gg) * T + tt) * 2;
        o[0] = a;
        o[1] = b;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// naive twin: one thread per output element.  Same parameters, same semantics.
// ------------------------------------------------------------------------------------------
__global__ void conv_naive_kernel(const dmd_conv_params p, int stat_tw) {
  const size_t total = (size_t)p.N * p.H * p.W * p.Cout;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int co = idx % p.Cout;
  const size_t pixel = idx / p.Cout;
  const int ox = pixel % p.W;
  const int oy = (pixel / p.W) % p.H;
  const int n = pixel / ((size_t)p.W * p.H);
  const int S = p.stride, up = p.upsample, pad = p.taps == 9 ? 1 : 0, k = p.taps == 9 ? 3 : 1;
  const int Hin = p.H * S, Win = p.W * S, Hs = Hin >> up, Ws = Win >> up;
  const int Hvo = p.valid_h ? p.valid_h : p.H, Wvo = p.valid_w ? p.valid_w : p.W;
  const int Hvi = Hvo * S, Wvi = Wvo * S, Hvs = Hvi >> up, Wvs = Wvi >> up;
  float acc = 0.f;
  int cbase = 0;
  for (int si = 0; si < p.nsrc; ++si) {
    const dmd_conv_src& sc = p.src[si];
    for (int c = 0; c < sc.C; ++c) {
      float m = 0.f, a = 1.f, ad = 0.f;
      if (sc.prologue != DMD_PROLOGUE_NONE) norm_entry(sc.norm, n, c, sc.C, (double)DMD_GN_GROUP * Hvs * Wvs, &m, &a, &ad);
      const int cc = cbase + c;
      for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) {
          const int iy = oy * S - pad + dy, ix = ox * S - pad + dx;
          if (iy < 0 || iy >= Hvi || ix < 0 || ix >= Wvi) continue;
          float v = sc.x[(((size_t)n * Hs + (iy >> up)) * Ws + (ix >> up)) * sc.C + c];
          if (sc.prologue != DMD_PROLOGUE_NONE) {
            v = (v - m) * a + ad;
            if (sc.prologue == DMD_PROLOGUE_NORM_SILU) v = dmd_silu(v);
          }
          const float w = p.w[(((size_t)(cc >> 4) * p.taps + dy * k + dx) * p.CoutPad + co) * 16 + (cc & 15)];
          acc = fmaf(w, v, acc);
        }
    }
    cbase += sc.C;
  }
  if (p.bias) acc += p.bias[co];
  if (p.residual) {
    float r = p.residual[pixel * p.Cout + co];
    if (p.residual_norm.stats) {
      float m, a, ad;
      norm_entry(p.residual_norm, n, co, p.Cout, (double)DMD_GN_GROUP * Hvo * Wvo, &m, &a, &ad);
      r = (r - m) * a + ad;
    }
    acc += r;
  }
  if (p.out_nchw)
    p.out[(((size_t)n * p.Cout + co) * p.H + oy) * p.W + ox] = acc;
  else
    p.out[idx] = acc;
}

// stats for the naive path: one thread per (n, group, tile), same tiling as the MFMA kernel
__global__ void conv_naive_stats_kernel(const float* out, double* stats, int N, int H, int W, int C, int TW, int Hv, int Wv) {
  const int tiles_x = W / TW, tiles_y = H / 8, T = tiles_x * tiles_y, G = C / DMD_GN_GROUP;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * G * T) return;
  const int t = idx % T, g = (idx / T) % G, n = idx / (T * G);
  const int y0 = (t / tiles_x) * 8, x0 = (t % tiles_x) * TW;
  double s = 0.0, ss = 0.0;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < TW; ++x)
      for (int c = 0; c < DMD_GN_GROUP; ++c) {
        if (y0 + y >= Hv || x0 + x >= Wv) continue;  // outside the valid extent
        const double v = out[(((size_t)n * H + y0 + y) * W + x0 + x) * C + g * DMD_GN_GROUP + c];
        s += v;
        ss += v * v;
      }
  stats[(size_t)idx * 2] = s;
  stats[(size_t)idx * 2 + 1] = ss;
}

// OIHW -> packed [CinPad/16][taps][CoutPad][16]
__global__ void pack_weight_kernel(const float* oihw, float* packed, int Cout, int Cin, int k, int CoutPad, int CinPad) {
  const int taps = k * k;
  const size_t total = (size_t)(CinPad / 16) * taps * CoutPad * 16;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % 16;
  const int co = (idx / 16) % CoutPad;
  const int tap = (idx / (16 * (size_t)CoutPad)) % taps;
  const int chunk = idx / (16 * (size_t)CoutPad * taps);
  const int c = chunk * 16 + ci;
  float v = 0.f;
  if (co < Cout && c < Cin) v = oihw[((size_t)co * Cin + c) * taps + tap];
  packed[idx] = v;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
extern "C" int dmd_conv2d_proj_eligible(const dmd_conv_params* p);
static int validate_conv(const dmd_conv_params* p) {
  DMD_CHECK_ARG(p != nullptr, "conv: null params");
  DMD_CHECK_ARG(p->N > 0 && p->H > 0 && p->W > 0, "conv: bad N/H/W %d %d %d", p->N, p->H, p->W);
  DMD_CHECK_ARG(p->H % 8 == 0 && p->W % 8 == 0, "conv: H, W must be multiples of 8 (got %d x %d)", p->H, p->W);
  DMD_CHECK_ARG(p->taps == 9 || p->taps == 1, "conv: taps must be 9 or 1");
  DMD_CHECK_ARG(p->stride == 1 || (p->stride == 2 && p->taps == 9), "conv: bad stride");
  DMD_CHECK_ARG(!(p->upsample && p->stride != 1), "conv: upsample with stride");
  DMD_CHECK_ARG(p->nsrc == 1 || p->nsrc == 2, "conv: nsrc");
  int cin = 0;
  for (int i = 0; i < p->nsrc; ++i) {
    DMD_CHECK_ARG(p->src[i].x && p->src[i].C > 0 && p->src[i].C % 16 == 0, "conv: src %d channels %d", i, p->src[i].C);
    if (p->src[i].prologue != DMD_PROLOGUE_NONE) {
      DMD_CHECK_ARG(p->src[i].norm.stats && p->src[i].norm.stat_tiles > 0, "conv: src %d prologue without stats", i);
      DMD_CHECK_ARG(p->src[i].C % DMD_GN_GROUP == 0, "conv: normalised source needs C %% 32 == 0");
    }
    cin += p->src[i].C;
  }
  DMD_CHECK_ARG(cin <= DMD_CIN_MAX, "conv: Cin %d > %d", cin, DMD_CIN_MAX);
  DMD_CHECK_ARG(p->Cout > 0 && p->CoutPad >= p->Cout && p->CoutPad % 16 == 0, "conv: Cout/CoutPad");
  DMD_CHECK_ARG(p->w && p->out, "conv: null weight/out");
  if (p->out_stats) DMD_CHECK_ARG(p->Cout % DMD_GN_GROUP == 0 && p->CoutPad == p->Cout, "conv: out_stats needs Cout %% 32 == 0");
  if (p->residual_norm.stats) DMD_CHECK_ARG(p->residual && p->Cout % DMD_GN_GROUP == 0, "conv: residual_norm");
  if (p->upsample) DMD_CHECK_ARG(p->H % 2 == 0 && p->W % 2 == 0, "conv: upsample needs even output");
  DMD_CHECK_ARG(p->valid_h >= 0 && p->valid_h <= p->H && p->valid_w >= 0 && p->valid_w <= p->W && (p->valid_h == 0) == (p->valid_w == 0),
                "conv: valid extent %d x %d outside the %d x %d buffer", p->valid_h, p->valid_w, p->H, p->W);
  if (p->upsample) DMD_CHECK_ARG(p->valid_h % 2 == 0 && p->valid_w % 2 == 0, "conv: upsample needs an even valid extent");
  if (p->proj_nsrc)
    DMD_CHECK_ARG(dmd_conv2d_proj_eligible(p), "conv: fused skip projection on parameters dmd_conv2d_proj_eligible() rejects "
                  "(needs F16X2 3x3 stride 1, Cout 64, H, W %% 16 == 0, two 64-channel sources, no residual)");
  return 0;
}

// an "f16x2" launch that conv_f16ws does not cover (stride 2, qkv / out_proj, odd shapes, the data gradients of the training
// steps) runs the SPLIT instance of the generic kernel
static int conv_mfma_split(const dmd_conv_params& p) { return (p.precision & 0xff) == DMD_PRECISION_F16X2; }

template <int WN, bool CFGB, int TAPS, int STRIDE>
static void launch_conv(const dmd_conv_params& p, int groups, hipStream_t st) {
  using G = ConvGeom<WN, CFGB, TAPS, STRIDE>;
  const int tiles = (p.H / G::TH) * (p.W / G::TW) * p.N;
  dim3 grid((tiles + G::SUB - 1) / G::SUB, groups);
  if (conv_mfma_split(p))
    hipLaunchKernelGGL((conv_mfma_kernel<ConvGeom<WN, CFGB, TAPS, STRIDE, true>>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<G>), grid, dim3(256), 0, st, p);
}

template <int TAPS, int STRIDE>
static void dispatch_wn(const dmd_conv_params& p, hipStream_t st) {
  const bool cfgb = (p.W % 16) != 0;
  if (p.CoutPad % 64 == 0) {
    cfgb ? launch_conv<4, true, TAPS, STRIDE>(p, p.CoutPad / 64, st) : launch_conv<4, false, TAPS, STRIDE>(p, p.CoutPad / 64, st);
  } else if (p.CoutPad % 32 == 0) {
    cfgb ? launch_conv<2, true, TAPS, STRIDE>(p, p.CoutPad / 32, st) : launch_conv<2, false, TAPS, STRIDE>(p, p.CoutPad / 32, st);
  } else {
    cfgb ? launch_conv<1, true, TAPS, STRIDE>(p, p.CoutPad / 16, st) : launch_conv<1, false, TAPS, STRIDE>(p, p.CoutPad / 16, st);
  }
}

extern "C" int dmd_conv_stat_tiles(int H, int W) { return (H / 8) * (W / ((W % 16) ? 8 : 16)); }

int dmd_launch_conv_f16ws(const dmd_conv_params& p, hipStream_t st);  // dmd_conv_f16ws.hip (wave-specialised, persistent)
int dmd_launch_conv1x1_stream(const dmd_conv_params& p, hipStream_t st);  // dmd_conv1x1.hip (streaming 1x1, exact fp32)
extern "C" int dmd_conv1x1_stream_eligible(const dmd_conv_params* p);

extern "C" int dmd_conv2d(const dmd_conv_params* p, dmd_stream_t stream) {
  if (int e = validate_conv(p)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (dmd_conv1x1_stream_eligible(p)) {
    dmd_launch_conv1x1_stream(*p, st);
  } else if (dmd_conv2d_f16x2_eligible(p)) {
    if (int e = dmd_launch_conv_f16ws(*p, st)) return e;
  } else if (p->taps == 1)
    dispatch_wn<1, 1>(*p, st);
  else if (p->stride == 2)
    dispatch_wn<9, 2>(*p, st);
  else
    dispatch_wn<9, 1>(*p, st);
  DMD_LAUNCH_CHECK();
  return 0;
}

// Name of the kernel instantiation dmd_conv2d launches for these parameters, spelled like rocprofv3's kernel trace
// (without the "void " prefix and the argument list), so that bench.py's per-kernel records, profiles/*_kernel_stats.csv
// and the PMC tables share one key.
extern "C" int dmd_conv2d_kernel_name(const dmd_conv_params* p, char* buf, int buf_len) {
  if (int e = validate_conv(p)) return e;
  DMD_CHECK_ARG(buf && buf_len > 0, "kernel_name: buffer");
  const bool b8 = p->W % 16 != 0;
  if (dmd_conv1x1_stream_eligible(p)) {
    int cin = 0;
    for (int i = 0; i < p->nsrc; ++i) cin += p->src[i].C;
    snprintf(buf, buf_len, "conv1x1_stream_kernel<%d, %d, %s>", cin / 16, cin == 128 ? 2 : 4,
             (p->precision & 0xff) == DMD_PRECISION_F16X2 ? "true" : "false");
  } else if (p->proj_nsrc) {
    snprintf(buf, buf_len, "conv_f16ws_kernel<WsGeomProj>");
  } else if (dmd_conv2d_f16x2_eligible(p)) {
    snprintf(buf, buf_len, "conv_f16ws_kernel<WsGeom<%s, %d, %d>>", b8 ? "true" : "false", p->CoutPad == 64 ? 2 : 1, p->taps);
  } else {
    const int wn = p->CoutPad % 64 == 0 ? 4 : (p->CoutPad % 32 == 0 ? 2 : 1);
    snprintf(buf, buf_len, "conv_mfma_kernel<ConvGeom<%d, %s, %d, %d, %s>>", wn, b8 ? "true" : "false", p->taps, p->stride,
             conv_mfma_split(*p) ? "true" : "false");
  }
  return 0;
}

extern "C" int dmd_conv2d_naive(const dmd_conv_params* p, dmd_stream_t stream) {
  if (int e = validate_conv(p)) return e;
  DMD_CHECK_ARG(!p->proj_nsrc, "naive conv: no fused skip projection (run the projection as its own launch)");
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)p->N * p->H * p->W * p->Cout;
  dmd_conv_params q = *p;
  float* tmp_out = p->out;
  hipLaunchKernelGGL(conv_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q, 0);
  if (p->out_stats) {
    DMD_CHECK_ARG(!p->out_nchw, "naive conv: stats need NHWC output");
    const int TW = (p->W % 16) ? 8 : 16;
    const int n = p->N * (p->Cout / DMD_GN_GROUP) * dmd_conv_stat_tiles(p->H, p->W);
    hipLaunchKernelGGL(conv_naive_stats_kernel, dim3((n + 63) / 64), dim3(64), 0, st, tmp_out, p->out_stats, p->N, p->H,
                       p->W, p->Cout, TW, p->valid_h ? p->valid_h : p->H, p->valid_w ? p->valid_w : p->W);
  }
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_pack_conv_weight(const float* oihw, float* packed, int Cout, int Cin, int k, int CoutPad, int CinPad,
                                    dmd_stream_t stream) {
  DMD_CHECK_ARG(oihw && packed, "pack: null");
  DMD_CHECK_ARG((k == 1 || k == 3) && CoutPad >= Cout && CoutPad % 16 == 0 && CinPad >= Cin && CinPad % 16 == 0,
                "pack: bad sizes");
  const size_t total = (size_t)(CinPad / 16) * k * k * CoutPad * 16;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, oihw,
                     packed, Cout, Cin, k, CoutPad, CinPad);
  DMD_LAUNCH_CHECK();
  return 0;
}
