// conv_f16s_kernel -- the 64-output-channel 3x3 stride-1 convolutions of the U-Net (92 % of the
// denoiser's FLOPs) on the f16 matrix cores with SPLIT fp32 operands.
//
// Why: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, dmd_conv.hip) runs at the fp32 vector rate
// (157 TFLOP/s chip peak); the f16 MFMA is 16x faster.  Every fp32 operand x is split as
//     x = h + l + e,   h = fp16(x),  l = fp16(x - h),   |e| <= max(2^-22 |x|, 2^-25)
// (gfx950's MFMA honours fp16 subnormals, tools/probe/mfma_f16_probe.hip, so l needs no scaling)
// and a product is evaluated as  w_h*x_h + w_h*x_l + w_l*x_h  -- three v_mfma_f32_32x32x16_f16
// into ONE fp32 accumulator; the dropped w_l*x_l term is 2^-22 relative.  Result: fp32-class
// accuracy (measured ~2e-7 of the output scale, same order as oneDNN-vs-fp64) at an effective peak of
// 2.5 PFLOP/s / 3.  Requires |x| < 65504 (post-GroupNorm/SiLU activations and residual streams
// are O(1..100)); inputs are clamped to the fp16 range so an outlier saturates instead of
// producing inf.  dmd_conv_params.precision selects this kernel; the exact-fp32 kernel remains
// the default for everything that feeds gradients.
//
// Work decomposition (256 threads = 4 wave64):
//   * workgroup tile = 256 output pixels x 64 output channels.  CFG A16: one 16x16 patch of one
//     image (H, W multiples of 16); CFG B8: four 8x8 patches (8x8 maps / W not a multiple of 16).
//   * wave (cb, ph) = 32 output channels (== one GroupNorm group) x 128 pixels (== one statistics
//     tile of dmd_conv_stat_tiles): 4 MFMA blocks of 32 couts x 32 pixels, fp32 accumulators
//     f32x16 acc[4].  GEMM view D[cout][pixel]: weights = A operand, pixels = B operand, so a lane
//     ends with 4 x 4 consecutive couts of one pixel -> 16-byte NHWC stores.
//   * K loop: 16 input channels per step (= MFMA K).  The halo'd patch of a chunk is staged once
//     in LDS as [patch pixel][4 x 16 B]: {h[0:8], h[8:16], l[0:8], l[8:16]}, slot rotated by
//     (px >> 1) -> conflict-free ds_read_b128 for every tap (tools/lds_sim.py rules); GroupNorm /
//     FiLM + SiLU and the h/l split are applied while staging; double buffered.
//   * weights: pre-split on the host side of the ABI (dmd_pack_conv_weight_f16x2), streamed from
//     L2 into registers one (chunk, tap) step ahead: [chunk][tap][h|l][64 cout][16 cin] halfs.
#include "dmd_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define F16S_CIN_MAX 128

template <bool B8_>
struct F16Geom {
  static constexpr bool B8 = B8_;
  static constexpr int SUB = B8_ ? 4 : 1;
  static constexpr int TS = B8_ ? 8 : 16;
  static constexpr int PW = TS + 2;
  static constexpr int PPS = PW * PW;
  static constexpr int NPP = SUB * PPS;
  static constexpr int ITEMS = (NPP * 4 + 255) / 256;
};

struct F16Tile {
  int n, y0, x0;
  bool valid;
};

template <class G>
__device__ __forceinline__ F16Tile f16_subtile(const dmd_conv_params& p, int tile, int s) {
  const int tx = p.W / G::TS, per_img = tx * (p.H / G::TS);
  const int gs = tile * G::SUB + s;
  F16Tile t;
  t.valid = gs < p.N * per_img;
  const int g2 = t.valid ? gs : 0;
  t.n = g2 / per_img;
  const int r = g2 - t.n * per_img;
  const int ty = r / tx;
  t.y0 = ty * G::TS;
  t.x0 = (r - ty * tx) * G::TS;
  return t;
}

__device__ __forceinline__ float f16_clamp(float v) { return fminf(fmaxf(v, -65504.0f), 65504.0f); }

template <class G>
__global__ __launch_bounds__(256, 2) void conv_f16s_kernel(const dmd_conv_params p) {
  __shared__ u32x4 patch[2][G::NPP * 4];
  __shared__ float tab_mean[G::SUB][F16S_CIN_MAX];
  __shared__ float tab_a[G::SUB][F16S_CIN_MAX];
  __shared__ float tab_add[G::SUB][F16S_CIN_MAX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave & 1;   // 32-cout block == GroupNorm group
  const int ph = wave >> 1;  // pixel half of the workgroup tile
  const int n31 = lane & 31, g = lane >> 5;
  const int tile = blockIdx.x;

  F16Tile ti[G::SUB];
#pragma unroll
  for (int s = 0; s < G::SUB; ++s) ti[s] = f16_subtile<G>(p, tile, s);

  const int up = p.upsample;
  const int Hs = p.H >> up, Ws = p.W >> up;
  const int C0 = p.src[0].C;
  const int C1 = p.nsrc > 1 ? p.src[1].C : 0;
  const int nch0 = C0 >> 4;
  const int nchunks = (C0 + C1) >> 4;

  // ---- prologue tables ----
  for (int c = tid; c < C0 + C1; c += 256) {
    const int si = c < C0 ? 0 : 1;
    const dmd_conv_src& sc = p.src[si];
    const int cl = si ? c - C0 : c;
#pragma unroll
    for (int s = 0; s < G::SUB; ++s) {
      float m = 0.f, a = 1.f, ad = 0.f;
      if (sc.prologue != DMD_PROLOGUE_NONE && ti[s].valid)
        norm_entry(sc.norm, ti[s].n, cl, sc.C, (double)DMD_GN_GROUP * Hs * Ws, &m, &a, &ad);
      tab_mean[s][c] = m;
      tab_a[s][c] = a;
      tab_add[s][c] = ad;
    }
  }

  // ---- staging items (chunk invariant): thread -> (patch pixel, channel quad q) ----
  int goff[G::ITEMS];  // source pixel index, -1: zero
  int loff[G::ITEMS];  // 8-byte unit index of the h half-quad in a patch buffer, -1: no item
  int isub[G::ITEMS];
  const int q = tid & 3;
#pragma unroll
  for (int it = 0; it < G::ITEMS; ++it) {
    const int id = it * 256 + tid;
    const int pp = id >> 2;
    const bool ok = pp < G::NPP;
    const int s = G::SUB == 1 ? 0 : (ok ? pp / G::PPS : 0);
    const int rem = pp - s * G::PPS;
    const int py = rem / G::PW, px = rem - py * G::PW;
    F16Tile t = ti[0];
#pragma unroll
    for (int k = 1; k < G::SUB; ++k)
      if (s == k) t = ti[k];
    const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
    const bool inb = ok && t.valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    goff[it] = inb ? ((t.n * Hs + (iy >> up)) * Ws + (ix >> up)) : -1;
    // slot of h[8*(q>>1) ..] = ((q >> 1) + (px >> 1)) & 3; 8-byte unit inside the slot = q & 1
    loff[it] = ok ? (pp * 4 + (((q >> 1) + (px >> 1)) & 3)) * 2 + (q & 1) : -1;
    isub[it] = s;
  }

  // ---- B-operand addressing: lane = (pixel n31 of a 32-pixel block, k group g) ----
  int pixbase[4];
  int col;
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    if (G::B8) {
      const int s = ph * 2 + (blk >> 1);
      const int row = (blk & 1) * 4 + (n31 >> 3);
      pixbase[blk] = s * G::PPS + row * G::PW + (n31 & 7);
    } else {
      const int row = ph * 8 + blk * 2 + (n31 >> 4);
      pixbase[blk] = row * G::PW + (n31 & 15);
    }
  }
  col = G::B8 ? (n31 & 7) : (n31 & 15);
  int posh[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) posh[dx] = (g + ((col + dx) >> 1)) & 3;

  f32x16 acc[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;

  // weights: [chunk][tap][piece][64][16] halfs; lane -> cout cb*32 + n31, k offset 8 g
  const _Float16* wlane = (const _Float16*)p.w_f16 + (size_t)(cb * 32 + n31) * 16 + 8 * g;
  constexpr size_t WPIECE = 64 * 16, WSTEP = 2 * WPIECE;

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
  auto store_chunk = [&](int ck, int buf) {
    const int si = ck < nch0 ? 0 : 1;
    const int prologue = p.src[si].prologue;
    const int cc = ck * 16 + 4 * q;
    uint2* pb = (uint2*)&patch[buf][0];
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
      f32x4 v = stage[it];
      if (prologue != DMD_PROLOGUE_NONE && goff[it] >= 0) {
        const int s = isub[it];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (v[e] - tab_mean[s][cc + e]) * tab_a[s][cc + e] + tab_add[s][cc + e];
          if (prologue == DMD_PROLOGUE_NORM_SILU) t = dmd_silu(t);
          v[e] = t;
        }
      }
      h4 hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = f16_clamp(v[e]);
        const _Float16 h = (_Float16)x;
        hv[e] = h;
        lv[e] = (_Float16)(x - (float)h);
      }
      if (loff[it] >= 0) {
        pb[loff[it]] = __builtin_bit_cast(uint2, hv);
        pb[loff[it] ^ 4] = __builtin_bit_cast(uint2, lv);  // l slots = h slot + 2 (mod 4): unit index bit 2
      }
    }
  };

  __syncthreads();  // tables visible
  load_chunk(0);
  store_chunk(0, 0);
  h8 wh = *(const h8*)wlane;
  h8 wl = *(const h8*)(wlane + WPIECE);
  __syncthreads();

  for (int ck = 0; ck < nchunks; ++ck) {
    const int buf = ck & 1;
    const bool more = ck + 1 < nchunks;
    if (more) load_chunk(ck + 1);
    const u32x4* pbuf = &patch[buf][0];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const h8 ah = wh, al = wl;
      {
        const int step = ck * 9 + tap + 1;
        if (step < nchunks * 9) {
          wh = *(const h8*)(wlane + (size_t)step * WSTEP);
          wl = *(const h8*)(wlane + (size_t)step * WSTEP + WPIECE);
        }
      }
      h8 bh[4], bl[4];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int pix = pixbase[blk] + dy * G::PW + dx;
        bh[blk] = __builtin_bit_cast(h8, pbuf[pix * 4 + posh[dx]]);
        bl[blk] = __builtin_bit_cast(h8, pbuf[pix * 4 + (posh[dx] ^ 2)]);
      }
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[blk], acc[blk], 0, 0, 0);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[blk], acc[blk], 0, 0, 0);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[blk], acc[blk], 0, 0, 0);
    }
    if (more) store_chunk(ck + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns couts cb*32 + 8 qd + 4 g + (0..3), qd = 0..3, of pixel n31 of each block ----
  f32x4 bias[4];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    bias[qd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias[qd] = *(const f32x4*)(p.bias + cb * 32 + 8 * qd + 4 * g);
  }
  double ssum[2] = {0.0, 0.0}, ssq[2] = {0.0, 0.0};  // B8: [0] blocks 0-1, [1] blocks 2-3
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    F16Tile t;
    int oy, ox;
    if (G::B8) {
      const int s = ph * 2 + (blk >> 1);
      t = ti[0];
#pragma unroll
      for (int k = 1; k < G::SUB; ++k)
        if (s == k) t = ti[k];
      oy = t.y0 + (blk & 1) * 4 + (n31 >> 3);
      ox = t.x0 + (n31 & 7);
    } else {
      t = ti[0];
      oy = t.y0 + ph * 8 + blk * 2 + (n31 >> 4);
      ox = t.x0 + (n31 & 15);
    }
    if (!t.valid) continue;
    const size_t pixel = ((size_t)t.n * p.H + oy) * p.W + ox;
    float* op = p.out + pixel * 64 + cb * 32 + 4 * g;
    const float* rp = p.residual ? p.residual + pixel * 64 + cb * 32 + 4 * g : nullptr;
    const int slot = G::B8 ? (blk >> 1) : 0;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      f32x4 v = (f32x4){acc[blk][4 * qd], acc[blk][4 * qd + 1], acc[blk][4 * qd + 2], acc[blk][4 * qd + 3]};
      v += bias[qd];
      if (rp) v += *(const f32x4*)(rp + 8 * qd);
      *(f32x4*)(op + 8 * qd) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        ssum[slot] += d;
        ssq[slot] += d * d;
      }
    }
  }
  if (p.out_stats) {
    const int Gt = 2;  // Cout / 32
#pragma unroll
    for (int k = 0; k < (G::B8 ? 2 : 1); ++k) {
      const double a = dmd_wave_sum(ssum[k]);
      const double b = dmd_wave_sum(ssq[k]);
      F16Tile t = ti[0];
      int T, tt;
      if (G::B8) {
        const int s = ph * 2 + k;
#pragma unroll
        for (int kk = 1; kk < G::SUB; ++kk)
          if (s == kk) t = ti[kk];
        const int tx8 = p.W / 8;
        T = tx8 * (p.H / 8);
        tt = (t.y0 / 8) * tx8 + t.x0 / 8;
      } else {
        const int tx16 = p.W / 16;
        T = tx16 * (p.H / 8);
        tt = (t.y0 / 8 + ph) * tx16 + t.x0 / 16;
      }
      if (lane == 0 && t.valid) {
        double* o = p.out_stats + ((size_t)(t.n * Gt + cb) * T + tt) * 2;
        o[0] = a;
        o[1] = b;
      }
    }
  }
}

// OIHW fp32 -> [CinPad/16][9][h|l][64][16] halfs
__global__ void pack_weight_f16x2_kernel(const float* __restrict__ oihw, _Float16* __restrict__ packed, int Cout, int Cin,
                                         int CinPad) {
  const size_t total = (size_t)(CinPad / 16) * 9 * 64 * 16;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % 16;
  const int co = (idx / 16) % 64;
  const int tap = (idx / (16 * 64)) % 9;
  const int chunk = idx / (16 * 64 * 9);
  const int c = chunk * 16 + ci;
  float v = 0.f;
  if (co < Cout && c < Cin) v = oihw[((size_t)co * Cin + c) * 9 + tap];
  v = fminf(fmaxf(v, -65504.0f), 65504.0f);
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  const size_t base = (((size_t)chunk * 9 + tap) * 2) * (64 * 16) + (size_t)co * 16 + ci;
  packed[base] = h;
  packed[base + 64 * 16] = l;
}

extern "C" int dmd_pack_conv_weight_f16x2(const float* oihw, void* packed, int Cout, int Cin, int CinPad, dmd_stream_t stream) {
  DMD_CHECK_ARG(oihw && packed, "pack_f16x2: null");
  DMD_CHECK_ARG(Cout == 64 && CinPad >= Cin && CinPad % 16 == 0, "pack_f16x2: needs Cout == 64 (got %d), CinPad %% 16 == 0", Cout);
  const size_t total = (size_t)(CinPad / 16) * 9 * 64 * 16;
  hipLaunchKernelGGL(pack_weight_f16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, oihw,
                     (_Float16*)packed, Cout, Cin, CinPad);
  DMD_LAUNCH_CHECK();
  return 0;
}

// 1: the parameters can run on conv_f16s_kernel
extern "C" int dmd_conv2d_f16x2_eligible(const dmd_conv_params* p) {
  if (!p || p->precision != DMD_PRECISION_F16X2 || !p->w_f16) return 0;
  if (p->taps != 9 || p->stride != 1 || p->Cout != 64 || p->CoutPad != 64 || p->out_nchw) return 0;
  if (p->residual_norm.stats) return 0;
  int cin = 0;
  for (int i = 0; i < p->nsrc; ++i) cin += p->src[i].C;
  if (cin > F16S_CIN_MAX) return 0;
  const bool a16 = p->H % 16 == 0 && p->W % 16 == 0;
  const bool b8 = p->W % 16 != 0;
  return (a16 || b8) ? 1 : 0;
}

int dmd_launch_conv_f16s(const dmd_conv_params& p, hipStream_t st) {
  if (p.W % 16 != 0) {
    using G = F16Geom<true>;
    const int sub = p.N * (p.H / 8) * (p.W / 8);
    hipLaunchKernelGGL((conv_f16s_kernel<G>), dim3((sub + 3) / 4), dim3(256), 0, st, p);
  } else {
    using G = F16Geom<false>;
    hipLaunchKernelGGL((conv_f16s_kernel<G>), dim3(p.N * (p.H / 16) * (p.W / 16)), dim3(256), 0, st, p);
  }
  return 0;
}
