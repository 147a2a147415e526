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
//     FiLM + SiLU (one fma + v_exp/v_rcp) and the h/l split are applied while staging.
//   * weights: pre-split by dmd_pack_conv_weight_f16x2 into [chunk][tap][h|l][k group][64 cout][8 cin]
//     halfs; the 36 KiB of a chunk are copied linearly into LDS next to the patch.  Both the next
//     chunk's activations and its weights are prefetched into registers DURING the 9-tap MFMA loop,
//     which itself contains no vmcnt wait (in-order vmcnt would otherwise serialise every tap's weight
//     fetch behind the HBM-latency activation loads).
#include "dmd_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define F16S_CIN_MAX 128

// Compile-time timing ablations (development only; results are WRONG when non-zero):
//   1: no global stores / residual loads in the epilogue   2: no MFMA   4: no staging global loads
//   8: no staging math (norm/SiLU)   32: no tap loop at all (no fragment reads, no MFMA)
#ifndef F16S_ABL
#define F16S_ABL 0
#endif

template <bool B8_>
struct F16Geom {
  static constexpr bool B8 = B8_;
  static constexpr int SUB = B8_ ? 4 : 1;
  static constexpr int TS = B8_ ? 8 : 16;
  static constexpr int PW = TS + 2;
  static constexpr int PPS = PW * PW;
  static constexpr int NPP = SUB * PPS;
  static constexpr int ITEMS = (NPP * 4 + 255) / 256;
  static constexpr int W_UNITS = 9 * 2 * 2 * 64;  // 16-byte units of one chunk's weights: [tap][h|l][g][64 cout]
  static constexpr int OT_STRIDE = 68;                      // floats per pixel row of the epilogue's output tile
  static constexpr int MAIN_BYTES = NPP * 64 + W_UNITS * 16;  // K loop: patch + weights
  static constexpr int OT_BYTES = 256 * OT_STRIDE * 4;        // epilogue: [256 pixels][64 couts (+4 pad)] fp32
  static constexpr int BODY_BYTES = MAIN_BYTES > OT_BYTES ? MAIN_BYTES : OT_BYTES;
  static constexpr int SMEM_BYTES = BODY_BYTES + 2 * SUB * F16S_CIN_MAX * 4 + 256;
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

// SiLU for the split-precision path: v_exp_f32 / v_rcp_f32 (1 ulp each, i.e. the same 2^-22 class as the
// operand split itself) instead of the IEEE expf + division of the exact kernel.
__device__ __forceinline__ float f16s_silu(float t) {
  return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
}

template <class G>
__global__ __launch_bounds__(256, 2) void conv_f16s_kernel(const dmd_conv_params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* patch = (u32x4*)smem_raw;                               // [NPP][4] 16-byte slots
  u32x4* wlds = patch + G::NPP * 4;                              // [9][2][2][64] 16-byte units
  float* tab_a = (float*)(smem_raw + G::BODY_BYTES);             // [SUB][CIN_MAX]: t = v * a + b
  float* tab_b = tab_a + G::SUB * F16S_CIN_MAX;
  double* red = (double*)(tab_b + G::SUB * F16S_CIN_MAX);        // 32 doubles of scratch

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

  // ---- staging items (chunk invariant): thread -> (patch pixel, channel quad q) ----
  int goff[G::ITEMS];  // source pixel index, -1: zero
  int loff[G::ITEMS];  // 8-byte unit index of the h half-quad in the patch, -1: no item
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
  const int col = G::B8 ? (n31 & 7) : (n31 & 15);
  int posh[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) posh[dx] = (g + ((col + dx) >> 1)) & 3;
  // A operand: unit ((tap * 2 + piece) * 2 + g) * 64 + cout
  const int wunit = g * 64 + cb * 32 + n31;

  f32x16 acc[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;

  // global weights: [chunk][W_UNITS] 16-byte units, same order as the LDS copy
  const u32x4* wglob = (const u32x4*)p.w_f16;

  // Activations are prefetched TWO chunks ahead (two register sets): with one set the HBM latency of a chunk's
  // loads (2-3 us under load) is longer than the chunk's own MFMA loop and every chunk stalls at its end.
  f32x4 stage0[G::ITEMS], stage1[G::ITEMS];
  u32x4 wstage[9];  // this thread's 9 units of the next chunk's weights
  auto load_chunk = [&](int ck, f32x4 (&stage)[G::ITEMS]) {
    const int si = ck < nch0 ? 0 : 1;
    const dmd_conv_src& sc = p.src[si];
    const int c0 = (si ? ck - nch0 : ck) * 16 + 4 * q;
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
      // branch-free: out-of-image / padding items read pixel 0 and are zeroed in store_chunk
      const int go = goff[it] < 0 ? 0 : goff[it];
#if F16S_ABL & 4
      stage[it] = (f32x4){(float)go, 1.f, 2.f, (float)c0};
#else
      stage[it] = *(const f32x4*)(sc.x + (size_t)go * sc.C + c0);
#endif
    }
  };
  auto store_chunk = [&](int ck, const f32x4 (&stage)[G::ITEMS]) {
    const int si = ck < nch0 ? 0 : 1;
    const int prologue = p.src[si].prologue;
    const int cc = ck * 16 + 4 * q;
    uint2* pb = (uint2*)patch;
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
      f32x4 v = stage[it];
      if (prologue != DMD_PROLOGUE_NONE && !(F16S_ABL & 8)) {
        const float* ta = tab_a + isub[it] * F16S_CIN_MAX + cc;
        const float* tb = tab_b + isub[it] * F16S_CIN_MAX + cc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = __builtin_fmaf(v[e], ta[e], tb[e]);
          if (prologue == DMD_PROLOGUE_NORM_SILU) t = f16s_silu(t);
          v[e] = t;
        }
      }
      h4 hv, lv;
      const bool zero = goff[it] < 0;  // conv zero padding is applied AFTER the activation (blocks.py:143-144)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = zero ? 0.f : __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
        const _Float16 h = (_Float16)x;
        hv[e] = h;
        lv[e] = (_Float16)(x - (float)h);
      }
      if (loff[it] >= 0) {
        pb[loff[it]] = __builtin_bit_cast(uint2, hv);
        pb[loff[it] ^ 4] = __builtin_bit_cast(uint2, lv);  // l slot = h slot + 2 (mod 4): unit index bit 2
      }
    }
  };

  load_chunk(0, stage0);
#pragma unroll
  for (int k = 0; k < 9; ++k) wstage[k] = wglob[tid + 256 * k];
  if (nchunks > 1) load_chunk(1, stage1);

  // (the loads above are in flight while the normalisation tables are built: one exposed memory latency
  // instead of three dependent ones per tile)
  // ---- prologue tables: y = (x - mean) * rstd * mul' + add  ==  x * a + b ----
  // group statistics first, one (sub-tile, source, group) per wave iteration: lanes fetch the producer's
  // per-tile partial sums in parallel (fixed shuffle-tree order -> deterministic)
  {
    const int G0 = p.src[0].prologue != DMD_PROLOGUE_NONE ? C0 / DMD_GN_GROUP : 0;
    const int G1 = (p.nsrc > 1 && p.src[1].prologue != DMD_PROLOGUE_NONE) ? C1 / DMD_GN_GROUP : 0;
    const int GT = G0 + G1;  // <= 4
    float* gms = (float*)red;  // [SUB][4][2] mean, rstd
    const double count = (double)DMD_GN_GROUP * Hs * Ws;
    // FiLM / affine parameters of this thread's channel: issued BEFORE the statistics reduction so that the two
    // global-memory latencies overlap (C0 + C1 <= 128 < 256 threads: one channel per thread)
    float pmul[G::SUB], padd[G::SUB];
    {
      const int c = tid < C0 + C1 ? tid : 0;
      const int si = c < C0 ? 0 : 1;
      const dmd_norm& nm = p.src[si].norm;
      const int cl = si ? c - C0 : c;
      const bool on = p.src[si].prologue != DMD_PROLOGUE_NONE;
#pragma unroll
      for (int s = 0; s < G::SUB; ++s) {
        pmul[s] = (on && nm.mul) ? nm.mul[(size_t)ti[s].n * nm.mul_stride + cl] : 1.0f;
        padd[s] = (on && nm.add) ? nm.add[(size_t)ti[s].n * nm.add_stride + cl] : 0.0f;
      }
    }
    for (int item = wave; item < G::SUB * GT; item += 4) {
      const int s = item / GT, gi = item - s * GT;
      const int si = gi < G0 ? 0 : 1;
      const int gl = si ? gi - G0 : gi;
      const dmd_norm& nm = p.src[si].norm;
      const int Gs = p.src[si].C / DMD_GN_GROUP;
      F16Tile t = ti[0];
#pragma unroll
      for (int k = 1; k < G::SUB; ++k)
        if (s == k) t = ti[k];
      const double* st = nm.stats + ((size_t)(t.n * Gs + gl) * nm.stat_tiles) * 2;
      double a = 0.0, b = 0.0;
      for (int tt = lane; tt < nm.stat_tiles; tt += 64) {
        a += st[2 * tt];
        b += st[2 * tt + 1];
      }
      a = dmd_wave_sum(a);
      b = dmd_wave_sum(b);
      if (lane == 0) {
        const double m = a / count;
        double var = b / count - m * m;
        var = var < 0.0 ? 0.0 : var;
        gms[(s * 4 + gi) * 2] = (float)m;
        gms[(s * 4 + gi) * 2 + 1] = (float)(1.0 / sqrt(var + (double)DMD_GN_EPS));
      }
    }
    __syncthreads();
    for (int c = tid; c < C0 + C1; c += 256) {
      const int si = c < C0 ? 0 : 1;
      const dmd_conv_src& sc = p.src[si];
      const int cl = si ? c - C0 : c;
      const int gi = si ? G0 + cl / DMD_GN_GROUP : cl / DMD_GN_GROUP;
#pragma unroll
      for (int s = 0; s < G::SUB; ++s) {
        float a = 1.f, b = 0.f;
        if (sc.prologue != DMD_PROLOGUE_NONE && ti[s].valid) {
          float mul = pmul[s];
          if (sc.norm.mul_plus_one) mul = 1.0f + mul;
          a = gms[(s * 4 + gi) * 2 + 1] * mul;
          b = padd[s] - gms[(s * 4 + gi) * 2] * a;
        }
        tab_a[s * F16S_CIN_MAX + c] = a;
        tab_b[s * F16S_CIN_MAX + c] = b;
      }
    }
  }

  __syncthreads();  // tables visible
  store_chunk(0, stage0);
#pragma unroll
  for (int k = 0; k < 9; ++k) wlds[tid + 256 * k] = wstage[k];
  __syncthreads();

  // one chunk: `hold` carries chunk ck + 1 (loaded a chunk ago), `recv` receives chunk ck + 2
  auto chunk_body = [&](int ck, const f32x4 (&hold)[G::ITEMS], f32x4 (&recv)[G::ITEMS]) {
    const bool more = ck + 1 < nchunks;
    if (more) {
      // next chunk's weights first (L2, needed at the end of this chunk), then the activations of the chunk
      // after it: the vmcnt wait before store_chunk can leave those youngest loads in flight
      const u32x4* wnext = wglob + (size_t)(ck + 1) * G::W_UNITS + tid;
#pragma unroll
      for (int k = 0; k < 9; ++k) wstage[k] = wnext[256 * k];
    }
    if (ck + 2 < nchunks) load_chunk(ck + 2, recv);
#pragma unroll
    for (int tap = 0; tap < ((F16S_ABL & 32) ? 0 : 9); ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      // LDS fragment reads are issued in the order the MFMAs consume them (LDS returns in order), two pixel
      // blocks at a time: the first MFMAs start after 3 of the 10 reads have landed, the rest overlap.
      h8 bh[4], bl[4];
      const int toff = dy * G::PW + dx;
      const h8 ah = __builtin_bit_cast(h8, wlds[(tap * 2 + 0) * 128 + wunit]);
      bh[0] = __builtin_bit_cast(h8, patch[(pixbase[0] + toff) * 4 + posh[dx]]);
      bh[1] = __builtin_bit_cast(h8, patch[(pixbase[1] + toff) * 4 + posh[dx]]);
      bl[0] = __builtin_bit_cast(h8, patch[(pixbase[0] + toff) * 4 + (posh[dx] ^ 2)]);
      bl[1] = __builtin_bit_cast(h8, patch[(pixbase[1] + toff) * 4 + (posh[dx] ^ 2)]);
      const h8 al = __builtin_bit_cast(h8, wlds[(tap * 2 + 1) * 128 + wunit]);
      bh[2] = __builtin_bit_cast(h8, patch[(pixbase[2] + toff) * 4 + posh[dx]]);
      bh[3] = __builtin_bit_cast(h8, patch[(pixbase[3] + toff) * 4 + posh[dx]]);
      bl[2] = __builtin_bit_cast(h8, patch[(pixbase[2] + toff) * 4 + (posh[dx] ^ 2)]);
      bl[3] = __builtin_bit_cast(h8, patch[(pixbase[3] + toff) * 4 + (posh[dx] ^ 2)]);
#if F16S_ABL & 2
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[blk][tap] += (float)(bh[blk][0] + bl[blk][1] + ah[0] + al[1]);
#else
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int b0 = 2 * pr, b1 = 2 * pr + 1;
        acc[b0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[b0], acc[b0], 0, 0, 0);
        acc[b1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[b1], acc[b1], 0, 0, 0);
        acc[b0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[b0], acc[b0], 0, 0, 0);
        acc[b1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[b1], acc[b1], 0, 0, 0);
        acc[b0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[b0], acc[b0], 0, 0, 0);
        acc[b1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[b1], acc[b1], 0, 0, 0);
      }
#endif
    }
    if (more) {
      __syncthreads();  // every wave is done reading this chunk's patch and weights
      store_chunk(ck + 1, hold);
#pragma unroll
      for (int k = 0; k < 9; ++k) wlds[tid + 256 * k] = wstage[k];
      __syncthreads();
    }
  };
  for (int ck = 0; ck < nchunks; ck += 2) {
    chunk_body(ck, stage1, stage0);
    if (ck + 1 < nchunks) chunk_body(ck + 1, stage0, stage1);
  }

  // ---- epilogue ----
  // The MFMA result layout gives a lane 4 x 4 consecutive couts of ONE pixel: storing that directly writes 32-byte
  // fragments at a 256-byte stride.  Instead the 256 x 64 tile is transposed through LDS so that every global
  // access (residual read, output write) is a fully coalesced 1 KiB per wave instruction, and the GroupNorm
  // partial sums are formed on the final values in the same pass.
  __syncthreads();  // every wave is done with the patch / weights
  float* otile = (float*)smem_raw;
  {
    f32x4 bias[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      bias[qd] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias) bias[qd] = *(const f32x4*)(p.bias + cb * 32 + 8 * qd + 4 * g);
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      float* orow = otile + (ph * 128 + blk * 32 + n31) * G::OT_STRIDE + cb * 32 + 4 * g;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 v = (f32x4){acc[blk][4 * qd], acc[blk][4 * qd + 1], acc[blk][4 * qd + 2], acc[blk][4 * qd + 3]};
        *(f32x4*)(orow + 8 * qd) = v + bias[qd];
      }
    }
  }
  __syncthreads();
  // wave w re-reads pixels [64 w, 64 w + 64) of the tile (raster order inside a sub-tile); lane = (pixel lane >> 4,
  // channel quad lane & 15); GroupNorm group of the lane = (lane & 15) >> 3.
  {
    const int quad = lane & 15;
    F16Tile t = ti[0];
    if (G::B8) {
#pragma unroll
      for (int k = 1; k < G::SUB; ++k)
        if (wave == k) t = ti[k];
    }
    double dsum = 0.0, dsq = 0.0;
    if (t.valid) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float fs = 0.f, fq = 0.f;  // fp32 over 16 values, fp64 across (keeps E[x^2] - E[x]^2 accurate)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const int pl = wave * 64 + (jj * 4 + j4) * 4 + (lane >> 4);
          int oy, ox;
          if (G::B8) {
            const int r = pl & 63;
            oy = t.y0 + (r >> 3);
            ox = t.x0 + (r & 7);
          } else {
            oy = t.y0 + (pl >> 4);
            ox = t.x0 + (pl & 15);
          }
          const size_t off = (((size_t)t.n * p.H + oy) * p.W + ox) * 64 + 4 * quad;
          f32x4 v = *(const f32x4*)(otile + pl * G::OT_STRIDE + 4 * quad);
#if F16S_ABL & 1
          if (v[0] == 123.456f) *(f32x4*)(p.out + off) = v;
#else
          if (p.residual) v += *(const f32x4*)(p.residual + off);
          *(f32x4*)(p.out + off) = v;
#endif
          fs += (v[0] + v[1]) + (v[2] + v[3]);
          fq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        dsum += (double)fs;
        dsq += (double)fq;
      }
    }
    if (p.out_stats) {  // uniform
      // lanes of one group: bits {0,1,2} (quad within the group) and {4,5} (pixel lane)
#pragma unroll
      for (int m = 1; m <= 32; m <<= 1) {
        if (m == 8) continue;
        dsum += __shfl_xor(dsum, m, 64);
        dsq += __shfl_xor(dsq, m, 64);
      }
      if ((lane & 0x37) == 0) {  // lane 0 (group 0) and lane 8 (group 1)
        red[(wave * 2 + (lane >> 3)) * 2] = dsum;
        red[(wave * 2 + (lane >> 3)) * 2 + 1] = dsq;
      }
      __syncthreads();
      if (G::B8) {
        if (tid < 8) {  // (sub-tile, group)
          const int s = tid >> 1, gg = tid & 1;
          F16Tile ts = ti[0];
#pragma unroll
          for (int k = 1; k < G::SUB; ++k)
            if (s == k) ts = ti[k];
          if (ts.valid) {
            const int tx8 = p.W / 8;
            const int T = tx8 * (p.H / 8), tt = (ts.y0 / 8) * tx8 + ts.x0 / 8;
            double* o = p.out_stats + ((size_t)(ts.n * 2 + gg) * T + tt) * 2;
            o[0] = red[(s * 2 + gg) * 2];
            o[1] = red[(s * 2 + gg) * 2 + 1];
          }
        }
      } else if (tid < 4 && ti[0].valid) {  // (8-row half, group): waves 2h and 2h + 1
        const int h = tid >> 1, gg = tid & 1;
        const int tx16 = p.W / 16;
        const int T = tx16 * (p.H / 8), tt = (ti[0].y0 / 8 + h) * tx16 + ti[0].x0 / 16;
        double* o = p.out_stats + ((size_t)(ti[0].n * 2 + gg) * T + tt) * 2;
        o[0] = red[((2 * h) * 2 + gg) * 2] + red[((2 * h + 1) * 2 + gg) * 2];
        o[1] = red[((2 * h) * 2 + gg) * 2 + 1] + red[((2 * h + 1) * 2 + gg) * 2 + 1];
      }
    }
  }
}

// OIHW fp32 -> [CinPad/16][9][h|l][k group g = 0|1][Cout][8 cin] halfs  (cin = 16 chunk + 8 g + e), Cout in {32, 64}:
// one chunk = 36 * Cout contiguous 16-byte units, copied linearly into LDS by the kernels.
__global__ void pack_weight_f16x2_kernel(const float* __restrict__ oihw, _Float16* __restrict__ packed, int Cout, int Cin,
                                         int CinPad, int taps) {
  const size_t total = (size_t)(CinPad / 16) * taps * Cout * 16;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx % 8;
  const int co = (idx / 8) % Cout;
  const int g = (idx / (8 * (size_t)Cout)) % 2;
  const int tap = (idx / (8 * (size_t)Cout * 2)) % taps;
  const int chunk = idx / (8 * (size_t)Cout * 2 * taps);
  const int c = chunk * 16 + g * 8 + e;
  float v = 0.f;
  if (c < Cin) v = oihw[((size_t)co * Cin + c) * taps + tap];
  v = fminf(fmaxf(v, -65504.0f), 65504.0f);
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  const size_t base = ((((size_t)chunk * taps + tap) * 2 + 0) * 2 + g) * ((size_t)Cout * 8) + (size_t)co * 8 + e;
  packed[base] = h;
  packed[base + 2 * (size_t)Cout * 8] = l;
}

extern "C" int dmd_pack_conv_weight_f16x2(const float* oihw, void* packed, int Cout, int Cin, int k, int CinPad,
                                          dmd_stream_t stream) {
  DMD_CHECK_ARG(oihw && packed, "pack_f16x2: null");
  DMD_CHECK_ARG((Cout == 64 || Cout == 32) && (k == 3 || k == 1) && CinPad >= Cin && CinPad % 16 == 0,
                "pack_f16x2: needs Cout in {32, 64} (got %d), k in {1, 3}, CinPad %% 16 == 0", Cout);
  const int taps = k * k;
  const size_t total = (size_t)(CinPad / 16) * taps * Cout * 16;
  hipLaunchKernelGGL(pack_weight_f16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, oihw,
                     (_Float16*)packed, Cout, Cin, CinPad, taps);
  DMD_LAUNCH_CHECK();
  return 0;
}

// 1: the parameters can run on the split-fp16 kernels (Cout = 64: both kernels; Cout = 32: conv_f16ws only)
extern "C" int dmd_conv2d_f16x2_eligible(const dmd_conv_params* p) {
  if (!p || (p->precision & 0xff) != DMD_PRECISION_F16X2 || !p->w_f16) return 0;
  if (p->stride != 1 || p->residual_norm.stats || (p->taps != 9 && p->taps != 1)) return 0;
  if (p->taps == 1 && p->upsample) return 0;
  // few-channel NCHW head (conv_out): Cout <= 4 zero-padded to 32, no residual / statistics
  const bool head = p->out_nchw && p->Cout <= 4 && p->CoutPad == 32 && !p->residual && !p->out_stats;
  if (!head && ((p->Cout != 64 && p->Cout != 32) || p->CoutPad != p->Cout || p->out_nchw)) return 0;
  int cin = 0;
  for (int i = 0; i < p->nsrc; ++i) cin += p->src[i].C;
  if (cin > (p->CoutPad == 64 ? F16S_CIN_MAX : 64)) return 0;
  const bool a16 = p->H % 16 == 0 && p->W % 16 == 0;
  const bool b8 = p->W % 16 != 0;
  return (a16 || b8) ? 1 : 0;
}

template <class G>
static int launch_f16s(const dmd_conv_params& p, int grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::SMEM_BYTES);
    DMD_CHECK_ARG(e == hipSuccess, "conv_f16s: hipFuncSetAttribute(%d bytes): %s", G::SMEM_BYTES, hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_f16s_kernel<G>), dim3(grid), dim3(256), G::SMEM_BYTES, st, p);
  return 0;
}

int dmd_launch_conv_f16s(const dmd_conv_params& p, hipStream_t st) {
  if (p.W % 16 != 0) return launch_f16s<F16Geom<true>>(p, (p.N * (p.H / 8) * (p.W / 8) + 3) / 4, st);
  return launch_f16s<F16Geom<false>>(p, p.N * (p.H / 16) * (p.W / 16), st);
}
