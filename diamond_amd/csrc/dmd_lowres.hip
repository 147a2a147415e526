// lowres_chain_kernel -- the whole 8x8 level of the denoiser's U-Net (reference blocks.py:232-246: the deepest
// d_blocks, the two mid blocks with attention, the deepest u_blocks = 7 ResBlocks, 14 3x3 convolutions, 3 skip
// projections, 2 attention blocks at the default configuration) in ONE launch, one workgroup per image.
//
// Why: at 8x8 a convolution of the whole batch is 1.2 GFLOP; launched on its own it is a ~25 us latency chain (launch,
// GroupNorm tables, pipeline fill, 4-8 chunk steps, write-out) on 64 of the 256 CUs, and the level is 14 such launches
// plus 1x1s and attention per denoiser call: 520 us for 1.5 % of the FLOPs (HISTORY.md).  Here every activation of the
// chain (64 pixels x 64 channels fp32 = 16 KiB) stays in LDS, GroupNorm statistics are two-group reductions inside the
// workgroup, and the only global traffic is the chain input / output and the (L2-resident) weights.
//
// Arithmetic: the split-fp16 form of dmd_conv_f16ws.hip (x = h + l, three v_mfma_f32_32x32x16_f16 per product, fp32
// accumulate, same packed weights, same v_exp/v_rcp SiLU); GroupNorm sums leave the epilogue registers through the DPP
// network (fp32 inside a wave, fp64 across waves).  Work split of the 64-channel kernel: wave (cb, kh) accumulates all 64
// pixels x the 32 couts of block cb over the K chunks of parity kh; the two K halves meet in LDS.
// LDS: 5 activation slots (3 saved skips, X, H) + the halo'd split patch of one conv input (2 sources x 100 px x 64 ch,
// the slot-rotated layout of dmd_conv_f16ws.hip: conflict-free ds_read_b128 for every tap) -- overlaid by q | k | v
// during attention and by the K-split exchange after a convolution's MFMAs -- = 139 KiB.
// lowres_chain32_kernel (further down) is the 32-channel variant for the reward / end encoder's tail.
#include "dmd_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LR_SS 68                    // floats per pixel row of a slot (64 + 4: staggers the banks of consecutive pixels)
#define LR_SLOT (64 * LR_SS)
#define LR_NSLOT 5                  // 0..2: saved skips, 3: X (current), 4: H (intermediate)
#define LR_X 3
#define LR_H 4
#define LR_P_UNITS (2 * 100 * 4 * 4)  // 16-byte units: [source][patch pixel][16-channel chunk][rotated position]
#define LR_QS 196                   // floats per pixel row of the q | k | v overlay
// DMD_LAB builds only (tools/chain_bench.py; never the shipped library): LR_ABL = timing proxies with WRONG results (1 = no weight
// loads, 2 = no attention core, 4 = no statistics pass, 8 = no FiLM table loads), LR_TRACE = s_memtime stamps of workgroup 0
#ifndef DMD_LAB
#undef LR_ABL
#undef LR_TRACE
#endif
#ifndef LR_ABL
#define LR_ABL 0
#endif
#ifndef LR_TRACE
#define LR_TRACE 0
#endif
#if LR_TRACE
__device__ unsigned long long lr_trace_buf[512];
__device__ int lr_trace_n;
#define LR_STAMP(tag_)                                                                              \
  do {                                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0 && lr_ti < 512)                                         \
      lr_trace_buf[lr_ti++] = (__builtin_readcyclecounter() << 8) | (unsigned long long)(tag_);   \
  } while (0)
extern "C" int dmd_lowres_trace_dump(unsigned long long* host) {
  hipDeviceSynchronize();
  int n = 0;
  hipMemcpyFromSymbol(&n, HIP_SYMBOL(lr_trace_n), sizeof(int));
  hipMemcpyFromSymbol(host, HIP_SYMBOL(lr_trace_buf), sizeof(unsigned long long) * 512);
  return n;
}
#else
#define LR_STAMP(tag_) do {} while (0)
#endif
#define LR_SMEM_BYTES ((LR_NSLOT * LR_SLOT + LR_P_UNITS * 4 + 256 + 32) * 4 + 64)

__device__ __forceinline__ float lr_silu(float t) {
  return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
}

// 8 fp32 values -> split-fp16 pieces (no clamp: the range contract of dmd_conv_f16ws.hip)
__device__ __forceinline__ void lr_split8(const f32x4 a, const f32x4 b, h8& h, h8& l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 ha = (_Float16)a[e], hb = (_Float16)b[e];
    h[e] = ha;
    h[4 + e] = hb;
    l[e] = (_Float16)(a[e] - (float)ha);
    l[4 + e] = (_Float16)(b[e] - (float)hb);
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// wave64 sum on the DPP network (no LDS round trips): quad swaps and row mirrors leave the 16-lane row total in every
// lane of a row, row_bcast:15 / row_bcast:31 chain the four rows; the total is read from lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float lr_dpp_add(float x) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float lr_wave_sum(float x) {
  x = lr_dpp_add<0xB1, 0xf>(x);   // quad_perm [1, 0, 3, 2]
  x = lr_dpp_add<0x4E, 0xf>(x);   // quad_perm [2, 3, 0, 1]
  x = lr_dpp_add<0x141, 0xf>(x);  // row_half_mirror
  x = lr_dpp_add<0x140, 0xf>(x);  // row_mirror
  x = lr_dpp_add<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3
  x = lr_dpp_add<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

__global__ __launch_bounds__(256) void lowres_chain_kernel(const dmd_lowres_chain_params p) {
  DMD_DYNAMIC_LDS(unsigned char, lr_smem);
  float* slots = (float*)lr_smem;
  u32x4* P = (u32x4*)(slots + LR_NSLOT * LR_SLOT);
  float* QKV = (float*)P;               // overlay during attention
  float* SCR = (float*)P;               // overlay after a convolution's MFMAs: [wave][16][64] K-split partial tiles
  float* tabA = (float*)(P + LR_P_UNITS);
  float* tabB = tabA + 128;
  float* stat = tabB + 128;             // [slot][group][mean, rstd]
  double* red = (double*)(stat + 32);   // [wave][sum, sum of squares]

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, g = lane >> 5;
  // A wave accumulates ALL 64 pixels (two 32-pixel blocks) x the 32 couts of block cb over the K chunks of parity kh
  // (each weight fragment feeds two pixel blocks, half the L2 weight traffic of a pixel split, and twice the MFMA work
  // behind every fragment fetch); the two K halves are added through LDS and wave (cb, kh) finishes pixel block kh.
  const int cb = wave & 1, kh = wave >> 1;
  const int pxl = 32 * kh + n31;  // the pixel this lane finishes (epilogue)
  const float* trow = p.table + (size_t)n * p.table_stride;
  const int wlane = g * 64 + cb * 32 + n31;  // this lane's 16-byte unit inside a (tap, h|l) weight row

  auto slot = [&](int s) -> float* { return slots + s * LR_SLOT; };
#if LR_TRACE
  int lr_ti = 0;
#endif

  // ---- mean / rstd of the two GroupNorm groups from the per-wave partial sums in `red` (wave = (cb, kh): group cb) ----
  auto finish_stats = [&](int s, int also) {
    __syncthreads();
    if (tid < 2) {
      const double sum = red[2 * tid] + red[2 * (tid + 2)], ssq = red[2 * tid + 1] + red[2 * (tid + 2) + 1];
      const double m = sum / 2048.0;
      double var = ssq / 2048.0 - m * m;
      var = var < 0.0 ? 0.0 : var;
      const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)DMD_GN_EPS));
      stat[(s * 2 + tid) * 2] = mean;
      stat[(s * 2 + tid) * 2 + 1] = rstd;
      if (also >= 0) {
        stat[(also * 2 + tid) * 2] = mean;
        stat[(also * 2 + tid) * 2 + 1] = rstd;
      }
    }
    __syncthreads();
  };
  // statistics of a slot that no epilogue produced (the chain input): wave (cb, kh) sums channels [32 cb + 16 kh, +16)
  auto slot_stats = [&](int s, int also) {
    __syncthreads();
    const float* r = slot(s) + lane * LR_SS + 32 * cb + 16 * kh;
    float fs = 0.f, fq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(const f32x4*)(r + 4 * q);
      fs += (v[0] + v[1]) + (v[2] + v[3]);
      fq = __builtin_fmaf(v[0], v[0], fq);
      fq = __builtin_fmaf(v[1], v[1], fq);
      fq = __builtin_fmaf(v[2], v[2], fq);
      fq = __builtin_fmaf(v[3], v[3], fq);
    }
    const float a = lr_wave_sum(fs), b = lr_wave_sum(fq);
    if (lane == 0) {
      red[2 * wave] = (double)a;
      red[2 * wave + 1] = (double)b;
    }
    finish_stats(s, also);
  };

  // ---- per-channel (a, b) of y = x * a + b for the sources of a convolution; (mul, add) = this thread's FiLM entries ----
  auto tab_film = [&](int nsrc, int s0, int s1, float mul, float add) {
    if (tid < 64 * nsrc) {
      const int src = tid >> 6, ch = tid & 63, sl = src ? s1 : s0;
      const float mean = stat[(sl * 2 + (ch >> 5)) * 2], rstd = stat[(sl * 2 + (ch >> 5)) * 2 + 1];
      const float a = rstd * (1.0f + mul);  // AdaGroupNorm: xn * (1 + scale) + shift (blocks.py:41-45)
      tabA[tid] = a;
      tabB[tid] = add - mean * a;
    }
    __syncthreads();
  };
  auto tab_affine = [&](int s, const float* gamma, const float* beta) {
    if (tid < 64) {
      const float mean = stat[(s * 2 + (tid >> 5)) * 2], rstd = stat[(s * 2 + (tid >> 5)) * 2 + 1];
      const float a = rstd * gamma[tid];
      tabA[tid] = a;
      tabB[tid] = beta[tid] - mean * a;
    }
    __syncthreads();
  };

  // ---- SiLU(norm(x)) of the sources, split and zero-padded, into the halo'd patch ----
  // item it * 256 + tid = (patch pixel pp = it * 16 + (tid >> 4), channel quad tid & 15): the same for every convolution
  int soff[7];  // float offset of the item's source pixel inside a slot, -1: zero padding (halo), -2: no item
  int poff[7];  // 8-byte unit of the item's h piece inside a source's patch; the l piece sits 4 units (one rotation of 2) away
  {
    const int q16 = tid & 15, chunk = q16 >> 2, q = q16 & 3;
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int pp = it * 16 + (tid >> 4);
      const int py = pp / 10, px = pp - 10 * py;
      const bool inside = pp < 100 && py >= 1 && py <= 8 && px >= 1 && px <= 8;
      soff[it] = inside ? ((py - 1) * 8 + (px - 1)) * LR_SS + 4 * q16 : (pp < 100 ? -1 : -2);
      const int pos = ((q >> 1) + (px >> 1)) & 3;
      poff[it] = ((pp * 4 + chunk) * 4 + pos) * 2 + (q & 1);
    }
  }
  auto stage_patch = [&](int nsrc, int s0, int s1) {
    uint2* P8 = (uint2*)P;
    for (int src = 0; src < nsrc; ++src) {
      const float* sp = slot(src ? s1 : s0);
      const f32x4 ta = *(const f32x4*)(tabA + src * 64 + 4 * (tid & 15)), tb = *(const f32x4*)(tabB + src * 64 + 4 * (tid & 15));
#pragma unroll
      for (int it = 0; it < 7; ++it) {
        if (soff[it] == -2) continue;  // beyond the patch (last item row)
        h4 hv = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
        if (soff[it] >= 0) {
          const f32x4 v = *(const f32x4*)(sp + soff[it]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = lr_silu(__builtin_fmaf(v[e], ta[e], tb[e]));
            const _Float16 h = (_Float16)x;
            hv[e] = h;
            lv[e] = (_Float16)(x - (float)h);
          }
        }
        const int u = src * 3200 + poff[it];
        P8[u] = __builtin_bit_cast(uint2, hv);
        P8[u ^ 4] = __builtin_bit_cast(uint2, lv);
      }
    }
    __syncthreads();
  };

  // ---- 3x3 convolution of the staged patch: acc[blk] += W * patch over this wave's K chunks (kh, kh + 2, ...).
  // Weights: 18 fragments (9 taps x h | l) per chunk straight from L2 into registers, one of the wave's chunks ahead;
  // its first chunk is prefetched into `wpre` by the caller BEFORE the statistics / table / staging phases that precede
  // the MFMA loop (one wave per SIMD: nothing else hides an L2 round trip) ----
  u32x4 wpre[18];
  auto load_w = [&](const void* w16, int ck, u32x4 (&w)[18]) {
    const u32x4* wg = (const u32x4*)w16 + wlane;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
#if LR_ABL & 1
      w[i] = (u32x4){0x3c003c00u, (unsigned)(ck + i), 0x3c003c00u, (unsigned)lane};
#else
      w[i] = wg[(size_t)(ck * 18 + i) * 128];
#endif
    }
  };
  auto prefetch_w = [&](const void* w16) { load_w(w16, kh, wpre); };
  const int prow = n31 >> 3, pcol = n31 & 7;  // this lane's pixel inside a 32-pixel block (4 rows of 8)
  auto conv3x3 = [&](f32x16 (&acc)[2], int nck, const void* w16) {
    u32x4 wb[18];
    auto compute = [&](int ck, const u32x4 (&w)[18]) {
      const int src = ck >> 2, chunk = ck & 3;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        const int pos = (g + ((pcol + dx) >> 1)) & 3;
        const int pp0 = (prow + dy) * 10 + pcol + dx;  // block 0: rows 0..3; block 1: 4 rows (40 patch pixels) further
        const int base0 = ((src * 100 + pp0) * 4 + chunk) * 4, base1 = base0 + 40 * 16;
        const h8 bh0 = __builtin_bit_cast(h8, P[base0 + pos]), bh1 = __builtin_bit_cast(h8, P[base1 + pos]);
        const h8 bl0 = __builtin_bit_cast(h8, P[base0 + (pos ^ 2)]), bl1 = __builtin_bit_cast(h8, P[base1 + (pos ^ 2)]);
        const h8 ah = __builtin_bit_cast(h8, w[2 * tap]), al = __builtin_bit_cast(h8, w[2 * tap + 1]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, acc[1], 0, 0, 0);
      }
    };
    for (int ck = kh; ck < nck; ck += 4) {  // nck = 4 | 8; wpre holds chunk ck on entry
      load_w(w16, ck + 2, wb);              // (ck + 2 < nck always: nck is a multiple of 4)
      compute(ck, wpre);
      if (ck + 4 < nck) load_w(w16, ck + 4, wpre);
      compute(ck + 2, wb);
    }
  };

  // ---- 1x1 convolution of a raw fp32 slot (64 channels = 4 chunks, row stride `ss`): acc[blk] += W[ck0 + c] * x over the
  // chunks c of parity kh, operands split on the fly ----
  auto conv1x1 = [&](f32x16 (&acc)[2], const float* x, int ss, const void* w16, int ck0) {
    const u32x4* wg = (const u32x4*)w16 + wlane;
    u32x4 w[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
#if LR_ABL & 1
        w[2 * i + hl] = (u32x4){0x3c003c00u, (unsigned)(ck0 + i), 0x3c003c00u, (unsigned)lane};
#else
        w[2 * i + hl] = wg[(size_t)((ck0 + kh + 2 * i) * 2 + hl) * 128];
#endif
      }
    }
    const float* xr = x + n31 * ss + 8 * g;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = kh + 2 * i;
      const h8 ah = __builtin_bit_cast(h8, w[2 * i]), al = __builtin_bit_cast(h8, w[2 * i + 1]);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        h8 bh, bl;
        const float* xp = xr + blk * 32 * ss + 16 * chunk;
        lr_split8(*(const f32x4*)xp, *(const f32x4*)(xp + 4), bh, bl);
        acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[blk], 0, 0, 0);
        acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[blk], 0, 0, 0);
        acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[blk], 0, 0, 0);
      }
    }
  };

  // ---- K-split reduction + epilogue.  Every wave has finished its MFMAs (and with them its reads of the patch, which the
  // scratch overlays): wave (cb, kh) hands the tile of pixel block 1 - kh to its partner through LDS, adds the partner's
  // tile of block kh (always as partial(kh = 0) + partial(kh = 1)), then bias / residual / store; `scr`: 16 KiB of LDS nobody
// reads at that point (the patch region, or the X slot while q | k | v are being built); lane owns couts
  // cb*32 + 8 qd + 4 g + (0..3) of pixel pxl.  want_stats: GroupNorm sums of the written tensor (the wave's 32 couts are
  // exactly group cb) -> stat[stat_slot]. ----
  auto finish = [&](f32x16 (&acc)[2], float* dst, int ds, int c0, const float* bias_a, const float* bias_b, const float* resid,
                    float* scr, int stat_slot) {
    __syncthreads();  // every wave is past its MFMAs / the previous user of the scratch
    f32x4* mine = (f32x4*)(scr + (size_t)wave * 1024);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      mine[qd * 64 + lane] = kh ? (f32x4){acc[0][4 * qd], acc[0][4 * qd + 1], acc[0][4 * qd + 2], acc[0][4 * qd + 3]}
                                : (f32x4){acc[1][4 * qd], acc[1][4 * qd + 1], acc[1][4 * qd + 2], acc[1][4 * qd + 3]};
    __syncthreads();
    const f32x4* theirs = (const f32x4*)(scr + (size_t)(wave ^ 2) * 1024);
    f32x16 v16;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4 other = theirs[qd * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float own = kh ? acc[1][4 * qd + e] : acc[0][4 * qd + e];
        v16[4 * qd + e] = kh ? other[e] + own : own + other[e];
      }
    }
    float fs = 0.f, fq = 0.f;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int c = cb * 32 + 8 * qd + 4 * g;
      f32x4 v = (f32x4){v16[4 * qd], v16[4 * qd + 1], v16[4 * qd + 2], v16[4 * qd + 3]};
      if (bias_a) v += *(const f32x4*)(bias_a + c);
      if (bias_b) v += *(const f32x4*)(bias_b + c);
      if (resid) v += *(const f32x4*)(resid + pxl * LR_SS + c);
      *(f32x4*)(dst + pxl * ds + c0 + c) = v;
      fs += (v[0] + v[1]) + (v[2] + v[3]);
      fq = __builtin_fmaf(v[0], v[0], fq);
      fq = __builtin_fmaf(v[1], v[1], fq);
      fq = __builtin_fmaf(v[2], v[2], fq);
      fq = __builtin_fmaf(v[3], v[3], fq);
    }
    if (stat_slot >= 0) {
#if !(LR_ABL & 4)
      const float a = lr_wave_sum(fs), b = lr_wave_sum(fq);
      if (lane == 0) {
        red[2 * wave] = (double)a;
        red[2 * wave + 1] = (double)b;
      }
#endif
      finish_stats(stat_slot, -1);
    }
  };
  auto zero = [&](f32x16 (&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  };

  // FiLM entries of this thread for a block: (scale, shift) of norm1 (tid < 128: source tid >> 6, channel tid & 63) and of
  // norm2 (tid < 64), fetched one block ahead (a table row is touched for the first time here: HBM latency)
  auto film_fetch = [&](int b, float (&f)[4]) {
    f[0] = f[1] = f[2] = f[3] = 0.f;
    if (b >= p.nblocks) return;
    const dmd_chain_block& B = p.blocks[b];
#if !(LR_ABL & 8)
    if (tid < 128) {
      const int src = tid >> 6, ch = tid & 63;
      if (src == 0 || B.skip_slot >= 0) {
        f[0] = trow[B.film1_mul[src] + ch];
        f[1] = trow[B.film1_add[src] + ch];
      }
      if (src == 0) {
        f[2] = trow[B.film2_mul + ch];
        f[3] = trow[B.film2_add + ch];
      }
    }
#endif
  };

  // =============================== the chain ===============================
  float* X = slot(LR_X);
  float* H = slot(LR_H);
  float film_cur[4], film_nxt[4];
  film_fetch(0, film_nxt);
  prefetch_w(p.blocks[0].w1);
  for (int id = tid; id < 64 * 16; id += 256) {
    const int px = id >> 4, q = id & 15;
    const f32x4 v = *(const f32x4*)(p.x + ((size_t)n * 64 + px) * 64 + 4 * q);
    *(f32x4*)(X + px * LR_SS + 4 * q) = v;
    if (p.input_save_slot >= 0) *(f32x4*)(slot(p.input_save_slot) + px * LR_SS + 4 * q) = v;
  }
  slot_stats(LR_X, p.input_save_slot);
  LR_STAMP(0);

  for (int b = 0; b < p.nblocks; ++b) {
    const dmd_chain_block& B = p.blocks[b];
    const int nsrc = B.skip_slot >= 0 ? 2 : 1;
    const int sk = B.skip_slot >= 0 ? B.skip_slot : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) film_cur[e] = film_nxt[e];
    film_fetch(b + 1, film_nxt);
    f32x16 acc[2];
    // ---- conv1(SiLU(AdaGN1(cat(x, skip)))) -> H  (its first weights were prefetched during the previous block) ----
    LR_STAMP(1);
    tab_film(nsrc, LR_X, sk, film_cur[0], film_cur[1]);
    LR_STAMP(2);
    stage_patch(nsrc, LR_X, sk);
    LR_STAMP(3);
    zero(acc);
    conv3x3(acc, 4 * nsrc, B.w1);
    LR_STAMP(4);
    prefetch_w(B.w2);
    finish(acc, H, LR_SS, 0, B.b1, nullptr, nullptr, SCR, LR_H);
    LR_STAMP(6);
    // ---- conv2(SiLU(AdaGN2(h))) + r -> X,  r = proj(cat(x, skip)) (accumulated first) or x ----
    zero(acc);
    if (B.wproj) {
      conv1x1(acc, X, LR_SS, B.wproj, 0);
      conv1x1(acc, slot(sk), LR_SS, B.wproj, 4);
    }
    LR_STAMP(7);
    tab_film(1, LR_H, 0, film_cur[2], film_cur[3]);
    stage_patch(1, LR_H, 0);  // (its barriers also order the projection's reads of X before the epilogue's writes)
    LR_STAMP(8);
    conv3x3(acc, 4, B.w2);
    LR_STAMP(9);
    if (!B.has_attn && b + 1 < p.nblocks) prefetch_w(p.blocks[b + 1].w1);
    finish(acc, X, LR_SS, 0, B.b2, B.wproj ? B.bproj : nullptr, B.wproj ? nullptr : X, SCR, LR_X);
    LR_STAMP(10);
    // ---- SelfAttention2d: x_n = GN(x); y = softmax(q k^T / sqrt 8) v per head; x = x_n + out_proj(y) ----
    if (B.has_attn) {
      tab_affine(LR_X, B.gn_gamma, B.gn_beta);
      for (int id = tid; id < 64 * 16; id += 256) {
        const int px = id >> 4, q = id & 15;
        const f32x4 v = *(const f32x4*)(X + px * LR_SS + 4 * q);
        const f32x4 ta = *(const f32x4*)(tabA + 4 * q), tb = *(const f32x4*)(tabB + 4 * q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(v[e], ta[e], tb[e]);
        *(f32x4*)(H + px * LR_SS + 4 * q) = o;
      }
      __syncthreads();  // x_n complete
      LR_STAMP(11);
      // q, k, v -> the q | k | v overlay of the patch region; their K-split reductions go through the X slot (x is dead:
      // its normalised copy is in H, and y is written only after the attention core)
      const void* wqkv[3] = {B.wq, B.wk, B.wv};
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        zero(acc);
        conv1x1(acc, H, LR_SS, wqkv[part], 0);
        finish(acc, QKV, LR_QS, 64 * part, B.bqkv + 64 * part, nullptr, nullptr, X, -1);
      }
      __syncthreads();
      LR_STAMP(12);
      // one (head, query) per thread and pass: 8 heads x 64 queries; a wave = the 64 queries of one head (K / V rows broadcast)
      for (int it = 0; it < ((LR_ABL & 2) ? 0 : 2); ++it) {
        const int head = 4 * it + wave, qi = lane;
        const float* qp = QKV + qi * LR_QS + head * 8;
        const f32x4 q0 = *(const f32x4*)qp, q1 = *(const f32x4*)(qp + 4);
        const float inv = 1.0f / sqrtf(8.0f);  // (q k^T) / sqrt(d), blocks.py:68
        // Variants measured (per attention block, one wave per SIMD: the K / V rows are wave-uniform LDS reads whose latency is
        // the cost): this row-at-a-time loop 11.5 us; v_pk_fma_f32 dot products 14 us; rows fetched 16 at a time 9 us, but
        // the extra live registers spill into the convolution loops of the same kernel (+11 us there).
        float sc[64];  // the row of scores stays in registers (fully unrolled loops)
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          const float* kp = QKV + j * LR_QS + 64 + head * 8;
          const f32x4 k0 = *(const f32x4*)kp, k1 = *(const f32x4*)(kp + 4);
          float sv = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) sv = __builtin_fmaf(q0[e], k0[e], sv);
#pragma unroll
          for (int e = 0; e < 4; ++e) sv = __builtin_fmaf(q1[e], k1[e], sv);
          sv *= inv;
          sc[j] = sv;
          m = fmaxf(m, sv);
        }
        float l = 0.f;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          const float pr = __builtin_amdgcn_exp2f((sc[j] - m) * 1.4426950408889634f);
          l += pr;
          const float* vp = QKV + j * LR_QS + 128 + head * 8;
          const f32x4 v0 = *(const f32x4*)vp, v1 = *(const f32x4*)(vp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = __builtin_fmaf(pr, v0[e], o0[e]);
            o1[e] = __builtin_fmaf(pr, v1[e], o1[e]);
          }
        }
        const float rl = 1.0f / l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] *= rl;
          o1[e] *= rl;
        }
        *(f32x4*)(X + qi * LR_SS + head * 8) = o0;  // y overwrites x (dead: its normalised copy is in H)
        *(f32x4*)(X + qi * LR_SS + head * 8 + 4) = o1;
      }
      __syncthreads();  // y complete; q | k | v are dead (the scratch may overwrite them)
      LR_STAMP(13);
      zero(acc);
      conv1x1(acc, X, LR_SS, B.wo, 0);
      if (b + 1 < p.nblocks) prefetch_w(p.blocks[b + 1].w1);
      finish(acc, X, LR_SS, 0, B.bo, nullptr, H, SCR, LR_X);  // (its first barrier: every wave has read y)
      LR_STAMP(14);
    }
    if (B.save_slot >= 0) {
      for (int id = tid; id < 64 * 16; id += 256) {
        const int px = id >> 4, q = id & 15;
        *(f32x4*)(slot(B.save_slot) + px * LR_SS + 4 * q) = *(const f32x4*)(X + px * LR_SS + 4 * q);
      }
      if (tid < 4) stat[B.save_slot * 4 + tid] = stat[LR_X * 4 + tid];
      __syncthreads();
    }
  }
  for (int id = tid; id < 64 * 16; id += 256) {
    const int px = id >> 4, q = id & 15;
    *(f32x4*)(p.out + ((size_t)n * 64 + px) * 64 + 4 * q) = *(const f32x4*)(X + px * LR_SS + 4 * q);
  }
  LR_STAMP(15);
#if LR_TRACE
  if (blockIdx.x == 0 && threadIdx.x == 0) lr_trace_n = lr_ti;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// lowres_chain32_kernel -- the same idea for the reward / end model's encoder (reference rew_end_model.py:93-133): its last
// two ResBlocks groups run at 8x8 with 32 channels (one GroupNorm group), no concatenated skips, attention (4 heads) in
// the final group.  Wave (pb, kh) = 32-pixel block pb x all 32 couts over the K chunk kh (a 32-channel input is two
// 16-channel chunks); the two K halves are exchanged through LDS and wave kh finishes cout quads {2 kh, 2 kh + 1}.
// ------------------------------------------------------------------------------------------------------------------
#define L3_SS 36                     // floats per pixel row of a slot (32 + 4)
#define L3_SLOT (64 * L3_SS)
#define L3_REGION_FLOATS (64 * 100)  // patch (100 px x 2 chunks x 4 units x 4 floats = 3200) | q k v overlay [64][100] | scratch
#define L3_SMEM_BYTES ((2 * L3_SLOT + L3_REGION_FLOATS + 64 + 32) * 4 + 64)

__global__ __launch_bounds__(256) void lowres_chain32_kernel(const dmd_lowres_chain_params p) {
  DMD_DYNAMIC_LDS(unsigned char, lr_smem);
  float* X = (float*)lr_smem;
  float* H = X + L3_SLOT;
  u32x4* P = (u32x4*)(H + L3_SLOT);
  float* QKV = (float*)P;  // [64][100]: q | k | v (96) + 4 pad
  float* SCR = (float*)P;  // [wave][8][64] K-split partial half tiles
  float* tabA = (float*)P + L3_REGION_FLOATS;
  float* tabB = tabA + 32;
  float* stat = tabB + 32;            // [0]: X (mean, rstd), [1]: H
  double* red = (double*)(stat + 32);

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, g = lane >> 5;
  const int pb = wave & 1, kh = wave >> 1;
  const int pxl = 32 * pb + n31;
  const int prow = 4 * pb + (n31 >> 3), pcol = n31 & 7;
  const float* trow = p.table + (size_t)n * p.table_stride;
  const int wlane = g * 32 + n31;  // 16-byte unit inside a (tap, h|l) weight row of a 32-cout pack

  auto finish_stats = [&](int s) {
    __syncthreads();
    if (tid == 0) {
      const double sum = (red[0] + red[2]) + (red[4] + red[6]), ssq = (red[1] + red[3]) + (red[5] + red[7]);
      const double m = sum / 2048.0;
      double var = ssq / 2048.0 - m * m;
      var = var < 0.0 ? 0.0 : var;
      stat[2 * s] = (float)m;
      stat[2 * s + 1] = (float)(1.0 / sqrt(var + (double)DMD_GN_EPS));
    }
    __syncthreads();
  };
  auto publish_sums = [&](float fs, float fq) {
    const float a = lr_wave_sum(fs), b = lr_wave_sum(fq);
    if (lane == 0) {
      red[2 * wave] = (double)a;
      red[2 * wave + 1] = (double)b;
    }
  };
  auto tab_film = [&](int s, float mul, float add) {
    if (tid < 32) {
      const float a = stat[2 * s + 1] * (1.0f + mul);
      tabA[tid] = a;
      tabB[tid] = add - stat[2 * s] * a;
    }
    __syncthreads();
  };
  auto tab_affine = [&](int s, const float* gamma, const float* beta) {
    if (tid < 32) {
      const float a = stat[2 * s + 1] * gamma[tid];
      tabA[tid] = a;
      tabB[tid] = beta[tid] - stat[2 * s] * a;
    }
    __syncthreads();
  };

  // staging items: it * 256 + tid = (patch pixel pp = it * 32 + (tid >> 3), channel quad q8 = tid & 7)
  int soff[4], poff[4];
  {
    const int q8 = tid & 7, chunk = q8 >> 2, q = q8 & 3;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pp = it * 32 + (tid >> 3);
      const int py = pp / 10, px = pp - 10 * py;
      const bool inside = pp < 100 && py >= 1 && py <= 8 && px >= 1 && px <= 8;
      soff[it] = inside ? ((py - 1) * 8 + (px - 1)) * L3_SS + 4 * q8 : (pp < 100 ? -1 : -2);
      const int pos = ((q >> 1) + (px >> 1)) & 3;
      poff[it] = ((pp * 2 + chunk) * 4 + pos) * 2 + (q & 1);
    }
  }
  auto stage_patch = [&](const float* sp) {
    uint2* P8 = (uint2*)P;
    const f32x4 ta = *(const f32x4*)(tabA + 4 * (tid & 7)), tb = *(const f32x4*)(tabB + 4 * (tid & 7));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (soff[it] == -2) continue;
      h4 hv = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
      if (soff[it] >= 0) {
        const f32x4 v = *(const f32x4*)(sp + soff[it]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = lr_silu(__builtin_fmaf(v[e], ta[e], tb[e]));
          const _Float16 h = (_Float16)x;
          hv[e] = h;
          lv[e] = (_Float16)(x - (float)h);
        }
      }
      P8[poff[it]] = __builtin_bit_cast(uint2, hv);
      P8[poff[it] ^ 4] = __builtin_bit_cast(uint2, lv);
    }
    __syncthreads();
  };

  u32x4 wpre[18];  // this wave's chunk (kh) of the next 3x3 convolution, fetched ahead of the phases that precede its use
  auto prefetch_w = [&](const void* w16) {
    const u32x4* wg = (const u32x4*)w16 + wlane;
#pragma unroll
    for (int i = 0; i < 18; ++i) wpre[i] = wg[(size_t)(kh * 18 + i) * 64];
  };
  auto conv3x3 = [&](f32x16& acc) {  // acc += W[chunk kh] * patch, weights in wpre
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const int pp = (prow + dy) * 10 + pcol + dx;
      const int base = (pp * 2 + kh) * 4;
      const int pos = (g + ((pcol + dx) >> 1)) & 3;
      const h8 bh = __builtin_bit_cast(h8, P[base + pos]), bl = __builtin_bit_cast(h8, P[base + (pos ^ 2)]);
      const h8 ah = __builtin_bit_cast(h8, wpre[2 * tap]), al = __builtin_bit_cast(h8, wpre[2 * tap + 1]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
  };
  auto conv1x1 = [&](f32x16& acc, const float* x, const void* w16) {  // acc += W[chunk kh] * x, raw fp32 slot, split on the fly
    const u32x4* wg = (const u32x4*)w16 + wlane;
    const h8 ah = __builtin_bit_cast(h8, wg[(size_t)(kh * 2) * 64]), al = __builtin_bit_cast(h8, wg[(size_t)(kh * 2 + 1) * 64]);
    const float* xp = x + pxl * L3_SS + 16 * kh + 8 * g;
    h8 bh, bl;
    lr_split8(*(const f32x4*)xp, *(const f32x4*)(xp + 4), bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  };
  // K-split exchange + epilogue: wave (pb, kh) finishes cout quads qd = 2 kh, 2 kh + 1 (couts 8 qd + 4 g + (0..3)) of pixel pxl
  auto finish = [&](const f32x16& acc, float* dst, int ds, int c0, const float* bias, const float* resid, float* scr, int stat_slot) {
    __syncthreads();
    f32x4* mine = (f32x4*)(scr + (size_t)wave * 512);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qd = 2 * (1 - kh) + i;  // the quads the partner finishes
      mine[i * 64 + lane] = (f32x4){acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
    }
    __syncthreads();
    const f32x4* theirs = (const f32x4*)(scr + (size_t)(wave ^ 2) * 512);
    float fs = 0.f, fq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qd = 2 * kh + i;
      const int c = 8 * qd + 4 * g;
      const f32x4 other = theirs[i * 64 + lane];
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = kh ? other[e] + acc[4 * qd + e] : acc[4 * qd + e] + other[e];  // partial(kh = 0) + partial(kh = 1)
      if (bias) v += *(const f32x4*)(bias + c);
      if (resid) v += *(const f32x4*)(resid + pxl * L3_SS + c);
      *(f32x4*)(dst + pxl * ds + c0 + c) = v;
      fs += (v[0] + v[1]) + (v[2] + v[3]);
      fq = __builtin_fmaf(v[0], v[0], fq);
      fq = __builtin_fmaf(v[1], v[1], fq);
      fq = __builtin_fmaf(v[2], v[2], fq);
      fq = __builtin_fmaf(v[3], v[3], fq);
    }
    if (stat_slot >= 0) {
      publish_sums(fs, fq);
      finish_stats(stat_slot);
    }
  };
  auto zero = [&](f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };
  auto film_fetch = [&](int b, float (&f)[4]) {
    f[0] = f[1] = f[2] = f[3] = 0.f;
    if (b >= p.nblocks || tid >= 32) return;
    const dmd_chain_block& B = p.blocks[b];
    f[0] = trow[B.film1_mul[0] + tid];
    f[1] = trow[B.film1_add[0] + tid];
    f[2] = trow[B.film2_mul + tid];
    f[3] = trow[B.film2_add + tid];
  };

  float film_cur[4], film_nxt[4];
  film_fetch(0, film_nxt);
  prefetch_w(p.blocks[0].w1);
  {
    float fs = 0.f, fq = 0.f;
    for (int id = tid; id < 64 * 8; id += 256) {
      const int px = id >> 3, q = id & 7;
      const f32x4 v = *(const f32x4*)(p.x + ((size_t)n * 64 + px) * 32 + 4 * q);
      *(f32x4*)(X + px * L3_SS + 4 * q) = v;
      fs += (v[0] + v[1]) + (v[2] + v[3]);
      fq = __builtin_fmaf(v[0], v[0], fq);
      fq = __builtin_fmaf(v[1], v[1], fq);
      fq = __builtin_fmaf(v[2], v[2], fq);
      fq = __builtin_fmaf(v[3], v[3], fq);
    }
    publish_sums(fs, fq);
    finish_stats(0);
  }
  for (int b = 0; b < p.nblocks; ++b) {
    const dmd_chain_block& B = p.blocks[b];
#pragma unroll
    for (int e = 0; e < 4; ++e) film_cur[e] = film_nxt[e];
    film_fetch(b + 1, film_nxt);
    f32x16 acc;
    // conv1(SiLU(AdaGN1(x))) -> H
    tab_film(0, film_cur[0], film_cur[1]);
    stage_patch(X);
    zero(acc);
    conv3x3(acc);
    prefetch_w(B.w2);
    finish(acc, H, L3_SS, 0, B.b1, nullptr, SCR, 1);
    // conv2(SiLU(AdaGN2(h))) + x -> X
    tab_film(1, film_cur[2], film_cur[3]);
    stage_patch(H);
    zero(acc);
    conv3x3(acc);
    if (b + 1 < p.nblocks) prefetch_w(p.blocks[b + 1].w1);
    finish(acc, X, L3_SS, 0, B.b2, X, SCR, 0);
    if (B.has_attn) {
      tab_affine(0, B.gn_gamma, B.gn_beta);
      for (int id = tid; id < 64 * 8; id += 256) {
        const int px = id >> 3, q = id & 7;
        const f32x4 v = *(const f32x4*)(X + px * L3_SS + 4 * q);
        const f32x4 ta = *(const f32x4*)(tabA + 4 * q), tb = *(const f32x4*)(tabB + 4 * q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(v[e], ta[e], tb[e]);
        *(f32x4*)(H + px * L3_SS + 4 * q) = o;
      }
      __syncthreads();
      const void* wqkv[3] = {B.wq, B.wk, B.wv};
#pragma unroll
      for (int part = 0; part < 3; ++part) {  // q | k | v -> the overlay; the K-split exchange goes through the (dead) X slot
        zero(acc);
        conv1x1(acc, H, wqkv[part]);
        finish(acc, QKV, 100, 32 * part, B.bqkv + 32 * part, nullptr, X, -1);
      }
      __syncthreads();
      {  // 4 heads x 64 queries: wave = head, lane = query
        const int head = wave, qi = lane;
        const float* qp = QKV + qi * 100 + head * 8;
        const f32x4 q0 = *(const f32x4*)qp, q1 = *(const f32x4*)(qp + 4);
        const float inv = 1.0f / sqrtf(8.0f);
        float sc[64];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          const float* kp = QKV + j * 100 + 32 + head * 8;
          const f32x4 k0 = *(const f32x4*)kp, k1 = *(const f32x4*)(kp + 4);
          float sv = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) sv = __builtin_fmaf(q0[e], k0[e], sv);
#pragma unroll
          for (int e = 0; e < 4; ++e) sv = __builtin_fmaf(q1[e], k1[e], sv);
          sv *= inv;
          sc[j] = sv;
          m = fmaxf(m, sv);
        }
        float l = 0.f;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          const float pr = __builtin_amdgcn_exp2f((sc[j] - m) * 1.4426950408889634f);
          l += pr;
          const float* vp = QKV + j * 100 + 64 + head * 8;
          const f32x4 v0 = *(const f32x4*)vp, v1 = *(const f32x4*)(vp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = __builtin_fmaf(pr, v0[e], o0[e]);
            o1[e] = __builtin_fmaf(pr, v1[e], o1[e]);
          }
        }
        const float rl = 1.0f / l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] *= rl;
          o1[e] *= rl;
        }
        *(f32x4*)(X + qi * L3_SS + head * 8) = o0;
        *(f32x4*)(X + qi * L3_SS + head * 8 + 4) = o1;
      }
      __syncthreads();
      zero(acc);
      conv1x1(acc, X, B.wo);
      finish(acc, X, L3_SS, 0, B.bo, H, SCR, 0);  // its first barrier: every wave has read y
    }
  }
  for (int id = tid; id < 64 * 8; id += 256) {
    const int px = id >> 3, q = id & 7;
    *(f32x4*)(p.out + ((size_t)n * 64 + px) * 32 + 4 * q) = *(const f32x4*)(X + px * L3_SS + 4 * q);
  }
}

extern "C" int dmd_lowres_chain(const dmd_lowres_chain_params* p, dmd_stream_t stream) {
  DMD_CHECK_ARG(p && p->x && p->out && p->table, "lowres_chain: null");
  DMD_CHECK_ARG(p->N > 0 && p->nblocks > 0 && p->nblocks <= DMD_CHAIN_MAX_BLOCKS, "lowres_chain: N %d, nblocks %d", p->N, p->nblocks);
  DMD_CHECK_ARG(p->input_save_slot >= -1 && p->input_save_slot <= 2, "lowres_chain: input_save_slot");
  for (int b = 0; b < p->nblocks; ++b) {
    const dmd_chain_block& B = p->blocks[b];
    DMD_CHECK_ARG(B.w1 && B.w2 && B.b1 && B.b2, "lowres_chain: block %d: null conv weights", b);
    DMD_CHECK_ARG(B.skip_slot >= -1 && B.skip_slot <= 2 && B.save_slot >= -1 && B.save_slot <= 2, "lowres_chain: block %d: slots", b);
    DMD_CHECK_ARG((B.skip_slot >= 0) == (B.wproj != nullptr), "lowres_chain: block %d: a concatenated input needs its 1x1 projection (and only it)", b);
    DMD_CHECK_ARG(!B.wproj || B.bproj, "lowres_chain: block %d: projection bias", b);
    DMD_CHECK_ARG(!B.has_attn || (B.gn_gamma && B.gn_beta && B.wq && B.wk && B.wv && B.wo && B.bqkv && B.bo), "lowres_chain: block %d: attention parameters", b);
  }
  static bool attr_set[DMD_MAX_DEVICES] = {};  // the attribute belongs to the DEVICE the launch goes to
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DMD_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lowres_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LR_SMEM_BYTES);
    DMD_CHECK_ARG(e == hipSuccess, "lowres_chain: hipFuncSetAttribute(%d bytes): %s", (int)LR_SMEM_BYTES, hipGetErrorString(e));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(lowres_chain_kernel, dim3(p->N), dim3(256), LR_SMEM_BYTES, (hipStream_t)stream, *p);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_lowres_chain32(const dmd_lowres_chain_params* p, dmd_stream_t stream) {
  DMD_CHECK_ARG(p && p->x && p->out && p->table, "lowres_chain32: null");
  DMD_CHECK_ARG(p->N > 0 && p->nblocks > 0 && p->nblocks <= DMD_CHAIN_MAX_BLOCKS, "lowres_chain32: N %d, nblocks %d", p->N, p->nblocks);
  for (int b = 0; b < p->nblocks; ++b) {
    const dmd_chain_block& B = p->blocks[b];
    DMD_CHECK_ARG(B.w1 && B.w2 && B.b1 && B.b2, "lowres_chain32: block %d: null conv weights", b);
    DMD_CHECK_ARG(B.skip_slot < 0 && B.save_slot < 0 && !B.wproj, "lowres_chain32: block %d: no concatenated inputs at 32 channels", b);
    DMD_CHECK_ARG(!B.has_attn || (B.gn_gamma && B.gn_beta && B.wq && B.wk && B.wv && B.wo && B.bqkv && B.bo), "lowres_chain32: block %d: attention parameters", b);
  }
  static bool attr_set[DMD_MAX_DEVICES] = {};  // the attribute belongs to the DEVICE the launch goes to
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DMD_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lowres_chain32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L3_SMEM_BYTES);
    DMD_CHECK_ARG(e == hipSuccess, "lowres_chain32: hipFuncSetAttribute(%d bytes): %s", (int)L3_SMEM_BYTES, hipGetErrorString(e));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(lowres_chain32_kernel, dim3(p->N), dim3(256), L3_SMEM_BYTES, (hipStream_t)stream, *p);
  DMD_LAUNCH_CHECK();
  return 0;
}
