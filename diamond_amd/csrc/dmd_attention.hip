// dmd_attention -- softmax(q k^T / sqrt(d)) v for SelfAttention2d (models/blocks.py:62-72),
// flash-style: K/V tiles staged in LDS, online softmax, QK^T and PV on
// v_mfma_f32_16x16x4_f32 (head_dim d = 8, ATTN_HEAD_DIM blocks.py:14).
//
// Layout trick: compute the TRANSPOSED score block S^T[key][query] = K Q^T so that a lane
// (j = lane & 15, kg = lane >> 4) owns 4 keys {4 kg + r} of ONE query j:
//   * the softmax statistics of query j live in the 4 lanes {j, j+16, j+32, j+48}
//     (two __shfl_xor to combine),
//   * those same 4 registers are exactly the MFMA B operand of O^T[d][query] += V^T P^T
//     (k-remap: MFMA t contracts keys {4 k' + t}), so P never moves between lanes,
//   * O^T's D layout again has the query in the column (lane & 15): the running rescale
//     exp(m_old - m_new) is a per-lane scalar.
// One workgroup = 4 waves = 64 queries of one (image, head); keys stream in tiles of 256.
#include <stdlib.h>

#include "dmd_common.h"

#define ATT_KB 256          // keys per LDS tile
#define ATT_VSTRIDE (ATT_KB + 16)

// Wp > 0 (dmd_attention_valid): the T tokens are a row-major (T / Wp) x Wp grid of which only rows < hv, columns < wv
// exist; the other keys get the score -inf (token 0 always exists, so the running maximum is finite from the first block on).
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                        int C, float inv_scale_den, int Wp, int hv, int wv) {
  __shared__ float Ks[ATT_KB][8];
  __shared__ float Vt[8][ATT_VSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const size_t row = (size_t)3 * C;
  const float* base = qkv + (size_t)n * T * row;

  // Q fragment (B operand of S^T = K Q^T): lane (query j, k' = kg), step s uses dim 2 kg + s
  const float* qp = base + (size_t)(q0 + j) * row + h * 8 + 2 * kg;
  const float qa = qp[0], qb = qp[1];

  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};  // O^T[dd = 4 kg + r][query j]  (kg >= 2: padding rows)
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt = 0; kt < T; kt += ATT_KB) {
    const int nk = (T - kt) < ATT_KB ? (T - kt) : ATT_KB;
    __syncthreads();
    if (tid < nk) {
      const float* kp = base + (size_t)(kt + tid) * row + C + h * 8;
      const float* vp = kp + C;
      const f32x4 k0 = *(const f32x4*)kp, k1 = *(const f32x4*)(kp + 4);
      const f32x4 v0 = *(const f32x4*)vp, v1 = *(const f32x4*)(vp + 4);
      *(f32x4*)&Ks[tid][0] = k0;
      *(f32x4*)&Ks[tid][4] = k1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Vt[e][tid] = v0[e];
        Vt[4 + e][tid] = v1[e];
      }
    }
    __syncthreads();
    for (int k0 = 0; k0 < nk; k0 += 16) {
      // S^T block: A = K[key i = lane & 15][dim], B = Q[dim][query j]
      const float ka = Ks[k0 + j][2 * kg], kb = Ks[k0 + j][2 * kg + 1];
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, qa, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kb, qb, s, 0, 0, 0);
      // s[r] = q_j . k_{k0 + 4 kg + r}
      float bmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[r] = s[r] / inv_scale_den;  // (q @ k^T) / sqrt(d), blocks.py:68
        if (Wp > 0) {
          const int key = kt + k0 + 4 * kg + r, ky = key / Wp, kx = key - ky * Wp;
          if (ky >= hv || kx >= wv) s[r] = -INFINITY;
        }
        bmax = fmaxf(bmax, s[r]);
      }
      bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
      bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
      const float m_new = fmaxf(m_run, bmax);
      const float alpha = expf(m_run - m_new);
      f32x4 pr;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = expf(s[r] - m_new);
        psum += pr[r];
      }
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      l_run = alpha * l_run + psum;
      m_run = m_new;
      acc *= alpha;
      // O^T += V^T P^T : A = V^T[dd i = lane & 15][key 4 kg + t], B = P^T[key][query j] = pr[t]
      f32x4 vf = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (j < 8) vf = *(const f32x4*)&Vt[j][k0 + 4 * kg];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[t], pr[t], acc, 0, 0, 0);
    }
  }
  if (kg < 2) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = acc[r] / l_run;
    *(f32x4*)(out + ((size_t)n * T + q0 + j) * C + h * 8 + 4 * kg) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// attention_f16x2_kernel -- the same attention for long sequences (T a multiple of 256: 1024 / 4096 tokens at the 32x32 /
// 64x64 levels of the 256x256 configuration) on the f16 matrix cores with SPLIT fp32 operands (x = h + l, fp16 pieces,
// as in dmd_conv_f16ws.hip).  With head_dim 8 a (query, key) pair costs 16 MACs and one exponential: the kernel is bound by
// the vector unit's issue port (v_exp_f32 takes two of its slots: 8 cycles per wave64 instruction, additive with plain VALU work,
// tools/probe/trans_probe.hip / profiles/r04_trans_probe.txt), so everything else is taken off it:
//   * TWO passes over the keys instead of an online softmax.  Pass 1: S^T = K Q^T blocks and a running per-lane maximum
//     (one cross-lane reduction per query at its end).  Pass 2: the score MFMA starts from the accumulator -m_q + 13, so
//     its output is directly the exponent: p = 2^(s - m + 13) -- no subtraction, no per-block maximum, no rescaling of O,
//     and exactly softmax's x - max(x) (blocks.py:69).  The 2^13 moves the split's absolute floor (2^-25) to 2^-38 of the
//     largest weight and cancels in O / l.
//   * log2(e) / sqrt(d) is folded into Q once per query (p = exp2: one v_exp_f32 per pair, no expf range code).
//   * one v_mfma_f32_16x16x32_f16 per 16 keys x 16 queries does the whole split product q_h k_h + q_h k_l + q_l k_h:
//     the K = 32 slots are {k_h | k_l | k_h | 0} against {q_h | q_h | q_l | 0} (8 dims each).
//   * O^T[dim][query] += V^T P^T with A = [v_h ; v_l] stacked in the 16 rows and B = p_h, then p_l: two MFMAs per
//     32 keys give (p_h + p_l)(v_h + v_l); rows dim and 8 + dim are added once at the end.  P never moves between lanes:
//     the score MFMA leaves lane (j = lane & 15, kg = lane >> 4) with keys {4 kg + r} of a 16-key block for query j,
//     and the P^T operand's k-slots are simply DEFINED as those keys (A reads V^T with the same map).
// One workgroup = 4 waves x 64 queries (4 groups of 16) of one (image, head); K / V^T tiles of 256 keys are split once per
// workgroup while staging and double-buffered in LDS.
// ------------------------------------------------------------------------------------------------
typedef _Float16 att_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 att_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 att_h2 __attribute__((ext_vector_type(2)));
// lw = {fp16(p0 - h0), fp16(p1 - h1)} for hw = {h0, h1}: one mixed-precision fma per element (tests/simt, the host build of
// these sources, defines the macro with the same arithmetic in C++ before this point)
#ifndef ATT_SPLIT_LOW_PAIR
#define ATT_SPLIT_LOW_PAIR(lw, p0, p1, hw)                                          \
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"       \
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"           \
      : "=&v"(lw)                                                                   \
      : "v"(p0), "v"(p1), "v"(hw))
#endif

// max(a, b, c) as ONE v_max3_f32.  Pairwise fmaxf() trees made hipcc quiet every MFMA output first (v_max_f32 v, v, v: 72 of the
// 92 vector instructions of pass 1's inner loop, which made that pass VALU-bound instead of MFMA-bound); the three-input form
// needs no quieting.  (Not inline asm: the compiler has to see the MFMA -> VALU dependency to place the wait states.)
__device__ __forceinline__ float af_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// lab builds only (tools/build_attention_ablations.sh; WRONG results): timing proxies, bits: 1 no pass 1, 2 no exponentials,
// 4 no l-piece of P, 8 no row sums, 16 no PV MFMAs, 32 no score MFMAs in pass 2, 64 tiles staged once
#if defined(DMD_LAB) && defined(AF_ABL)
#define AF_LAB(bit) ((AF_ABL) & (bit))
#else
#define AF_LAB(bit) 0
#endif

#define AF_KT 256                   // keys per LDS tile
#define AF_VS (AF_KT + 8)           // V^T row stride in halfs (+16 bytes: rows start on different banks)
#define AF_QG 4                     // 16-query groups per wave
#define AF_SHIFT 13.0f              // exponent offset of the weights (see above)

struct AfTile {
  att_h8 kh[AF_KT];       // [key] dims 0..7, h pieces
  att_h8 kl[AF_KT];       // l pieces
  _Float16 vt[16][AF_VS];  // rows 0..7: v_h[dim][key], rows 8..15: v_l[dim][key]
};

__device__ __forceinline__ void af_split8(const f32x4& a, const f32x4& b, att_h8& h, att_h8& l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (_Float16)a[e];
    l[e] = (_Float16)(a[e] - (float)h[e]);
    h[4 + e] = (_Float16)b[e];
    l[4 + e] = (_Float16)(b[e] - (float)h[4 + e]);
  }
}

__global__ __launch_bounds__(256, 2) void attention_f16x2_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int C,
                                                              float qscale) {
  __shared__ AfTile tiles[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (64 * 4) + wave * 64;
  const size_t row = (size_t)3 * C;
  const float* base = qkv + (size_t)n * T * row;
  const int ntiles = T / AF_KT;

  // Q operand of S^T = K Q^T, per 16-query group: k-slots {q_h | q_h | q_l | 0}, scaled by log2(e) / sqrt(d)
  att_h8 bq[AF_QG];
#pragma unroll
  for (int g = 0; g < AF_QG; ++g) {
    const float* qp = base + (size_t)(q0 + g * 16 + j) * row + h * 8;
    f32x4 a = *(const f32x4*)qp, b = *(const f32x4*)(qp + 4);
    a *= qscale;
    b *= qscale;
    att_h8 qh, ql;
    af_split8(a, b, qh, ql);
    att_h8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
    bq[g] = kg < 2 ? qh : (kg == 2 ? ql : z);
  }

  // staging of one key tile: thread = key; K always, V^T in pass 2
  f32x4 sk0, sk1, sv0, sv1;
  auto stage_load = [&](int t, bool with_v) {
    const float* kp = base + (size_t)(t * AF_KT + tid) * row + C + h * 8;
    sk0 = *(const f32x4*)kp;
    sk1 = *(const f32x4*)(kp + 4);
    if (with_v) {
      sv0 = *(const f32x4*)(kp + C);
      sv1 = *(const f32x4*)(kp + C + 4);
    }
  };
  auto stage_store = [&](AfTile& tl, bool with_v) {
    att_h8 hh, ll;
    af_split8(sk0, sk1, hh, ll);
    tl.kh[tid] = hh;
    tl.kl[tid] = ll;
    if (with_v) {
      af_split8(sv0, sv1, hh, ll);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        tl.vt[e][tid] = hh[e];
        tl.vt[8 + e][tid] = ll[e];
      }
    }
  };
  // A operand of a score block: lane (key i = lane & 15, kg): slots {k_h | k_l | k_h | (k_h x 0)}
  auto k_frag = [&](const AfTile& tl, int k0) -> att_h8 { return kg == 1 ? tl.kl[k0 + j] : tl.kh[k0 + j]; };

  // ---------------- pass 1: row maxima ----------------
  float mx[AF_QG];
#pragma unroll
  for (int g = 0; g < AF_QG; ++g) mx[g] = -INFINITY;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  stage_load(0, false);
  stage_store(tiles[0], false);
  __syncthreads();
  for (int t = 0; t < (AF_LAB(1) ? 1 : ntiles); ++t) {
    const AfTile& tl = tiles[t & 1];
    if (t + 1 < ntiles) stage_load(t + 1, false);
#pragma unroll 2
    for (int k0 = 0; k0 < AF_KT; k0 += 32) {
      const att_h8 ka0 = k_frag(tl, k0), ka1 = k_frag(tl, k0 + 16);
#pragma unroll
      for (int g = 0; g < AF_QG; ++g) {
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka0, bq[g], zero4, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka1, bq[g], zero4, 0, 0, 0);
        mx[g] = af_max3(af_max3(mx[g], s0[0], s0[1]), af_max3(s0[2], s0[3], s1[0]), af_max3(s1[1], s1[2], s1[3]));
      }
    }
    if (t + 1 < ntiles) stage_store(tiles[(t + 1) & 1], false);
    __syncthreads();
  }
  f32x4 negm[AF_QG];
#pragma unroll
  for (int g = 0; g < AF_QG; ++g) {
    float m = mx[g];
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float c = AF_SHIFT - m;
    negm[g] = (f32x4){c, c, c, c};
  }

  // ---------------- pass 2: weights, row sums, O^T = V^T P^T ----------------
  f32x4 oacc[AF_QG];
  float lsum[AF_QG];
#pragma unroll
  for (int g = 0; g < AF_QG; ++g) {
    oacc[g] = zero4;
    lsum[g] = 0.f;
  }
  stage_load(0, true);
  stage_store(tiles[0], true);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const AfTile& tl = tiles[AF_LAB(64) ? 0 : (t & 1)];
    if (t + 1 < ntiles && !AF_LAB(64)) stage_load(t + 1, true);
#pragma unroll 2
    for (int k0 = 0; k0 < AF_KT; k0 += 32) {
      const att_h8 ka0 = k_frag(tl, k0), ka1 = k_frag(tl, k0 + 16);
      // V^T operand: lane (row i = lane & 15, kg): k-slots 0..3 = keys k0 + 4 kg + (0..3), 4..7 = keys k0 + 16 + 4 kg + (0..3)
      const att_h4 va = *(const att_h4*)&tl.vt[j][k0 + 4 * kg], vb = *(const att_h4*)&tl.vt[j][k0 + 16 + 4 * kg];
      const att_h8 vf = (att_h8){va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
      for (int g = 0; g < AF_QG; ++g) {
        const f32x4 s0 = AF_LAB(32) ? negm[g] + oacc[g] : __builtin_amdgcn_mfma_f32_16x16x32_f16(ka0, bq[g], negm[g], 0, 0, 0);
        const f32x4 s1 = AF_LAB(32) ? negm[g] - oacc[g] : __builtin_amdgcn_mfma_f32_16x16x32_f16(ka1, bq[g], negm[g], 0, 0, 0);
        float p[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = AF_LAB(2) ? s0[r] : __builtin_amdgcn_exp2f(s0[r]);
          p[4 + r] = AF_LAB(2) ? s1[r] : __builtin_amdgcn_exp2f(s1[r]);
        }
        if (!AF_LAB(8)) lsum[g] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        // split p = h + l: packed fp16 conversion for h, one mixed-precision fma per element for l = fp16(p - h)
        unsigned hw[4], lw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hw[r] = __builtin_bit_cast(unsigned, (att_h2){(_Float16)p[2 * r], (_Float16)p[2 * r + 1]});
          if (AF_LAB(4))
            lw[r] = hw[r];
          else
            ATT_SPLIT_LOW_PAIR(lw[r], p[2 * r], p[2 * r + 1], hw[r]);
        }
        typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
        const att_h8 ph = __builtin_bit_cast(att_h8, (att_u4){hw[0], hw[1], hw[2], hw[3]});
        const att_h8 pl = __builtin_bit_cast(att_h8, (att_u4){lw[0], lw[1], lw[2], lw[3]});
        if (AF_LAB(16)) {
          oacc[g] += __builtin_bit_cast(f32x4, ph) + __builtin_bit_cast(f32x4, pl);  // (keeps the operands alive: 8 plain adds)
        } else {
          oacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, ph, oacc[g], 0, 0, 0);
          oacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pl, oacc[g], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntiles && !AF_LAB(64)) stage_store(tiles[(t + 1) & 1], true);
    __syncthreads();
  }
  // rows dim (v_h) and 8 + dim (v_l) live in lanes kg and kg + 2; the row sum is spread over the 4 kg lanes of a query
#pragma unroll
  for (int g = 0; g < AF_QG; ++g) {
    float l = lsum[g];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    f32x4 o = oacc[g];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += __shfl_xor(o[r], 32, 64);
    if (kg < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = o[r] / l;
      *(f32x4*)(out + ((size_t)n * T + q0 + g * 16 + j) * C + h * 8 + 4 * kg) = o;
    }
  }
}

extern "C" int dmd_attention(const float* qkv, float* out, int N, int T, int C, int head_dim, dmd_stream_t stream) {
  DMD_CHECK_ARG(qkv && out, "attention: null");
  DMD_CHECK_ARG(head_dim == 8, "attention: head_dim must be 8 (ATTN_HEAD_DIM), got %d", head_dim);
  DMD_CHECK_ARG(C % 8 == 0 && T % 64 == 0 && N > 0, "attention: need C %% 8 == 0, T %% 64 == 0 (T=%d C=%d)", T, C);
  if (T % 256 == 0) {
    // long sequences (1024 / 4096 tokens of the 256x256 configuration): split-fp16 two-pass kernel
    hipLaunchKernelGGL(attention_f16x2_kernel, dim3(T / 256, C / 8, N), dim3(256), 0, (hipStream_t)stream, qkv, out, T, C,
                       1.4426950408889634f / sqrtf((float)head_dim));
    DMD_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(T / 64, C / 8, N);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, T, C, sqrtf((float)head_dim), 0, 0, 0);
  DMD_LAUNCH_CHECK();
  return 0;
}

// The same over an (H, W) token grid of which only (valid_h, valid_w) exists (dmd_conv_params: VALID EXTENT): keys outside
// it do not take part in the softmax; the outputs of queries outside it are unspecified.
extern "C" int dmd_attention_valid(const float* qkv, float* out, int N, int H, int W, int valid_h, int valid_w, int C, int head_dim,
                                   dmd_stream_t stream) {
  DMD_CHECK_ARG(qkv && out, "attention: null");
  DMD_CHECK_ARG(head_dim == 8, "attention: head_dim must be 8 (ATTN_HEAD_DIM), got %d", head_dim);
  const int T = H * W;
  DMD_CHECK_ARG(C % 8 == 0 && T % 64 == 0 && N > 0, "attention: need C %% 8 == 0, T %% 64 == 0 (T=%d C=%d)", T, C);
  DMD_CHECK_ARG(valid_h > 0 && valid_h <= H && valid_w > 0 && valid_w <= W, "attention: valid extent %d x %d of %d x %d", valid_h, valid_w, H, W);
  dim3 grid(T / 64, C / 8, N);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, T, C, sqrtf((float)head_dim), W, valid_h, valid_w);
  DMD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Attention backward (denoiser training step, denoiser.py:93-122 -> autograd of blocks.py:66-71).
//   P = softmax(q k^T / sqrt(d)),  y = P v.   Given dy:
//   D_i = dy_i . y_i ;  dP_ij = dy_i . v_j ;  dS_ij = P_ij (dP_ij - D_i)
//   dq_i = sum_j dS_ij k_j / sqrt(d) ;  dk_j = sum_i dS_ij q_i / sqrt(d) ;  dv_j = sum_i P_ij dy_i
// Two kernels, fp32 VALU (the default model attends over 64 tokens at the 8x8 level only: 3 MFLOP per image;
// this path is launch-bound, not FLOP-bound):
//   rows kernel: one thread per query row i -- softmax statistics (m_i, l_i) by a first sweep over the keys, then dq_i;
//                writes (m_i, l_i, D_i) for the second kernel;
//   cols kernel: one thread per key row j -- dk_j, dv_j by a sweep over the queries.
// d == 8.  qkv / dqkv are NHWC (N, T, 3C): q | k | v channel thirds, head h = channels [8h, 8h + 8).
// ------------------------------------------------------------------------------------------------
struct f8 {
  float v[8];
};
__device__ __forceinline__ f8 ld8(const float* p) {
  f8 r;
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r.v[e] = a[e];
    r.v[4 + e] = b[e];
  }
  return r;
}
__device__ __forceinline__ float dot8(const f8& a, const f8& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s = __builtin_fmaf(a.v[e], b.v[e], s);
  return s;
}

__global__ __launch_bounds__(64) void attention_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ y,
                                                                const float* __restrict__ dy, float* __restrict__ dqkv,
                                                                float* __restrict__ rowstat, int T, int C) {
  const int i = blockIdx.x * 64 + threadIdx.x, h = blockIdx.y, n = blockIdx.z, H = C / 8;
  if (i >= T) return;
  const float scale = 0.35355339059327373f;  // 1 / sqrt(8)
  const size_t row = ((size_t)n * T + i);
  const f8 q = ld8(qkv + row * 3 * C + h * 8);
  const f8 yo = ld8(y + row * C + h * 8), dyo = ld8(dy + row * C + h * 8);
  const float D = dot8(dyo, yo);
  const float* kbase = qkv + (size_t)n * T * 3 * C + C + h * 8;
  const float* vbase = kbase + C;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < T; ++j) {
    const float s = dot8(q, ld8(kbase + (size_t)j * 3 * C)) * scale;
    const float mn = fmaxf(m, s);
    l = l * expf(m - mn) + expf(s - mn);
    m = mn;
  }
  f8 dq;
#pragma unroll
  for (int e = 0; e < 8; ++e) dq.v[e] = 0.f;
  for (int j = 0; j < T; ++j) {
    const f8 k = ld8(kbase + (size_t)j * 3 * C);
    const float s = dot8(q, k) * scale;
    const float p = expf(s - m) / l;
    const float ds = p * (dot8(dyo, ld8(vbase + (size_t)j * 3 * C)) - D);
#pragma unroll
    for (int e = 0; e < 8; ++e) dq.v[e] = __builtin_fmaf(ds * scale, k.v[e], dq.v[e]);
  }
  float* o = dqkv + row * 3 * C + h * 8;
  *(f32x4*)o = (f32x4){dq.v[0], dq.v[1], dq.v[2], dq.v[3]};
  *(f32x4*)(o + 4) = (f32x4){dq.v[4], dq.v[5], dq.v[6], dq.v[7]};
  float* rs = rowstat + (((size_t)n * H + h) * T + i) * 4;
  rs[0] = m;
  rs[1] = l;
  rs[2] = D;
}

__global__ __launch_bounds__(64) void attention_bwd_cols_kernel(const float* __restrict__ qkv, const float* __restrict__ dy,
                                                                const float* __restrict__ rowstat, float* __restrict__ dqkv, int T,
                                                                int C) {
  const int j = blockIdx.x * 64 + threadIdx.x, h = blockIdx.y, n = blockIdx.z, H = C / 8;
  if (j >= T) return;
  const float scale = 0.35355339059327373f;
  const size_t row = ((size_t)n * T + j);
  const f8 k = ld8(qkv + row * 3 * C + C + h * 8), v = ld8(qkv + row * 3 * C + 2 * C + h * 8);
  const float* qbase = qkv + (size_t)n * T * 3 * C + h * 8;
  const float* dybase = dy + (size_t)n * T * C + h * 8;
  const float* rs = rowstat + ((size_t)n * H + h) * T * 4;
  f8 dk, dv;
#pragma unroll
  for (int e = 0; e < 8; ++e) dk.v[e] = dv.v[e] = 0.f;
  for (int i = 0; i < T; ++i) {
    const f8 q = ld8(qbase + (size_t)i * 3 * C);
    const f8 dyo = ld8(dybase + (size_t)i * C);
    const float s = dot8(q, k) * scale;
    const float p = expf(s - rs[4 * i]) / rs[4 * i + 1];
    const float ds = p * (dot8(dyo, v) - rs[4 * i + 2]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dk.v[e] = __builtin_fmaf(ds * scale, q.v[e], dk.v[e]);
      dv.v[e] = __builtin_fmaf(p, dyo.v[e], dv.v[e]);
    }
  }
  float* o = dqkv + row * 3 * C + C + h * 8;
  *(f32x4*)o = (f32x4){dk.v[0], dk.v[1], dk.v[2], dk.v[3]};
  *(f32x4*)(o + 4) = (f32x4){dk.v[4], dk.v[5], dk.v[6], dk.v[7]};
  o += C;
  *(f32x4*)o = (f32x4){dv.v[0], dv.v[1], dv.v[2], dv.v[3]};
  *(f32x4*)(o + 4) = (f32x4){dv.v[4], dv.v[5], dv.v[6], dv.v[7]};
}

extern "C" int64_t dmd_attention_bwd_workspace_floats(int N, int T, int C) { return (int64_t)N * (C / 8) * T * 4; }

extern "C" int dmd_attention_bwd(const float* qkv, const float* y, const float* dy, float* dqkv, float* workspace, int N, int T,
                                 int C, int head_dim, dmd_stream_t stream) {
  DMD_CHECK_ARG(qkv && y && dy && dqkv && workspace, "attention_bwd: null");
  DMD_CHECK_ARG(head_dim == 8 && C % 8 == 0 && N > 0 && T > 0, "attention_bwd: head_dim must be 8 (got %d), C %% 8 == 0", head_dim);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 63) / 64, C / 8, N);
  hipLaunchKernelGGL(attention_bwd_rows_kernel, grid, dim3(64), 0, st, qkv, y, dy, dqkv, workspace, T, C);
  hipLaunchKernelGGL(attention_bwd_cols_kernel, grid, dim3(64), 0, st, qkv, dy, (const float*)workspace, dqkv, T, C);
  DMD_LAUNCH_CHECK();
  return 0;
}
