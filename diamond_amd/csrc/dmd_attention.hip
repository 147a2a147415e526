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
#include "dmd_common.h"

#define ATT_KB 256          // keys per LDS tile
#define ATT_VSTRIDE (ATT_KB + 16)

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                        int C, float inv_scale_den) {
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

extern "C" int dmd_attention(const float* qkv, float* out, int N, int T, int C, int head_dim, dmd_stream_t stream) {
  DMD_CHECK_ARG(qkv && out, "attention: null");
  DMD_CHECK_ARG(head_dim == 8, "attention: head_dim must be 8 (ATTN_HEAD_DIM), got %d", head_dim);
  DMD_CHECK_ARG(C % 8 == 0 && T % 64 == 0 && N > 0, "attention: need C %% 8 == 0, T %% 64 == 0 (T=%d C=%d)", T, C);
  dim3 grid(T / 64, C / 8, N);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, T, C, sqrtf((float)head_dim));
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
