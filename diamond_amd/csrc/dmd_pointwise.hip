// Pointwise / small kernels of the imagined-rollout path: EDM preconditioning and sampler
// updates (denoiser.py:66-84, diffusion_sampler.py:45-56), the cond embedding
// (blocks.py:84-87, inner_model.py:27-30), layout shuffles at the NCHW API boundary,
// GroupNorm partial statistics, 2x2 max-pool (actor_critic.py:108-109), the LSTM cell
// nonlinearity and Categorical sampling with injected exponential draws.
// All HBM-bound: coalesced, 16-byte accesses where the layout allows, one pass.
#include "dmd_common.h"

// ---- cat(obs / sigma_data, x * c_in) : NCHW -> NHWC(CPad) ------------------------------------
// One thread per (pixel, channel quad): the NCHW reads of a wave are 16 consecutive pixels of 4 x 4 planes (64-byte
// segments), the NHWC writes of a wave are 1 KiB contiguous (16 bytes per lane, consecutive lanes).
// The conditioning frames may live in a RING of T frames of Cobs / T channels each: logical frame t is stored at
// physical slot (head + t) % T (WorldModelEnv keeps its context that way instead of rolling it every step,
// world_model_env.py:74-75); T = 1, head = 0 is a plain (N, Cobs, H, W) tensor.
__global__ void edm_pack_input_kernel(const float* __restrict__ x, const float* __restrict__ obs,
                                      const float* __restrict__ cond, int cond_stride, float sd,
                                      float* __restrict__ out, int N, int Cx, int Cobs, int HW, int CPad, int T, int head) {
  // thread = (4 consecutive pixels, 4 consecutive output channels): four 16-byte plane reads (a wave reads 256 contiguous
  // bytes of each of its 4 planes), a 4x4 transpose in registers, four 16-byte NHWC stores
  const int Q = CPad >> 2, HW4 = HW >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * HW4 * Q) return;
  const int qd = idx % Q;
  const size_t npq = idx / Q;
  const int n = npq / HW4;
  const int pix0 = 4 * (int)(npq - (size_t)n * HW4);
  const float c_in = cond[(size_t)n * cond_stride + 0];
  const int cimg = Cobs / T;
  f32x4 t[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cc = qd * 4 + e;
    t[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (cc < Cobs) {
      const int fr = cc / cimg, ci = cc - fr * cimg;
      int slot = fr + head;
      slot = slot >= T ? slot - T : slot;
      const f32x4 v = *(const f32x4*)(obs + ((size_t)n * Cobs + slot * cimg + ci) * HW + pix0);
#pragma unroll
      for (int j = 0; j < 4; ++j) t[e][j] = v[j] / sd;  // rescaled_obs = obs / sigma_data, denoiser.py:75
    } else if (cc < Cobs + Cx) {
      const f32x4 v = *(const f32x4*)(x + ((size_t)n * Cx + (cc - Cobs)) * HW + pix0);
#pragma unroll
      for (int j = 0; j < 4; ++j) t[e][j] = v[j] * c_in;  // denoiser.py:76
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *(f32x4*)(out + ((size_t)n * HW + pix0 + j) * CPad + qd * 4) = (f32x4){t[0][j], t[1][j], t[2][j], t[3][j]};
}

// ---- cond input: [cos(f) | sin(f)] + flatten(Embedding(act)),  f = 2 pi c_noise w -----------
__global__ void cond_embed_kernel(const float* __restrict__ cond, int cond_stride,
                                  const float* __restrict__ fw, const int64_t* __restrict__ act,
                                  const float* __restrict__ emb, float* __restrict__ out, int N, int half, int T, int E,
                                  int act_head, int A) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int D = 2 * half;
  if (idx >= N * D) return;
  const int n = idx / D, jj = idx - n * D;
  const float c_noise = cond[(size_t)n * cond_stride + 3];
  // blocks.py:86: f = 2 * math.pi * input.unsqueeze(1) @ weight   (left to right: (2 pi c) * w)
  const float two_pi_c = (float)(2.0 * 3.141592653589793) * c_noise;
  const int k = jj < half ? jj : jj - half;
  const float f = two_pi_c * fw[k];
  float v = jj < half ? cosf(f) : sinf(f);
  const int t = jj / E, e = jj - t * E;  // flatten (T, E) -> T*E == D
  int slot = t + act_head;  // logical step t of a ring of T actions starting at act_head
  slot = slot >= T ? slot - T : slot;
  int64_t a = act[(size_t)n * T + slot];
  a = a < 0 ? 0 : (a >= A ? A - 1 : a);  // memory safety only: nn.Embedding raises on an out-of-range index, the host checks it in debug mode
  v += emb[(size_t)a * E + e];
  out[idx] = v;
}

// ---- denoised = quantise(c_skip * x + c_out * F) (denoiser.py:81-83) -------------------------
__device__ __forceinline__ float dmd_quantise(float d) {
  d = fminf(fmaxf(d, -1.0f), 1.0f);       // clamp(-1, 1)
  d = d + 1.0f;                           // add(1)
  d = d / 2.0f;                           // div(2)
  d = d * 255.0f;                         // mul(255)
  const float q = (float)(uint8_t)d;      // byte(): truncation toward zero, value in [0, 255]
  return (q / 255.0f) * 2.0f - 1.0f;      // div(255).mul(2).sub(1)
}

__global__ void edm_denoised_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                    const float* __restrict__ cond, int cond_stride,
                                    float* __restrict__ den, int N, int64_t per_sample) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * per_sample) return;
  const int n = idx / per_sample;
  const float c_out = cond[(size_t)n * cond_stride + 1], c_skip = cond[(size_t)n * cond_stride + 2];
  const float a = c_skip * x[idx];
  const float b = c_out * f[idx];
  den[idx] = dmd_quantise(a + b);
}

__global__ void euler_step_kernel(const float* __restrict__ x, const float* __restrict__ den, float sigma_hat, float dt,
                                  float* __restrict__ xo, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float xv = x[idx];
  const float d = (xv - den[idx]) / sigma_hat;  // diffusion_sampler.py:45
  xo[idx] = xv + d * dt;                        // :49
}

// Heun (2nd order) update, diffusion_sampler.py:52-56, same fp32 op order:
//   d = (x - D) / sigma_hat;  d_2 = (x_2 - D_2) / sigma_next;  x_out = x + ((d + d_2) / 2) * dt
__global__ void heun_step_kernel(const float* __restrict__ x, const float* __restrict__ den, const float* __restrict__ x2,
                                 const float* __restrict__ den2, float sigma_hat, float sigma_next, float dt,
                                 float* __restrict__ xo, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float xv = x[idx];
  const float d = (xv - den[idx]) / sigma_hat;
  const float d2 = (x2[idx] - den2[idx]) / sigma_next;
  const float dp = (d + d2) / 2.0f;
  xo[idx] = xv + dp * dt;
}

// ---- NCHW <-> NHWC(CPad) ---------------------------------------------------------------------
// one thread per (pixel, channel quad): 1 KiB contiguous NHWC writes per wave
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int HW, int CPad) {
  const int Q = CPad >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * HW * Q) return;
  const int qd = idx % Q;
  const size_t np = idx / Q;
  const int n = np / HW;
  const int pix = np - (size_t)n * HW;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (qd * 4 + e) < C ? in[((size_t)n * C + qd * 4 + e) * HW + pix] : 0.f;
  *(f32x4*)(out + np * CPad + qd * 4) = v;
}

// ---- uint8 frame pool (episode frames are uint8 on disk, data/episode.py:36-50) ----------------
// u8 = round((x + 1) / 2 * 255) of frames in [-1, 1]; *off_grid is set when a value is not exactly the
// dequantisation of its level (such a pool must stay fp32).
__global__ void quantize_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ q, int* __restrict__ off_grid, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float v = x[idx];
  const float lv = rintf((v + 1.0f) / 2.0f * 255.0f);
  const float lc = fminf(fmaxf(lv, 0.0f), 255.0f);
  q[idx] = (uint8_t)lc;
  const float back = (lc / 255.0f) * 2.0f - 1.0f;  // Episode.load: .div(255).mul(2).sub(1)
  if (!(back == v)) *off_grid = 1;                 // benign race: every writer stores 1
}

// dst[row[i]] frames (logical order, ring slot (head + t) % T) <- dequantised pool frames src[idx[i]]
// per_frame = C * H * W; one thread per 4 consecutive values
__global__ void dequant_gather_kernel(const uint8_t* __restrict__ pool, const int64_t* __restrict__ idx,
                                      const int64_t* __restrict__ rows, float* __restrict__ dst, int M, int T,
                                      int64_t per_frame, int head) {
  const int64_t q4 = per_frame >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)M * T * q4) return;
  const int64_t e4 = gid % q4;
  const int64_t mt = gid / q4;
  const int t = mt % T;
  const int m = mt / T;
  const int64_t src = idx ? idx[m] : m;
  const int64_t row = rows ? rows[m] : m;
  int slot = t + head;
  slot = slot >= T ? slot - T : slot;
  const uint32_t pk = *(const uint32_t*)(pool + (src * T + t) * per_frame + e4 * 4);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = ((float)((pk >> (8 * e)) & 0xff) / 255.0f) * 2.0f - 1.0f;
  *(f32x4*)(dst + (row * T + slot) * per_frame + e4 * 4) = v;
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int HW, int CPad) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * HW) return;
  const int n = idx / HW;
  const int pix = idx - (size_t)n * HW;
  const float* i = in + idx * CPad;
  for (int ch = 0; ch < C; ++ch) out[((size_t)n * C + ch) * HW + pix] = i[ch];
}

// ---- GroupNorm partial statistics of an NHWC tensor: one tile per image ----------------------
// grid (G, N), 256 threads: thread -> (pixel stripe, channel quad of the group)
// W > 0: the HW pixels are an (HW / W) x W grid of which only rows < hv, columns < wv are counted (VALID EXTENT)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int HW, int C, int W, int hv,
                                                       int wv) {
  __shared__ double red[4][2];
  const int g = blockIdx.x, n = blockIdx.y, G = gridDim.x;
  const int tid = threadIdx.x;
  const int quad = tid & 7;  // 8 quads = 32 channels
  double s = 0.0, ss = 0.0;
  for (int pix = tid >> 3; pix < HW; pix += 32) {
    if (W > 0) {
      const int py = pix / W;
      if (py >= hv || pix - py * W >= wv) continue;
    }
    const f32x4 v = *(const f32x4*)(x + ((size_t)n * HW + pix) * C + g * DMD_GN_GROUP + quad * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d = (double)v[e];
      s += d;
      ss += d * d;
    }
  }
  s = dmd_wave_sum(s);
  ss = dmd_wave_sum(ss);
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = s;
    red[tid >> 6][1] = ss;
  }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) {
      a += red[w][0];
      b += red[w][1];
    }
    stats[((size_t)n * G + g) * 2] = a;
    stats[((size_t)n * G + g) * 2 + 1] = b;
  }
}

// ---- 2x2 max pool, NHWC; one thread per (output pixel, channel quad) ---------------------------
// argmax: index 0..3 (dy*2+dx) of the FIRST maximum in scan order (ATen max_pool2d picks the
// first max it meets scanning h then w; ties only matter for the backward routing).
__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ out, uint8_t* __restrict__ argmax, int N,
                                int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, Cq = C / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * Ho * Wo * Cq) return;
  const int cq = idx % Cq;
  const size_t op = idx / Cq;
  const int ox = op % Wo;
  const int oy = (op / Wo) % Ho;
  const int n = op / ((size_t)Wo * Ho);
  f32x4 best;
  int bi[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
    const f32x4 v = *(const f32x4*)(x + (((size_t)n * H + iy) * W + ix) * C + cq * 4);
    if (k == 0)
      best = v;
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > best[e] || v[e] != v[e]) {  // NaN propagates like ATen
          best[e] = v[e];
          bi[e] = k;
        }
    }
  }
  *(f32x4*)(out + op * C + cq * 4) = best;
  if (argmax) {
    uint32_t packed = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    *(uint32_t*)(argmax + op * C + cq * 4) = packed;
  }
}

// ---- LSTM cell nonlinearity (gate order i, f, g, o) --------------------------------------------
__global__ void lstm_pointwise_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                      float* __restrict__ h, float* __restrict__ c, int N, int Hd) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * Hd) return;
  const int n = idx / Hd, k = idx - n * Hd;
  const float* gr = gates + (size_t)n * 4 * Hd;
  const float ig = dmd_sigmoid(gr[k]);
  const float fg = dmd_sigmoid(gr[Hd + k]);
  const float gg = tanhf(gr[2 * Hd + k]);
  const float og = dmd_sigmoid(gr[3 * Hd + k]);
  const float cn = fg * c_prev[idx] + ig * gg;
  c[idx] = cn;
  h[idx] = og * tanhf(cn);
}

// LSTM cell backward (autograd of nn.LSTMCell under the actor-critic loss, actor_critic.py:46,72 / trainer.py:366):
// gates = pre-activations (i, f, g, o) saved by the forward, dh / dc = gradients w.r.t. the new h / c (either may be NULL)
// -> dgates (N, 4 Hd), dc_prev (N, Hd)
__global__ void lstm_pointwise_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                          const float* __restrict__ c_new, const float* __restrict__ dh,
                                          const float* __restrict__ dc, float* __restrict__ dgates, float* __restrict__ dc_prev,
                                          int N, int Hd) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * Hd) return;
  const int n = idx / Hd, k = idx - n * Hd;
  const float* gr = gates + (size_t)n * 4 * Hd;
  const float ig = dmd_sigmoid(gr[k]);
  const float fg = dmd_sigmoid(gr[Hd + k]);
  const float gg = tanhf(gr[2 * Hd + k]);
  const float og = dmd_sigmoid(gr[3 * Hd + k]);
  const float tc = tanhf(c_new[idx]);
  const float dhv = dh ? dh[idx] : 0.f;
  const float dcv = (dc ? dc[idx] : 0.f) + dhv * og * (1.0f - tc * tc);
  float* dg = dgates + (size_t)n * 4 * Hd;
  dg[k] = dcv * gg * ig * (1.0f - ig);
  dg[Hd + k] = dcv * c_prev[idx] * fg * (1.0f - fg);
  dg[2 * Hd + k] = dcv * ig * (1.0f - gg * gg);
  dg[3 * Hd + k] = dhv * tc * og * (1.0f - og);
  dc_prev[idx] = dcv * fg;
}

// ---- Categorical(logits).sample() with injected E ~ Exp(1): argmax(softmax(logits) / E) -------
__global__ void categorical_sample_kernel(const float* __restrict__ logits, const float* __restrict__ expo,
                                          int64_t* __restrict__ out, int N, int A) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* l = logits + (size_t)n * A;
  const float* e = expo + (size_t)n * A;
  // torch: Categorical(logits=l) keeps ln = l - logsumexp(l); probs = softmax(ln)
  float m = -INFINITY;
  for (int a = 0; a < A; ++a) m = fmaxf(m, l[a]);
  float sum = 0.f;
  for (int a = 0; a < A; ++a) sum += expf(l[a] - m);
  const float lse = m + logf(sum);
  float m2 = -INFINITY;
  for (int a = 0; a < A; ++a) m2 = fmaxf(m2, l[a] - lse);
  float sum2 = 0.f;
  for (int a = 0; a < A; ++a) sum2 += expf((l[a] - lse) - m2);
  float best = -INFINITY;
  int bi = 0;
  for (int a = 0; a < A; ++a) {
    const float p = expf((l[a] - lse) - m2) / sum2;
    const float v = p / e[a];
    if (v > best) {
      best = v;
      bi = a;
    }
  }
  out[n] = bi;
}

// ------------------------------------------------------------------------------------------------
static inline unsigned nblk(size_t n, int b) { return (unsigned)((n + b - 1) / b); }

extern "C" int dmd_edm_pack_input(const float* x, const float* obs, const float* cond, int cond_stride, float sigma_data,
                                  float* out, int N, int Cx, int Cobs, int H, int W, int CPad, int T, int head,
                                  dmd_stream_t stream) {
  DMD_CHECK_ARG(x && obs && cond && out, "edm_pack_input: null");
  DMD_CHECK_ARG(CPad % 4 == 0 && CPad >= Cx + Cobs, "edm_pack_input: CPad");
  DMD_CHECK_ARG((H * W) % 4 == 0, "edm_pack_input: H * W must be a multiple of 4 (16-byte plane reads)");
  DMD_CHECK_ARG(T >= 1 && Cobs % T == 0 && head >= 0 && head < T, "edm_pack_input: ring T %d head %d Cobs %d", T, head, Cobs);
  hipLaunchKernelGGL(edm_pack_input_kernel, dim3(nblk((size_t)N * (H * W / 4) * (CPad / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     x, obs, cond, cond_stride, sigma_data, out, N, Cx, Cobs, H * W, CPad, T, head);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_cond_embed(const float* cond, int cond_stride, const float* fw, const int64_t* act,
                              const float* emb, float* out, int N, int half, int T, int E, int act_head, int A,
                              dmd_stream_t stream) {
  DMD_CHECK_ARG(cond && fw && act && emb && out, "cond_embed: null");
  DMD_CHECK_ARG(T * E == 2 * half, "cond_embed: T*E (%d) != cond channels (%d)", T * E, 2 * half);
  DMD_CHECK_ARG(act_head >= 0 && act_head < T && A > 0, "cond_embed: act_head %d outside [0, %d) or A %d", act_head, T, A);
  hipLaunchKernelGGL(cond_embed_kernel, dim3(nblk((size_t)N * 2 * half, 256)), dim3(256), 0, (hipStream_t)stream, cond,
                     cond_stride, fw, act, emb, out, N, half, T, E, act_head, A);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_edm_denoised(const float* x, const float* f, const float* cond, int cond_stride,
                                float* den, int N, int64_t per_sample, dmd_stream_t stream) {
  DMD_CHECK_ARG(x && f && cond && den, "edm_denoised: null");
  hipLaunchKernelGGL(edm_denoised_kernel, dim3(nblk((size_t)N * per_sample, 256)), dim3(256), 0, (hipStream_t)stream, x, f,
                     cond, cond_stride, den, N, per_sample);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_euler_step(const float* x, const float* den, float sigma_hat, float dt, float* xo, int64_t n,
                              dmd_stream_t stream) {
  DMD_CHECK_ARG(x && den && xo, "euler_step: null");
  hipLaunchKernelGGL(euler_step_kernel, dim3(nblk((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, x, den, sigma_hat, dt,
                     xo, n);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_heun_step(const float* x, const float* den, const float* x2, const float* den2, float sigma_hat,
                             float sigma_next, float dt, float* xo, int64_t n, dmd_stream_t stream) {
  DMD_CHECK_ARG(x && den && x2 && den2 && xo, "heun_step: null");
  hipLaunchKernelGGL(heun_step_kernel, dim3(nblk((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, x, den, x2, den2,
                     sigma_hat, sigma_next, dt, xo, n);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_quantize_u8(const float* x, uint8_t* q, int* off_grid, int64_t n, dmd_stream_t stream) {
  DMD_CHECK_ARG(x && q && off_grid, "quantize_u8: null");
  hipLaunchKernelGGL(quantize_u8_kernel, dim3(nblk((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, x, q, off_grid, n);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_dequant_gather(const uint8_t* pool, const int64_t* idx, const int64_t* rows, float* dst, int M, int T,
                                  int64_t per_frame, int head, dmd_stream_t stream) {
  DMD_CHECK_ARG(pool && dst, "dequant_gather: null");
  DMD_CHECK_ARG(per_frame % 4 == 0 && T >= 1 && head >= 0 && head < T, "dequant_gather: per_frame %% 4, ring");
  if (M == 0) return 0;
  hipLaunchKernelGGL(dequant_gather_kernel, dim3(nblk((size_t)M * T * (per_frame / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     pool, idx, rows, dst, M, T, per_frame, head);
  DMD_LAUNCH_CHECK();
  return 0;
}

// The small state of a reset in ONE launch (reference world_model_env.py:56-62 `reset_dead`: act_buffer[dead] = act,
// hx_rew_end[:, dead] = hx, cx likewise, ep_len[dead] = 0 -- four indexed assignments, ~12 launches of torch glue on the step's
// critical path, right behind its host synchronisation): block i handles env row rows[i] <- pool row idx[i].
__global__ __launch_bounds__(256) void reset_state_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ rows,
                                                          const int64_t* __restrict__ pool_act, int64_t* __restrict__ act_ring, int T, int head,
                                                          const float* __restrict__ pool_hx, const float* __restrict__ pool_cx,
                                                          float* __restrict__ hx, float* __restrict__ cx, int hd, int64_t* __restrict__ ep_len) {
  const int64_t q = idx[blockIdx.x], r = rows[blockIdx.x];
  for (int j = threadIdx.x; j < hd; j += 256) {
    hx[r * hd + j] = pool_hx[q * hd + j];
    cx[r * hd + j] = pool_cx[q * hd + j];
  }
  if ((int)threadIdx.x < T) act_ring[r * T + (head + threadIdx.x) % T] = pool_act[q * T + threadIdx.x];
  if (threadIdx.x == 0) ep_len[r] = 0;
}

extern "C" int dmd_reset_state(const int64_t* idx, const int64_t* rows, int count, const int64_t* pool_act, int64_t* act_ring, int T,
                               int head, const float* pool_hx, const float* pool_cx, float* hx, float* cx, int hd, int64_t* ep_len,
                               dmd_stream_t stream) {
  DMD_CHECK_ARG(idx && rows && pool_act && act_ring && pool_hx && pool_cx && hx && cx && ep_len, "reset_state: null");
  DMD_CHECK_ARG(T >= 1 && T <= 256 && head >= 0 && head < T && hd >= 1, "reset_state: T %d, head %d, hd %d", T, head, hd);
  if (count == 0) return 0;
  hipLaunchKernelGGL(reset_state_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, idx, rows, pool_act, act_ring, T, head,
                     pool_hx, pool_cx, hx, cx, hd, ep_len);
  DMD_LAUNCH_CHECK();
  return 0;
}

// ---- the deaths of an imagined step resolved ON THE DEVICE (ABI v11) -----------------------------------------------------
// The reference asks the host once per step which episodes ended (`if dead.any()`, world_model_env.py:77; env_loop.py:45) and
// shapes what follows by the answer.  Here the answer stays on the device: the step works on K SLOTS (K chosen by the host
// before it knows the step's deaths), the dead rows are assigned to slots in ascending row order -- the order in which the
// reference's boolean masks enumerate them and its pool serves them (world_model_env.py:56-57,133-139) -- and everything
// downstream is fixed-shape over the slots; an unused slot is marked -1.  The host reads `report` one step LATE (episode-length
// mirror, pool cursor, slot overflow), behind a full step of queued work.
//   per row r:  ep_len[r] += 1; trunc[r] = ep_len[r] >= horizon; dead[r] = end[r] | trunc[r]; ep_len[r] = 0 where dead
//   slot_row[j] = j-th dead row (-1: unused slot);  row_slot[r] = slot of a dead row (-1: alive, or no slot left)
//   report = {dead[0..B) as int32, k = number of dead rows, number of rows with end != 0, k > K, K}
__global__ __launch_bounds__(256) void resolve_deaths_kernel(const int64_t* __restrict__ end, int64_t* __restrict__ ep_len, int horizon,
                                                             int64_t* __restrict__ trunc, uint8_t* __restrict__ dead, int B, int K,
                                                             int64_t* __restrict__ slot_row, int32_t* __restrict__ row_slot,
                                                             int32_t* __restrict__ report) {
  __shared__ int flag[256];
  __shared__ int eflag[256];
  const int tid = threadIdx.x;
  int base = 0, ends = 0;
  for (int r0 = 0; r0 < B; r0 += 256) {
    const int r = r0 + tid;
    bool d = false, e = false;
    if (r < B) {
      const int64_t l = ep_len[r] + 1;
      const bool t = l >= (int64_t)horizon;
      e = end[r] != 0;
      d = e || t;
      trunc[r] = t ? 1 : 0;
      dead[r] = d ? 1 : 0;
      ep_len[r] = d ? 0 : l;
      report[r] = d ? 1 : 0;
    }
    flag[tid] = d ? 1 : 0;
    eflag[tid] = e ? 1 : 0;
    __syncthreads();
    int before = 0, total = 0, etotal = 0;  // (256 LDS reads per thread: a one-workgroup kernel of a few microseconds)
    for (int i = 0; i < 256; ++i) {
      before += i < tid ? flag[i] : 0;
      total += flag[i];
      etotal += eflag[i];
    }
    if (r < B) {
      const int slot = base + before;
      const bool has = d && slot < K;
      row_slot[r] = has ? slot : -1;
      if (has) slot_row[slot] = r;
    }
    base += total;
    ends += etotal;
    __syncthreads();
  }
  for (int j = base + tid; j < K; j += 256) slot_row[j] = -1;
  if (tid == 0) {
    report[B] = base;
    report[B + 1] = ends;
    report[B + 2] = base > K ? 1 : 0;
    report[B + 3] = K;
  }
}

extern "C" int dmd_resolve_deaths(const int64_t* end, int64_t* ep_len, int horizon, int64_t* trunc, uint8_t* dead, int B, int K,
                                  int64_t* slot_row, int32_t* row_slot, int32_t* report, dmd_stream_t stream) {
  DMD_CHECK_ARG(end && ep_len && trunc && dead && row_slot && report && (slot_row || K == 0), "resolve_deaths: null");
  DMD_CHECK_ARG(B >= 1 && K >= 0 && horizon >= 1, "resolve_deaths: B %d, K %d, horizon %d", B, K, horizon);
  hipLaunchKernelGGL(resolve_deaths_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, end, ep_len, horizon, trunc, dead, B, K, slot_row,
                     row_slot, report);
  DMD_LAUNCH_CHECK();
  return 0;
}

// Ring advance + reset of the dead rows + the policy's next input, one launch over B + T * K frames (grid.y):
//   frame f <  B          env row f: its newest frame -- the imagined one, or (dead row with a slot) the newest frame of its NEW
//                         episode -- into enc_in[f] AND into the ring slot the advance frees (logical T - 1 at the new head)
//   frame f <  B + K      slot j = f - B: the FINAL observation of the ended episode (the imagined frame of row slot_row[j])
//   frame f >= B + K      burn-in frame t of slot j (frame-major: f = B + K + t * K + j, t < T - 1): the new episode's context
//                         frame t, into enc_in[f] AND into ring slot (head + t) % T of the row
// A slot's new episode is row base + j of a pool round (dequantised: (u8 / 255) * 2 - 1, exact zeros where `pad` marks a padded
// frame; or an fp32 round).  Which round: pool[0] from pool_base on, or -- when the step's deaths do not fit what is left of it,
// the reference's rule (world_model_env.py:133-139) -- pool[1] from row 0 (ws_pool_select: the count is on the device only).
// Unused slots (slot_row -1) receive copies of row 0's imagined frame: finite values nobody reads.
__device__ __forceinline__ int ws_pool_select(const dmd_reset_slots_params& p, int64_t* base) {
  const bool second = p.pool[1].frames && p.num_dead && (p.pool_base + (int64_t)*p.num_dead > (int64_t)p.pool[0].rows);
  *base = second ? 0 : p.pool_base;
  return second ? 1 : 0;
}

__global__ __launch_bounds__(256) void reset_slot_frames_kernel(dmd_reset_slots_params p) {
  const int64_t q4 = p.per_frame >> 2;
  const int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e4 >= q4) return;
  const int f = blockIdx.y;
  const int B = p.B, K = p.K, T = p.T;
  int64_t base = 0;
  const dmd_pool_round& pool = p.pool[K > 0 ? ws_pool_select(p, &base) : 0];
  int64_t pool_row = -1, obs_row = 0, ring_row = -1;
  int pool_t = 0, ring_slot = 0;
  if (f < B) {
    const int s = p.row_slot[f];
    ring_row = f;
    ring_slot = (p.head + T - 1) % T;
    if (s >= 0) {
      pool_row = base + s;
      pool_t = T - 1;
    } else {
      obs_row = f;
    }
  } else if (f < B + K) {
    const int64_t r = p.slot_row[f - B];
    obs_row = r >= 0 ? r : 0;
  } else {
    const int g = f - B - K;
    const int t = g / K, j = g - t * K;
    const int64_t r = p.slot_row[j];
    if (r >= 0) {
      pool_row = base + j;
      pool_t = t;
      ring_row = r;
      ring_slot = (p.head + t) % T;
    }
  }
  f32x4 v;
  if (pool_row >= 0) {
    const int64_t pf = pool_row * T + pool_t;
    if (!pool.is_f32) {
      const uint32_t pk = *(const uint32_t*)((const uint8_t*)pool.frames + pf * p.per_frame + e4 * 4);
      const bool pad = pool.pad && pool.pad[pf];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pad ? 0.0f : ((float)((pk >> (8 * e)) & 0xff) / 255.0f) * 2.0f - 1.0f;
    } else {
      v = *(const f32x4*)((const float*)pool.frames + pf * p.per_frame + e4 * 4);
    }
  } else {
    v = *(const f32x4*)(p.next_obs + obs_row * p.per_frame + e4 * 4);
  }
  *(f32x4*)(p.enc_in + (int64_t)f * p.per_frame + e4 * 4) = v;
  if (ring_row >= 0) *(f32x4*)(p.ctx + (ring_row * T + ring_slot) * p.per_frame + e4 * 4) = v;
}

// the small state of the slots' resets: action ring, reward/end LSTM state (reset_state_kernel with the slot indirection;
// ep_len was zeroed by resolve_deaths_kernel)
__global__ __launch_bounds__(256) void reset_slot_state_kernel(dmd_reset_slots_params p) {
  const int j = blockIdx.x;
  const int64_t r = p.slot_row[j];
  if (r < 0) return;
  int64_t base;
  const dmd_pool_round& pool = p.pool[ws_pool_select(p, &base)];
  const int64_t q = base + j;
  const int hd = p.hd, T = p.T;
  for (int i = threadIdx.x; i < hd; i += 256) {
    p.hx[r * hd + i] = pool.hx[q * hd + i];
    p.cx[r * hd + i] = pool.cx[q * hd + i];
  }
  if ((int)threadIdx.x < T) p.act_ring[r * T + (p.head + threadIdx.x) % T] = pool.act[q * T + threadIdx.x];
}

extern "C" int dmd_reset_slots(const dmd_reset_slots_params* pp, dmd_stream_t stream) {
  const dmd_reset_slots_params& p = *pp;
  DMD_CHECK_ARG(p.B >= 1 && p.K >= 0 && p.T >= 1 && p.T <= 256 && p.head >= 0 && p.head < p.T && p.per_frame > 0 && p.per_frame % 4 == 0,
                "reset_slots: B %d, K %d, T %d, head %d, per_frame %lld", p.B, p.K, p.T, p.head, (long long)p.per_frame);
  DMD_CHECK_ARG(p.row_slot && p.next_obs && p.ctx && p.enc_in, "reset_slots: null");
  const dmd_pool_round& p0 = p.pool[0];
  DMD_CHECK_ARG(p.K == 0 || (p.slot_row && p0.frames && p0.act && p0.hx && p0.cx && p.act_ring && p.hx && p.cx && p.hd >= 1 && p.pool_base >= 0),
                "reset_slots: null pool / state argument with K = %d slots", p.K);
  const dmd_pool_round& p1 = p.pool[1];
  DMD_CHECK_ARG(p.K == 0 || !p1.frames || (p.num_dead && p1.act && p1.hx && p1.cx && p0.rows >= 1), "reset_slots: second pool round without "
                "its actions / states, the first round's row count or the device's death count");
  const int frames = p.B + p.T * p.K;
  DMD_CHECK_ARG(frames <= 65535, "reset_slots: %d frames in one launch", frames);
  const dim3 grid((unsigned)nblk((size_t)(p.per_frame / 4), 256), (unsigned)frames);
  hipLaunchKernelGGL(reset_slot_frames_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  DMD_LAUNCH_CHECK();
  if (p.K > 0) {
    hipLaunchKernelGGL(reset_slot_state_kernel, dim3((unsigned)p.K), dim3(256), 0, (hipStream_t)stream, p);
    DMD_LAUNCH_CHECK();
  }
  return 0;
}

// out[r] = alive row ? base[r] : slots[row_slot[r]]   (rows of D floats): the burnt-in LSTM state of the reset rows merged into
// the batch's state (reference env_loop.py:51-56: gate to zero, then burn in) -- and its transpose for the backward:
// d_base[r] = alive ? d_out[r] : 0;  d_slots[j] = slot used ? d_out[slot_row[j]] : 0
__global__ void merge_slots_kernel(const float* __restrict__ base, const float* __restrict__ slots, const int32_t* __restrict__ row_slot,
                                   float* __restrict__ out, int B, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * D) return;
  const int r = i / D, c = i - (int64_t)r * D;
  const int s = row_slot[r];
  out[i] = s >= 0 ? slots[(int64_t)s * D + c] : base[i];
}

__global__ void merge_slots_bwd_kernel(const float* __restrict__ d_out, const int32_t* __restrict__ row_slot, const int64_t* __restrict__ slot_row,
                                       float* __restrict__ d_base, float* __restrict__ d_slots, int B, int K, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)B * D) {
    const int r = i / D;
    if (d_base) d_base[i] = row_slot[r] >= 0 ? 0.0f : d_out[i];
  } else if (i < (int64_t)(B + K) * D) {
    const int64_t k = i - (int64_t)B * D;
    const int j = k / D, c = k - (int64_t)j * D;
    const int64_t r = slot_row[j];
    d_slots[k] = r >= 0 ? d_out[r * D + c] : 0.0f;
  }
}

extern "C" int dmd_merge_slots(const float* base, const float* slots, const int32_t* row_slot, float* out, int B, int D, dmd_stream_t stream) {
  DMD_CHECK_ARG(base && slots && row_slot && out && B >= 1 && D >= 1, "merge_slots: args");
  hipLaunchKernelGGL(merge_slots_kernel, dim3(nblk((size_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, base, slots, row_slot, out, B, D);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_merge_slots_bwd(const float* d_out, const int32_t* row_slot, const int64_t* slot_row, float* d_base, float* d_slots, int B,
                                   int K, int D, dmd_stream_t stream) {
  DMD_CHECK_ARG(d_out && row_slot && slot_row && d_slots && B >= 1 && K >= 1 && D >= 1, "merge_slots_bwd: args");
  hipLaunchKernelGGL(merge_slots_bwd_kernel, dim3(nblk((size_t)(B + K) * D, 256)), dim3(256), 0, (hipStream_t)stream, d_out, row_slot,
                     slot_row, d_base, d_slots, B, K, D);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, int CPad, dmd_stream_t stream) {
  DMD_CHECK_ARG(in && out && CPad % 4 == 0 && CPad >= C, "nchw_to_nhwc: args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblk((size_t)N * H * W * (CPad / 4), 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, N, C, H * W, CPad);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, int CPad, dmd_stream_t stream) {
  DMD_CHECK_ARG(in && out && CPad >= C, "nhwc_to_nchw: args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk((size_t)N * H * W, 256)), dim3(256), 0, (hipStream_t)stream, in, out, N, C,
                     H * W, CPad);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_gn_stats(const float* x, double* stats, int N, int HW, int C, dmd_stream_t stream) {
  DMD_CHECK_ARG(x && stats && C % DMD_GN_GROUP == 0, "gn_stats: C %% 32");
  hipLaunchKernelGGL(gn_stats_kernel, dim3(C / DMD_GN_GROUP, N), dim3(256), 0, (hipStream_t)stream, x, stats, HW, C, 0, 0, 0);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_gn_stats_valid(const float* x, double* stats, int N, int H, int W, int valid_h, int valid_w, int C, dmd_stream_t stream) {
  DMD_CHECK_ARG(x && stats && C % DMD_GN_GROUP == 0, "gn_stats: C %% 32");
  DMD_CHECK_ARG(valid_h > 0 && valid_h <= H && valid_w > 0 && valid_w <= W, "gn_stats: valid extent %d x %d of %d x %d", valid_h, valid_w, H, W);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(C / DMD_GN_GROUP, N), dim3(256), 0, (hipStream_t)stream, x, stats, H * W, C, W, valid_h, valid_w);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_maxpool2(const float* x, float* out, uint8_t* argmax, double* out_stats, int N, int H, int W, int C,
                            dmd_stream_t stream) {
  DMD_CHECK_ARG(x && out && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2: args");
  hipLaunchKernelGGL(maxpool2_kernel, dim3(nblk((size_t)N * (H / 2) * (W / 2) * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, out, argmax, N, H, W, C);
  DMD_LAUNCH_CHECK();
  if (out_stats) return dmd_gn_stats(out, out_stats, N, (H / 2) * (W / 2), C, stream);
  return 0;
}

extern "C" int dmd_lstm_pointwise(const float* gates, const float* c_prev, float* h, float* c, int N, int Hd,
                                  dmd_stream_t stream) {
  DMD_CHECK_ARG(gates && c_prev && h && c, "lstm_pointwise: null");
  hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(nblk((size_t)N * Hd, 256)), dim3(256), 0, (hipStream_t)stream, gates, c_prev,
                     h, c, N, Hd);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_lstm_pointwise_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh,
                                      const float* dc, float* dgates, float* dc_prev, int N, int Hd, dmd_stream_t stream) {
  DMD_CHECK_ARG(gates && c_prev && c_new && dgates && dc_prev, "lstm_pointwise_bwd: null");
  hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3(nblk((size_t)N * Hd, 256)), dim3(256), 0, (hipStream_t)stream, gates,
                     c_prev, c_new, dh, dc, dgates, dc_prev, N, Hd);
  DMD_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmd_categorical_sample(const float* logits, const float* expo, int64_t* out, int N, int A,
                                      dmd_stream_t stream) {
  DMD_CHECK_ARG(logits && expo && out && A > 0, "categorical_sample: args");
  hipLaunchKernelGGL(categorical_sample_kernel, dim3(nblk((size_t)N, 64)), dim3(64), 0, (hipStream_t)stream, logits, expo, out,
                     N, A);
  DMD_LAUNCH_CHECK();
  return 0;
}
