/* diamond_hip.h -- C ABI of libdiamond_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for DIAMOND's imagined-rollout hot path.  The reference
 * (eloialonso/diamond) has no FFI: every op below replaces a chain of ATen calls issued by
 * the cited reference Python lines.  All pointers are DEVICE pointers (HBM) unless noted,
 * all floating-point tensors are fp32, activations are NHWC, every entry point is
 * asynchronous on the given hipStream_t, allocates nothing, synchronises nothing and is
 * graph-capturable.  Return value: 0 = launched, non-zero = invalid arguments (message via
 * dmd_last_error()).  No exceptions cross this boundary.
 */
#ifndef DIAMOND_HIP_H
#define DIAMOND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dmd_stream_t; /* hipStream_t */

#define DMD_PROLOGUE_NONE 0      /* conv consumes the source as stored                       */
#define DMD_PROLOGUE_NORM_SILU 1 /* GroupNorm(+FiLM|affine) then SiLU fused into the load    */
#define DMD_PROLOGUE_NORM 2      /* GroupNorm(+affine) only (attention pre-norm)             */

#define DMD_GN_GROUP 32 /* models/blocks.py:12 GN_GROUP_SIZE */

/* GroupNorm statistics travel between kernels as per-tile partial sums in fp64:
 * stats[((n * G + g) * T + t) * 2 + {0: sum x, 1: sum x^2}], G = C / 32, T = producer tiles
 * per image.  The consumer reduces the T partials in a fixed order (deterministic). */
typedef struct dmd_norm {
  const double* stats;   /* NULL <=> no normalisation                                  */
  int32_t stat_tiles;    /* T                                                          */
  int32_t mul_plus_one;  /* 1: y = xn * (1 + mul) + add  (AdaGroupNorm, blocks.py:41-45) */
                         /* 0: y = xn * mul + add        (nn.GroupNorm affine, :28-31)  */
  const float* mul;      /* [n * mul_stride + c]                                       */
  const float* add;      /* [n * add_stride + c]                                       */
  int64_t mul_stride;    /* 0 for per-channel parameters shared by the batch           */
  int64_t add_stride;
} dmd_norm;

typedef struct dmd_conv_src {
  const float* x;   /* NHWC (N, Hs, Ws, C) */
  int32_t C;        /* channels, multiple of 16 */
  int32_t prologue; /* DMD_PROLOGUE_* */
  dmd_norm norm;
} dmd_conv_src;

/* dmd_conv2d: 3x3 (pad 1, stride 1|2) or 1x1 convolution as an implicit GEMM on
 * v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), with
 *   - channel concatenation of up to two sources       (torch.cat, blocks.py:174; inner_model.py:46)
 *   - nearest x2 upsampling folded into the gather      (Upsample.forward blocks.py:108-110)
 *   - GroupNorm/AdaGroupNorm + SiLU fused into the load (blocks.py:143-144; inner_model.py:48)
 *   - bias, residual add (optionally of the normalised residual, blocks.py:72) in the epilogue
 *   - GroupNorm partial statistics of the OUTPUT emitted by the epilogue.
 * Replaces F.conv2d call sites blocks.py:18-19,96,109,119,133-138 / inner_model.py:36,41. */
typedef struct dmd_conv_params {
  int32_t N, H, W;   /* batch and OUTPUT spatial size (H, W multiples of 8)            */
  int32_t Cout;      /* real output channels                                           */
  int32_t CoutPad;   /* Cout rounded up to 16/32/64-channel groups (weights are padded) */
  int32_t taps;      /* 9: 3x3 pad 1;  1: 1x1 pad 0                                    */
  int32_t stride;    /* 1 or 2 (taps == 9 only)                                        */
  int32_t upsample;  /* 1: sources stored at (H/2, W/2), nearest x2 before the conv    */
  int32_t nsrc;      /* 1 or 2                                                         */
  dmd_conv_src src[2];
  const float* w;    /* packed [Cin_total/16][taps][CoutPad][16]  (see dmd_pack_conv_weight) */
  const float* bias; /* [CoutPad] or NULL                                               */
  const float* residual; /* NHWC (N, H, W, Cout) or NULL                                */
  dmd_norm residual_norm; /* optional GroupNorm(+affine) applied to the residual        */
  float* out;        /* NHWC (N, H, W, Cout), or NCHW (N, Cout, H, W) if out_nchw       */
  int32_t out_nchw;
  int32_t precision; /* DMD_PRECISION_*: arithmetic of the contraction (see below)       */
  double* out_stats; /* (N, Cout/32, T, 2) partial sums of the output, or NULL          */
  const void* w_f16; /* dmd_pack_conv_weight_f16x2 layout, or NULL (needed for F16X2)    */
  /* Optional FUSED SKIP PROJECTION (ResBlock.forward, blocks.py:133,147: `self.proj(x) + conv2(...)` with
   * x = cat(x_up, skip)): a 1x1 convolution of proj_nsrc raw NHWC sources of the OUTPUT's N, H, W, accumulated into the
   * same fp32 accumulators as the 3x3 contraction -- the (N, H, W, Cout) projection is never written to or read from
   * HBM.  Honoured only where dmd_conv2d_proj_eligible() says so; dmd_conv2d fails loudly otherwise. */
  int32_t proj_nsrc;      /* 0: none; 2: two sources                                     */
  int32_t proj_C[2];      /* channels of each source                                     */
  int32_t proj_reserved;
  const float* proj_x[2]; /* NHWC (N, H, W, proj_C[i]), no prologue                      */
  const void* proj_w_f16; /* dmd_pack_conv_weight_f16x2(k = 1) of the (Cout, sum C, 1, 1) weights */
  const float* proj_bias; /* [Cout] or NULL                                              */
  /* VALID EXTENT (ABI v6).  The kernels tile in 8 / 16-pixel blocks; the reference's U-Net runs on any size that is a
   * multiple of 2^num_down (it pads to that and crops, blocks.py:227-229,247), e.g. 72x72 with levels 72/36/18/9.  Such a
   * tensor is stored inside a larger (N, H, W, C) buffer whose H, W satisfy the tiling rule, and (valid_h, valid_w) <= (H, W)
   * is the part that exists: conv-input positions outside the valid extent read as ZERO (like the convolution's own
   * padding, whatever the buffer holds there and before any fused normalisation), GroupNorm counts and the emitted
   * partial statistics cover the valid extent only.  Everything is still WRITTEN for the whole (H, W) buffer: positions
   * outside the valid extent hold unspecified values that no consumer reads unmasked.  The valid extent is the OUTPUT's;
   * a source's is (2 valid_h, 2 valid_w) at stride 2 and (valid_h / 2, valid_w / 2) under `upsample`.  0, 0 = (H, W). */
  int32_t valid_h, valid_w;
} dmd_conv_params;

/* DMD_PRECISION_F32:   v_mfma_f32_16x16x4_f32, bit-for-bit a k-ordered fp32 fma chain.
 * DMD_PRECISION_F16X2: fp32 operands split into two fp16 pieces each (x = h + l), three
 *   v_mfma_f32_32x32x16_f16 per product into an fp32 accumulator: fp32-class accuracy
 *   (representation error <= max(2^-22 |x|, 2^-25)), 16x the MFMA rate.  Range contract: finite operands must
 *   be inside the fp16 range (|x| < 65520); nothing is clamped, an operand beyond it makes every output it
 *   touches NaN (loud, never a silently wrong finite value), NaN / Inf inputs stay non-finite.  Honoured only
 *   where dmd_conv2d_f16x2_eligible() says so (3x3 / 1x1 stride 1, Cout in {32, 64}, Cin <= 128 / 64, NHWC
 *   out, or the few-channel NCHW head); everything else uses the exact kernel. */
#define DMD_PRECISION_F32 0
#define DMD_PRECISION_F16X2 1
int dmd_conv2d_f16x2_eligible(const dmd_conv_params* p);
/* 1: dmd_conv2d can fuse the skip projection described by the proj_* fields into this launch (split-fp16 3x3 stride 1,
 * one 64-channel
 * source, Cout == 64, H, W multiples of 16, two 64-channel projection sources, no other residual) */
int dmd_conv2d_proj_eligible(const dmd_conv_params* p);
/* 1: dmd_conv2d runs these parameters on the streaming 1x1 kernel (exact fp32; taps == 1, Cout == 64, no
 * prologue / residual / statistics, Cin in {32, 64, 128}): blocks.py:120,133 skip projections */
int dmd_conv1x1_stream_eligible(const dmd_conv_params* p);
/* OIHW (Cout in {32, 64}, Cin, k, k) fp32, k in {1, 3} -> [CinPad/16][k*k][h|l][2][Cout][8] fp16 pieces */
int dmd_pack_conv_weight_f16x2(const float* oihw, void* packed, int Cout, int Cin, int k, int CinPad, dmd_stream_t stream);

/* Every kernel-layout copy of a model's convolution parameters in ONE launch (a training step changes all of them:
 * trainer.py:366-388).  A job = one copy; the table lives in DEVICE memory and is reusable as long as the pointers are.
 *   kind DMD_PACK_F32   : dmd_pack_conv_weight's layout        kind DMD_PACK_F16X2: dmd_pack_conv_weight_f16x2's
 *   kind DMD_PACK_BIAS  : (CoutPad) fp32 <- bias (Cout), zero beyond
 *   transposed = 1      : the weight of the data-gradient convolution restricted to input channels [c0, c1):
 *                         W'[ci - c0][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx]; CoutPad / CinPad then refer to W'
 *                         (outputs c1 - c0, inputs Cout zero-padded to CinPad).
 * max_elems = the largest job's element count (fp32: CinPad * taps * CoutPad, f16x2: CinPad * taps * CoutPad, bias: CoutPad). */
#define DMD_PACK_F32 0
#define DMD_PACK_F16X2 1
#define DMD_PACK_BIAS 2
typedef struct dmd_pack_job {
  const float* src;      /* OIHW (Cout, Cin, k, k) fp32, or the bias (Cout) */
  void* dst;
  int32_t Cout, Cin, k;  /* of src */
  int32_t kind, transposed, c0, c1;
  int32_t CoutPad, CinPad; /* of the packed (possibly transposed) weight */
  int32_t reserved;
} dmd_pack_job;
int dmd_pack_jobs(const dmd_pack_job* jobs_device, int njobs, int64_t max_elems, dmd_stream_t stream);

/* ABI v9: fingerprints of parameter storage, for the always-on audit of the packed copies (engine.WeightAudit): a copy is keyed on
 * the parameter's version counter and storage pointer, and a write that changes neither (`p.data.copy_`, a collective on
 * `.data`) must not go unnoticed.  out[j * DMD_CHECKSUM_PARTS + b] = the sum, in 64 bits, of the 32-bit words w of jobs[j].src
 * (as signed integers) with (w / 256) % DMD_CHECKSUM_PARTS == b: exact and independent of the launch.  No reference
 * counterpart (torch keeps no second copy of a parameter: F.conv2d reads the parameter itself, /root/reference/src/models/blocks.py:18-31). */
#define DMD_CHECKSUM_PARTS 8
typedef struct dmd_checksum_job {
  const void* src;
  int64_t words; /* 32-bit words */
} dmd_checksum_job;
int dmd_checksums(const dmd_checksum_job* jobs_device, int njobs, long long* out_device, dmd_stream_t stream);

int dmd_conv2d(const dmd_conv_params* p, dmd_stream_t stream);
/* name of the kernel instantiation dmd_conv2d launches for these parameters, spelled like rocprofv3's kernel trace
 * (e.g. "conv_f16ws_kernel<WsGeom<false, 2, 9>>"): measurement plumbing for bench.py / profiles */
int dmd_conv2d_kernel_name(const dmd_conv_params* p, char* buf, int buf_len);
/* number of GroupNorm stat tiles per image a dmd_conv2d with output (H, W) emits */
int dmd_conv_stat_tiles(int H, int W);
/* OIHW (Cout, Cin, k, k) fp32 -> packed layout; Cin padded to 16, Cout padded to CoutPad. */
int dmd_pack_conv_weight(const float* oihw, float* packed, int Cout, int Cin, int k, int CoutPad, int CinPad,
                         dmd_stream_t stream);

/* Debug/verification twin of dmd_conv2d: one thread per output element, plain fp32 fma
 * loops, identical parameters and semantics (used by tests to localise MFMA-path bugs). */
int dmd_conv2d_naive(const dmd_conv_params* p, dmd_stream_t stream);

/* dmd_linear: C[M,N] (+)= A[M,K] . W[N,K]^T + bias[N], optional SiLU.  nn.Linear call sites:
 * AdaGroupNorm.linear blocks.py:39,44 (all of a forward batched into one call), cond_proj
 * inner_model.py:31-35, LSTM gates, actor/critic heads actor_critic.py:47-48,73. */
typedef struct dmd_linear_params {
  int32_t M, N, K;   /* K multiple of 16 */
  const float* A; int64_t lda;
  const float* W; int64_t ldw;
  const float* bias; /* [N] or NULL */
  float* C; int64_t ldc;
  int32_t accumulate; /* 1: C += ... */
  int32_t silu;       /* 1: C = silu(...) */
} dmd_linear_params;
int dmd_linear(const dmd_linear_params* p, dmd_stream_t stream);

/* dmd_attention: softmax(q k^T / sqrt(d)) v per (image, head) over T tokens, streaming
 * K/V tiles through LDS with an online softmax.  qkv is NHWC (N, T, 3C): q | k | v channel
 * thirds, head h = channels [h*d, (h+1)*d) of each third (blocks.py:66-71).  d == 8. */
int dmd_attention(const float* qkv, float* out, int N, int T, int C, int head_dim, dmd_stream_t stream);
/* ... over an (H, W) token grid of which only (valid_h, valid_w) exists (see dmd_conv_params: VALID EXTENT): keys outside it
 * do not take part in the softmax; outputs of queries outside it are unspecified */
int dmd_attention_valid(const float* qkv, float* out, int N, int H, int W, int valid_h, int valid_w, int C, int head_dim,
                        dmd_stream_t stream);
/* Backward of dmd_attention (autograd of blocks.py:66-71 under the denoiser training loss, denoiser.py:93-122):
 * y = the forward's output, dy its gradient -> dqkv (N, T, 3C) in the qkv layout.  workspace: dmd_attention_bwd_workspace_floats. */
int64_t dmd_attention_bwd_workspace_floats(int N, int T, int C);
int dmd_attention_bwd(const float* qkv, const float* y, const float* dy, float* dqkv, float* workspace, int N, int T, int C,
                      int head_dim, dmd_stream_t stream);

/* ---- dmd_lowres_chain: a whole chain of ResBlocks at the 8x8 level of the denoiser's U-Net in ONE launch ----
 * Replaces, for H = W = 8 and 64 channels, the launches of the deepest level of UNet.forward (reference blocks.py:232-246:
 * d_blocks[-1], mid_blocks, u_blocks[0]) -- per ResBlock (blocks.py:141-147): optional 1x1 `proj` of the raw (concatenated)
 * input, conv1(SiLU(AdaGN1(cat(x, skip)))), conv2(SiLU(AdaGN2(h))) + r, optional SelfAttention2d (blocks.py:62-72).
 * One workgroup per image keeps every activation of the chain in LDS (64 pixels x 64 channels = 16 KiB per tensor); the
 * convolutions use the split-fp16 MFMA arithmetic of dmd_conv2d's DMD_PRECISION_F16X2 path and the same packed weights
 * (dmd_pack_conv_weight_f16x2).  At batch 256 these launches are latency chains of ~25 us each for 1.2 GFLOP. */
#define DMD_CHAIN_MAX_BLOCKS 8
typedef struct dmd_chain_block {
  int32_t skip_slot;    /* -1: the block's input is x;  0..2: cat(x, saved[skip_slot]) (128 input channels)            */
  int32_t save_slot;    /* -1, or 0..2: keep the block's OUTPUT in this slot for a later concatenation                  */
  int32_t film1_mul[2]; /* FiLM-table column of AdaGN1's `scale` for the x part / the skip part                          */
  int32_t film1_add[2]; /* ... of `shift`                                                                                */
  int32_t film2_mul, film2_add;
  int32_t has_attn, reserved;
  const void* w1;       /* conv1 3x3, Cin 64 | 128 -> 64, dmd_pack_conv_weight_f16x2 layout                              */
  const void* w2;       /* conv2 3x3, 64 -> 64                                                                           */
  const void* wproj;    /* 1x1 128 -> 64 (f16x2 pack) or NULL (identity skip)                                            */
  const float* b1;      /* (64)                                                                                          */
  const float* b2;      /* (64)                                                                                          */
  const float* bproj;   /* (64) or NULL                                                                                  */
  const float* gn_gamma; /* attention: nn.GroupNorm affine (64), (64)                                                    */
  const float* gn_beta;
  const void* wq;       /* attention: 1x1 64 -> 64 packs of the q | k | v thirds of qkv_proj and of out_proj             */
  const void* wk;
  const void* wv;
  const void* wo;
  const float* bqkv;    /* (192) */
  const float* bo;      /* (64)  */
} dmd_chain_block;

typedef struct dmd_lowres_chain_params {
  int32_t N;                 /* images                                                            */
  int32_t nblocks;           /* <= DMD_CHAIN_MAX_BLOCKS                                           */
  int32_t input_save_slot;   /* -1, or the slot that keeps the chain INPUT for a later concatenation */
  int32_t reserved;
  const float* x;            /* NHWC (N, 8, 8, 64)                                                */
  float* out;                /* NHWC (N, 8, 8, 64)                                                */
  const float* table;        /* FiLM table (N, table_stride): every AdaGroupNorm.linear(cond) of the network, concatenated */
  int64_t table_stride;
  dmd_chain_block blocks[DMD_CHAIN_MAX_BLOCKS];
} dmd_lowres_chain_params;
int dmd_lowres_chain(const dmd_lowres_chain_params* p, dmd_stream_t stream);
/* The same for the 8x8 x 32-channel tail of the reward / end model's encoder (reference rew_end_model.py:93-133: the last
 * level's ResBlocks and the final attention ResBlocks; no concatenated inputs): x / out are NHWC (N, 8, 8, 32), packs are the
 * 32-cout form of dmd_pack_conv_weight_f16x2, q | k | v are 32-channel thirds (4 heads). */
int dmd_lowres_chain32(const dmd_lowres_chain_params* p, dmd_stream_t stream);

/* ---- EDM preconditioning / sampler pointwise (denoiser.py:74-84, diffusion_sampler.py:45-56) ---- */
/* The four EDM conditioners (c_in, c_out, c_skip, c_noise; compute_conditioners denoiser.py:66-72)
 * are four scalars per sample: the host evaluates them with the reference's own fp32 op order
 * (bit-exact; v_sqrt_f32 is not correctly rounded) and hands them over as a device array
 * cond[n * cond_stride + {0: c_in, 1: c_out, 2: c_skip, 3: c_noise}], cond_stride in {0, 4}. */

/* cat(obs / sigma_data, x * c_in) -> NHWC with CPad channels (zero padded).
 * x (N, Cx, H, W), obs (N, Cobs, H, W) NCHW.  obs may be a RING of T conditioning frames (N, T, Cobs / T, H, W):
 * logical frame t is stored at slot (head + t) % T, so that WorldModelEnv.step never rolls its context
 * (world_model_env.py:74-75).  T = 1, head = 0: plain tensor. */
int dmd_edm_pack_input(const float* x, const float* obs, const float* cond, int cond_stride, float sigma_data,
                       float* out_nhwc, int N, int Cx, int Cobs, int H, int W, int CPad, int T, int head,
                       dmd_stream_t stream);
/* cond input: fourier(c_noise) + flatten(embedding(act))  (blocks.py:84-87, inner_model.py:27-30,45);
 * act is a ring of T actions per sample starting at act_head (0: plain (N, T) tensor); A = rows of act_emb
 * (indices are clamped to [0, A) for memory safety -- validating them is the caller's job) */
int dmd_cond_embed(const float* cond, int cond_stride, const float* fourier_w /*[half]*/,
                   const int64_t* act /*(N, T)*/, const float* act_emb /*(A, E)*/, float* out /*(N, 2*half)*/, int N,
                   int half, int T, int E, int act_head, int A, dmd_stream_t stream);
/* denoised = quantise(c_skip * x + c_out * F)  (denoiser.py:81-83); all NCHW, elementwise */
int dmd_edm_denoised(const float* x, const float* model_out, const float* cond, int cond_stride,
                     float* denoised, int N, int64_t per_sample, dmd_stream_t stream);
/* x_out = x + ((x - denoised) / sigma_hat) * dt   (diffusion_sampler.py:45-49) */
int dmd_euler_step(const float* x, const float* denoised, float sigma_hat, float dt, float* x_out, int64_t n,
                   dmd_stream_t stream);
/* Heun combine, diffusion_sampler.py:52-56: d = (x - D) / sigma_hat, d_2 = (x_2 - D_2) / sigma_next,
 * x_out = x + ((d + d_2) / 2) * dt   (same fp32 op order as the reference) */
int dmd_heun_step(const float* x, const float* denoised, const float* x_2, const float* denoised_2, float sigma_hat,
                  float sigma_next, float dt, float* x_out, int64_t n, dmd_stream_t stream);

/* ---- device-resident uint8 pool of initial conditions (world_model_env.py:107-139; frames are uint8 on disk,
 *      data/episode.py:36-50, and `x.div(255).mul(2).sub(1)` when loaded) ------------------------------------- */
/* q = round((x + 1) / 2 * 255); *off_grid (device int, zeroed by the caller) becomes 1 if some x is not exactly the
 * dequantisation of its level -- such a pool cannot be kept as uint8 without changing results. */
int dmd_quantize_u8(const float* x, uint8_t* q, int* off_grid, int64_t n, dmd_stream_t stream);
/* dst ring row rows[i] (NULL: i), slot (head + t) % T  <-  (pool[idx[i]][t] / 255) * 2 - 1   for i < M, t < T.
 * pool (P, T, per_frame) uint8, dst (B, T, per_frame) fp32, per_frame = C * H * W (multiple of 4). */
int dmd_dequant_gather(const uint8_t* pool, const int64_t* idx, const int64_t* rows, float* dst, int M, int T,
                       int64_t per_frame, int head, dmd_stream_t stream);
/* ABI v9: the rest of a reset (envs/world_model_env.py:56-62 `reset_dead`) in one launch, for i < count, r = rows[i], q = idx[i]:
 * act_ring[r][(head + t) % T] = pool_act[q][t] (int64 (B, T) / (P, T)); hx[r] = pool_hx[q], cx[r] = pool_cx[q] ((B, hd) / (P, hd)
 * fp32: the reward/end LSTM state); ep_len[r] = 0. */
int dmd_reset_state(const int64_t* idx, const int64_t* rows, int count, const int64_t* pool_act, int64_t* act_ring, int T, int head,
                    const float* pool_hx, const float* pool_cx, float* hx, float* cx, int hd, int64_t* ep_len, dmd_stream_t stream);

/* ---- ABI v11: the deaths of an imagined step resolved on the device -----------------------------------------------------
 * Replaces the reference's per-step host round trip `if dead.any(): reset_dead(dead)` (envs/world_model_env.py:77-83, :56-62) and
 * what coroutines/env_loop.py:45-56 does with its answer: the step works on K SLOTS chosen by the host BEFORE it knows the step's
 * deaths; dead rows take slots in ascending row order (the order of the reference's boolean masks and of its pool,
 * world_model_env.py:133-139), unused slots are -1, and the host reads `report` one step late.
 *   ep_len[r] += 1; trunc[r] = ep_len[r] >= horizon (:71-72); dead[r] = end[r] | trunc[r]; ep_len[r] = 0 where dead (:61)
 *   slot_row (K) int64: j-th dead row or -1;  row_slot (B) int32: slot of a dead row, -1 if alive (or no slot was left)
 *   report (B + 4) int32: dead[0..B), k = number of dead rows, rows with end != 0, k > K (slot overflow), K */
int dmd_resolve_deaths(const int64_t* end, int64_t* ep_len, int horizon, int64_t* trunc, uint8_t* dead, int B, int K,
                       int64_t* slot_row, int32_t* row_slot, int32_t* report, dmd_stream_t stream);

/* Ring advance (world_model_env.py:74-75, as a ring: the oldest slot receives the imagined frame), reset of the slots' rows
 * from pool rows pool_base + j (:56-62: context frames, actions, reward/end LSTM state), and the policy's next input
 * enc_in (B + T * K, per_frame) = [ newest frame of every env (a reset row: of its NEW episode)
 *                                  | final observation of slot j (env_loop.py:49)
 *                                  | burn-in frame t of slot j, frame-major, t < T - 1 (env_loop.py:53-56) ]
 * in one call (two launches).  head = the ring's head AFTER the advance.  A pool round: frames (P, T, per_frame) uint8 (dequantised
 * as dmd_dequant_gather does; pad (P, T) uint8, optional: frames that are exact zeros) or fp32 (is_f32), act (P, T) int64, hx / cx
 * (P, hd) fp32.  TWO rounds may be given: the reference drops the rest of a round and preloads the next one when a request does not
 * fit (world_model_env.py:133-139) -- a decision that depends on the step's number of deaths, which only the device knows at launch
 * time: with pool[1].frames and num_dead set, the slots are served from pool[1] rows 0.. iff pool_base + *num_dead > pool[0].rows.
 * Unused slots produce copies of row 0's imagined frame in enc_in and touch nothing else. */
typedef struct dmd_pool_round {
  const void* frames;
  const uint8_t* pad;
  const int64_t* act;
  const float* hx;
  const float* cx;
  int32_t is_f32, rows;
} dmd_pool_round;
typedef struct dmd_reset_slots_params {
  int32_t B, K, T, head;
  int64_t per_frame; /* C * H * W, a multiple of 4 */
  int32_t hd, reserved; /* hd: width of the reward/end LSTM state */
  dmd_pool_round pool[2];
  int64_t pool_base;       /* first row of pool[0] this step would be served from */
  const int32_t* num_dead; /* device: the step's number of dead rows (dmd_resolve_deaths' report + B); NULL with one round */
  const int64_t* slot_row; /* (K), from dmd_resolve_deaths */
  const int32_t* row_slot; /* (B) */
  const float* next_obs;   /* (B, per_frame): the imagined frames of this step */
  float* ctx;              /* (B, T, per_frame) context ring */
  int64_t* act_ring;       /* (B, T) */
  float* hx;               /* (B, hd) reward/end LSTM state */
  float* cx;
  float* enc_in;           /* (B + T * K, per_frame) */
} dmd_reset_slots_params;
int dmd_reset_slots(const dmd_reset_slots_params* p, dmd_stream_t stream);

/* out[r] = row_slot[r] >= 0 ? slots[row_slot[r]] : base[r]  (rows of D floats): the burnt-in policy LSTM state of the reset rows
 * merged into the batch's state (env_loop.py:51-56), and the transposed operation for its backward:
 * d_base[r] = row_slot[r] >= 0 ? 0 : d_out[r] (d_base may be NULL);  d_slots[j] = slot_row[j] >= 0 ? d_out[slot_row[j]] : 0 */
int dmd_merge_slots(const float* base, const float* slots, const int32_t* row_slot, float* out, int B, int D, dmd_stream_t stream);
int dmd_merge_slots_bwd(const float* d_out, const int32_t* row_slot, const int64_t* slot_row, float* d_base, float* d_slots, int B, int K,
                        int D, dmd_stream_t stream);

/* NCHW (N, C, H, W) -> NHWC (N, H, W, CPad), zero padded channels */
int dmd_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, int CPad, dmd_stream_t stream);
int dmd_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, int CPad, dmd_stream_t stream);

/* GroupNorm partial statistics of an NHWC tensor (one tile per image): for tensors that
 * no dmd_* kernel produced. */
int dmd_gn_stats(const float* x, double* stats, int N, int HW, int C, dmd_stream_t stream);
/* ... of the (valid_h, valid_w) part of an (N, H, W, C) buffer (dmd_conv_params: VALID EXTENT) */
int dmd_gn_stats_valid(const float* x, double* stats, int N, int H, int W, int valid_h, int valid_w, int C, dmd_stream_t stream);

/* 2x2 max pooling, NHWC, also emits GroupNorm stats of the pooled output and the argmax
 * (0..3) for the backward pass (actor_critic.py:108-109). */
int dmd_maxpool2(const float* x, float* out, uint8_t* argmax, double* out_stats, int N, int H, int W, int C,
                 dmd_stream_t stream);

/* LSTM cell pointwise: gates (N, 4*Hd) in i,f,g,o order -> h, c (nn.LSTMCell, actor_critic.py:46,72) */
int dmd_lstm_pointwise(const float* gates, const float* c_prev, float* h, float* c, int N, int Hd,
                       dmd_stream_t stream);

/* backward of dmd_lstm_pointwise: gates = the forward's pre-activations, dh / dc = gradients w.r.t. the new h / c
 * (NULL = zero) -> dgates (N, 4*Hd), dc_prev (N, Hd)   (autograd of nn.LSTMCell, actor_critic.py:46,72) */
int dmd_lstm_pointwise_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh, const float* dc,
                           float* dgates, float* dc_prev, int N, int Hd, dmd_stream_t stream);

/* argmax(softmax(logits) / E) with injected exponential draws E == Categorical(logits).sample()
 * (env_loop.py:32, world_model_env.py:103-104). */
int dmd_categorical_sample(const float* logits, const float* expo, int64_t* out, int N, int A, dmd_stream_t stream);

/* ---- actor-critic encoder backward (what ATen autograd does for actor_critic.py:101-113 /
 *      blocks.py:116-123 under loss.backward(), trainer.py:366) ------------------------------ */

/* dx (N, H, W, C) <- scatter of dpooled (N, H/2, W/2, C) by the argmax dmd_maxpool2 saved */
int dmd_maxpool2_bwd(const float* dpooled, const uint8_t* argmax, float* dx, int N, int H, int W, int C,
                     dmd_stream_t stream);

/* Backward of a = act(GroupNorm(x) * mul' + add), act = SiLU or identity (mul' = mul or 1 + mul, as in dmd_norm):
 *   dx = d a / d x applied to da (+ dskip, the gradient of the residual branch around the block)
 *   dmul[n][c] = sum_hw du * xhat,  dadd[n][c] = sum_hw du   (the caller sums over n for affine
 *   parameters shared by the batch). */
typedef struct dmd_gn_bwd_params {
  int32_t N, HW, C;
  int32_t identity_activation; /* 0: a = SiLU(u) (AdaGroupNorm / SmallResBlock); 1: a = u (attention pre-norm, blocks.py:64) */
  const float* x;      /* NHWC input of the GroupNorm                          */
  dmd_norm norm;       /* its statistics + multiplicative/additive parameters  */
  const float* da;     /* gradient w.r.t. the activated tensor                 */
  const float* dskip;  /* NULL or NHWC, added to dx                            */
  float* dx;
  void* workspace;     /* dmd_gn_bwd_workspace_bytes(N, HW, C) bytes           */
  float* dmul;         /* (N, C) */
  float* dadd;         /* (N, C) */
  /* VALID EXTENT (ABI v8; all 0: the whole tensor): x is (N, HW / W, W, C) of which rows < valid_h, columns < valid_w exist.
   * Only those pixels enter the sums and the GroupNorm count; dx is written as ZERO outside (so that what consumes it --
   * max-pool backward, the next weight gradient -- sees no contribution from there). */
  int32_t W, valid_h, valid_w, reserved;
} dmd_gn_bwd_params;
int64_t dmd_gn_bwd_workspace_bytes(int N, int HW, int C);
int dmd_gn_silu_bwd(const dmd_gn_bwd_params* p, dmd_stream_t stream);

/* Weight/bias gradient of a stride-1 dmd_conv2d (3x3 pad 1 or 1x1) with a single source:
 *   dw[co][ci][ky][kx] = sum_{n,y,x} dy[n,y,x,co] * a[n, y+ky-1, x+kx-1, ci],  db[co] = sum dy,
 * a = the conv's (prologue-activated) input, recomputed from src exactly as the forward does. */
typedef struct dmd_wgrad_params {
  int32_t N, H, W;
  int32_t Cout;        /* multiple of 16 */
  int32_t taps;        /* 9 | 1 */
  int32_t cin_real;    /* input channels of the OIHW gradient (<= src.C, the padded count) */
  dmd_conv_src src;
  const float* dy;     /* NHWC (N, H, W, Cout) */
  float* workspace;    /* dmd_wgrad_workspace_floats(p) floats */
  float* dw;           /* OIHW (Cout, cin_real, k, k) */
  float* dbias;        /* (Cout) or NULL */
  int32_t precision;   /* DMD_PRECISION_F32 (exact fp32 fma chain) | DMD_PRECISION_F16X2 (split-fp16 operands, fp32 accumulate) */
  /* VALID EXTENT (ABI v8; both 0: the whole tensor): rows < valid_h, columns < valid_w of src and dy exist; positions outside are
   * the convolution's zero padding (input, after the prologue) / contribute nothing (dy), GroupNorm counts the valid pixels. */
  int32_t valid_h, valid_w;
  /* ABI v10 (was reserved, 0): 1 = stop after the per-workgroup partials; they stay in `workspace` (which the caller keeps alive)
   * until a dmd_wgrad_reduce_jobs launch sums them into dw / dbias -- the job comes from dmd_wgrad_job(p). */
  int32_t defer_reduce;
} dmd_wgrad_params;
int64_t dmd_wgrad_workspace_floats(const dmd_wgrad_params* p);
int dmd_conv2d_wgrad(const dmd_wgrad_params* p, dmd_stream_t stream);

/* Deferred reductions of weight gradients (ABI v10).  The reference leaves every weight gradient to autograd's per-layer kernels
 * (src/models/diffusion/denoiser.py:93-122 -> loss.backward(), src/trainer.py:349-388); here the partial sums of ALL layers of a
 * backward are reduced by one launch per 32 gradients once the last dmd_conv2d_wgrad(defer_reduce = 1) is issued.  The sums are
 * formed in the order of the undeferred call: the gradients are bit-identical.  A job names where its gradient lands:
 * element (co, ci, tap) -> dw[(co * ld_cin + c0 + ci) * taps + tap], so the sources of a convolution over concatenated inputs
 * (ld_cin = all input channels, c0 = this source's first) and the 64-channel pieces of a wide output (dw offset by whole rows)
 * fill ONE OIHW tensor.  The table is HOST memory: it travels in the kernel arguments. */
typedef struct dmd_wgrad_reduce_job {
  const float* partials; /* the workspace of the deferred dmd_conv2d_wgrad */
  float* dw;
  float* dbias;          /* or NULL */
  int32_t num_wg, NB, NCO, NCI, taps, cin_real; /* filled by dmd_wgrad_job */
  int32_t ld_cin, c0;    /* dmd_wgrad_job: cin_real, 0 */
} dmd_wgrad_reduce_job;
/* The job of dmd_conv2d_wgrad(p, defer_reduce = 1): its pointers are p's (which may still be null -- a deferred call needs only
 * num_wg * (NB * NCO * 256 + NCO * 16) floats of workspace, less than dmd_wgrad_workspace_floats, and the caller may size it from
 * the job before it sets p->workspace and job->partials). */
int dmd_wgrad_job(const dmd_wgrad_params* p, dmd_wgrad_reduce_job* job);
int dmd_wgrad_reduce_jobs(const dmd_wgrad_reduce_job* jobs, int njobs, dmd_stream_t stream);

const char* dmd_last_error(void);
int dmd_abi_version(void);
/* The library reads its DIAMOND_* switches from the environment once; this makes it read them again (test hook). */
void dmd_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif /* DIAMOND_HIP_H */
