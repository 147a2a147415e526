// conv1x1_stream_kernel -- 1x1 convolution with 64 output channels as a STREAMING GEMM (exact fp32,
// v_mfma_f32_16x16x4_f32): the U-Net's skip projections (blocks.py:133, 128 -> 64 over cat(x, skip)) and the
// actor-critic's skip_projection (blocks.py:120).
//
// These layers are HBM-bound (805 MB per launch at the 64x64 level for 17 GFLOP), and under load an HBM access
// takes ~4 us on this chip: bandwidth = bytes in flight / latency.  The LDS-staged conv kernels keep ~50 KB per
// CU in flight and stall at ~3 TB/s; this kernel keeps every operand of a tile in flight at once instead:
//   * NHWC activations ARE the GEMM's K-contiguous operand: lane (pixel j, k-group kg) reads the 16 bytes
//     x[pixel][k0 + 4 kg ..] straight from global memory -- no LDS staging, no barrier in the K loop;
//   * a wave owns 64 pixels x 64 couts and issues ALL its loads (Cin / 16 steps x 4 pixel blocks, up to 32
//     independent 16-byte loads per lane = 32 KB per wave) before the first MFMA;
//   * weights (64 x Cin fp32, <= 32 KB) sit in LDS for the lifetime of the persistent workgroup.
#include "dmd_common.h"

#define C1X1_CIN_MAX 128
// Development switches, all measured on the 64x64 level (805 MB per launch, default 189 us = 4.3 TB/s) and left off:
// nontemporal loads 203 us, nontemporal stores 186 us, PB = 4 at Cin = 128 196 us, 512 / 2048 persistent workgroups
// 187 / 189 us; timing proxies with fully coalesced 1 KiB-per-instruction loads 189 us, loads and stores 177 us.  The
// practical ceiling of this 2 : 1 read : write mix on the chip is 5.0 TB/s = 160 us (tools/probe/read_bw_probe.hip:
// 6.0 TB/s read-only, the same through registers and through LDS-DMA), so the kernel sits at 85 % of it.
// (the C1X1_* knobs below are compile-time lab settings: only a -DDMD_LAB build may change them)
#ifndef DMD_LAB
#undef C1X1_NT_LOAD
#undef C1X1_ABL_LINEAR
#undef C1X1_NT_STORE
#undef C1X1_PB128
#undef C1X1_NWG
#endif
#ifndef C1X1_NT_LOAD
#define C1X1_NT_LOAD 0
#endif
#ifndef C1X1_ABL_LINEAR
#define C1X1_ABL_LINEAR 0
#endif
#ifndef C1X1_NT_STORE
#define C1X1_NT_STORE 0
#endif
#ifndef C1X1_PB128
#define C1X1_PB128 2
#endif
#ifndef C1X1_NWG
#define C1X1_NWG 1024
#endif

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// SPLIT (DMD_PRECISION_F16X2): the exact-fp32 MFMA costs 131 us of matrix-pipe time at the 64x64 level -- as much as
// the memory time -- so the no-grad world-model launches use the split-fp16 form here too: operands x = h + l fp16
// pieces (converted in registers, the fragment layout of v_mfma_f32_16x16x16_f16 is the fp32 one: 4 consecutive k
// per lane), three MFMAs per product into the fp32 accumulator.

// NK = Cin / 16 (2, 4, 8); PB = 16-pixel blocks per wave (tile = 4 waves x PB x 16 pixels): PB = 2 at Cin = 128 keeps
// the tile's operands (NK x PB float4) + accumulators inside 128 registers -> 4 waves per SIMD, 16 KB in flight each
template <int NK, int PB, bool SPLIT>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const dmd_conv_params p, int tiles, int tiles_per_wg) {
  __shared__ f32x4 wlds[64 * NK * 4];  // [k-step][kg][cout]: unit (ks * 4 + kg) * 64 + cout; SPLIT: {h4 hi, h4 lo} bits
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int C0 = p.src[0].C;
  const int nk0 = C0 >> 4;
  const int Cin = 16 * NK;
  // weights: p.w is the dmd_pack_conv_weight layout [Cin/16][taps = 1][CoutPad = 64][16]
  for (int u = tid; u < 64 * NK * 4; u += 256) {
    const int co = u & 63, kq = (u >> 6) & 3, ks = u >> 8;
    f32x4 wv = *(const f32x4*)(p.w + ((size_t)ks * 64 + co) * 16 + 4 * kq);
    if (SPLIT) {
      h4 hh, ll;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = wv[e];
        hh[e] = (_Float16)x;
        ll[e] = (_Float16)(x - (float)hh[e]);
      }
      const uint2 a = __builtin_bit_cast(uint2, hh), b = __builtin_bit_cast(uint2, ll);
      wv = __builtin_bit_cast(f32x4, (uint4){a.x, a.y, b.x, b.y});
    }
    wlds[u] = wv;
  }
  f32x4 bias[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) bias[cb] = p.bias ? *(const f32x4*)(p.bias + cb * 16 + 4 * kg) : (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const size_t npix = (size_t)p.N * p.H * p.W;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(tiles, t_begin + tiles_per_wg);
  for (int t = t_begin; t < t_end; ++t) {
    const size_t pix0 = ((size_t)t * 4 + wave) * (PB * 16);  // this wave's pixels
    // ---- all loads of the tile in flight ----
    f32x4 xf[NK][PB];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const dmd_conv_src& sc = p.src[ks < nk0 ? 0 : 1];
      const int c0 = (ks < nk0 ? ks : ks - nk0) * 16 + 4 * kg;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        size_t pix = pix0 + pb * 16 + j;
        pix = pix < npix ? pix : npix - 1;  // clamp: duplicates are never stored
#if C1X1_ABL_LINEAR  // timing proxy (WRONG results): the same bytes as fully coalesced 1 KiB-per-instruction reads
        xf[ks][pb] = *((const f32x4*)(sc.x + pix0 * sc.C) + (size_t)((ks < nk0 ? ks : ks - nk0) * PB + pb) * 64 + lane);
#elif C1X1_NT_LOAD
        xf[ks][pb] = __builtin_nontemporal_load((const f32x4*)(sc.x + pix * sc.C + c0));
#else
        xf[ks][pb] = *(const f32x4*)(sc.x + pix * sc.C + c0);
#endif
      }
    }
    h4 xh[SPLIT ? NK : 1][PB], xl[SPLIT ? NK : 1][PB];
    if (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = xf[ks][pb][e];  // no clamp: see the range contract in dmd_conv_f16ws.hip
            const _Float16 h = (_Float16)x;
            xh[ks][pb][e] = h;
            xl[ks][pb][e] = (_Float16)(x - (float)h);
          }
    }
    f32x4 acc[4][PB];  // [cout block][pixel block]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) acc[cb][pb] = bias[cb];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const f32x4 wf = wlds[(ks * 4 + kg) * 64 + cb * 16 + j];  // A operand: row = cout j of block cb, k-group kg
        if (SPLIT) {
          const uint4 wb = __builtin_bit_cast(uint4, wf);
          const h4 wh = __builtin_bit_cast(h4, (uint2){wb.x, wb.y}), wl = __builtin_bit_cast(h4, (uint2){wb.z, wb.w});
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, xl[ks][pb], acc[cb][pb], 0, 0, 0);
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl, xh[ks][pb], acc[cb][pb], 0, 0, 0);
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, xh[ks][pb], acc[cb][pb], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], xf[ks][pb][e], acc[cb][pb], 0, 0, 0);
        }
      }
    }
    // D rows = cout (4 kg + r) of block cb, column = pixel j of block pb
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const size_t pix = pix0 + pb * 16 + j;
      if (pix < npix) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#if C1X1_ABL_LINEAR & 2  // timing proxy: fully coalesced stores
          *((f32x4*)(p.out + pix0 * 64) + (size_t)(pb * 4 + cb) * 64 + lane) = acc[cb][pb];
#elif C1X1_NT_STORE
          __builtin_nontemporal_store(acc[cb][pb], (f32x4*)(p.out + pix * 64 + cb * 16 + 4 * kg));
#else
          *(f32x4*)(p.out + pix * 64 + cb * 16 + 4 * kg) = acc[cb][pb];
#endif
        }
      }
    }
  }
  (void)Cin;
}

extern "C" int dmd_conv1x1_stream_eligible(const dmd_conv_params* p) {
  if (!p || p->taps != 1 || p->stride != 1 || p->upsample || p->Cout != 64 || p->CoutPad != 64 || p->out_nchw) return 0;
  if (p->residual || p->residual_norm.stats || p->out_stats) return 0;
  int cin = 0;
  for (int i = 0; i < p->nsrc; ++i) {
    if (p->src[i].prologue != DMD_PROLOGUE_NONE) return 0;
    cin += p->src[i].C;
  }
  return (cin == 32 || cin == 64 || cin == 128) ? 1 : 0;
}

template <int NK, int PB, bool SPLIT>
static void launch1x1s(const dmd_conv_params& p, hipStream_t st) {
  const size_t npix = (size_t)p.N * p.H * p.W;
  const int tpix = 64 * PB;
  const int tiles = (int)((npix + tpix - 1) / tpix);
  const int nwg = tiles < C1X1_NWG ? tiles : C1X1_NWG;  // persistent: the weights are staged into LDS once per workgroup
  const int tpw = (tiles + nwg - 1) / nwg;
  hipLaunchKernelGGL((conv1x1_stream_kernel<NK, PB, SPLIT>), dim3((tiles + tpw - 1) / tpw), dim3(256), 0, st, p, tiles, tpw);
}

template <int NK, int PB>
static void launch1x1(const dmd_conv_params& p, hipStream_t st) {
  if ((p.precision & 0xff) == DMD_PRECISION_F16X2)
    launch1x1s<NK, PB, true>(p, st);
  else
    launch1x1s<NK, PB, false>(p, st);
}

int dmd_launch_conv1x1_stream(const dmd_conv_params& p, hipStream_t st) {
  int cin = 0;
  for (int i = 0; i < p.nsrc; ++i) cin += p.src[i].C;
  if (cin == 128) launch1x1<8, C1X1_PB128>(p, st);
  else if (cin == 64) launch1x1<4, 4>(p, st);
  else launch1x1<2, 4>(p, st);
  return 0;
}
