// conv_f16ws_kernel -- the stride-1 3x3 / 1x1 convolutions with 32 or 64 output channels (92 % of the denoiser's
// FLOPs) on the f16 matrix cores with SPLIT fp32 operands, wave-specialised and persistent.
//
// Arithmetic (DMD_PRECISION_F16X2).  Exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, dmd_conv.hip) runs at the fp32 vector
// rate (157 TFLOP/s chip peak); the f16 MFMA is 16x faster.  Every fp32 operand x is split as
//     x = h + l + e,   h = fp16(x),  l = fp16(x - h),   |e| <= max(2^-22 |x|, 2^-25)
// (gfx950's MFMA honours fp16 subnormals, tools/probe/mfma_f16_probe.hip, so l needs no scaling) and a product is
// evaluated as  w_h*x_h + w_h*x_l + w_l*x_h  -- three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator; the
// dropped w_l*x_l term is 2^-22 relative.  Result: fp32-class accuracy (measured 3-8e-7 of the output scale per
// conv) at an effective peak of 2.5 PFLOP/s / 3.
// Range contract: finite operands must satisfy |x| < 65520 (the fp16 range).  Nothing is clamped: an operand beyond
// the range becomes h = +-inf, l = -+inf and every output it touches is NaN -- an out-of-range activation fails
// LOUDLY instead of silently saturating, and NaN / Inf inputs stay non-finite exactly where F.conv2d's would
// (tests/test_gpu_precision.py).  Operands below 2^-25 in magnitude are flushed (absolute floor): tensors whose scale
// is far below 1 (gradients) are pre-scaled by a power of two by the caller (ac_native._EncoderFn.backward).
//
// Structure.  One 768-thread workgroup per CU is split by ROLE:
//   * waves 0-3 and 4-7 = two CONSUMER groups that take alternate tiles: the group whose tile is
//     current does nothing but LDS fragment reads + MFMAs; the other group meanwhile writes its
//     finished tile out (bias, residual, store, GroupNorm partial sums), one 32-pixel block per
//     chunk step.  A CU can only store ~10 B/clk, so a 64 KiB tile takes longer to write than a
//     chunk takes to compute: with a single consumer group that write sits on the critical path.
//   * waves 8-11 = PRODUCERS: global loads, GroupNorm/FiLM (one fma) + SiLU (v_exp/v_rcp) + h/l split, LDS
//     writes of the halo'd patch [patch pixel][4 x 16 B] = {h[0:8], h[8:16], l[0:8], l[8:16]} (slot rotated by
//     (px >> 1) -> conflict-free ds_read_b128 for every tap, tools/lds_sim.py) and the linear copy of the chunk's
//     pre-split weights; they run one chunk ahead of the consumers through a double-buffered {patch, weights}
//     LDS pair, two chunks ahead for the activation loads.
//   * the workgroup is persistent: it walks a contiguous range of tiles as ONE stream of chunks,
//     so the producers prefetch the next tile's first chunks while the consumers finish the
//     current tile -- no per-tile pipeline fill.
// One s_barrier per chunk separates "consumers read buffer j, producers fill buffer j + 1".
#include <type_traits>

#include "dmd_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


// NCB = 32-output-channel blocks of the convolution (2: Cout = 64, the U-Net; 1: Cout = 32, the reward/end model and
// the first actor-critic blocks).  The workgroup's consumer group is always 4 waves = NCB cout blocks x NPH pixel
// halves of 128 pixels, so a Cout = 32 tile is 512 pixels (two 16x16 patches / eight 8x8 patches).
// TAPS = 9 (3x3, pad 1) or 1 (1x1: the same halo'd patch geometry, only the centre window is loaded and read).
// JOINT: all 8 consumer waves work on ONE tile of twice the pixels (two 16x16 patches x 64 couts) instead of taking
// alternate tiles: two MFMA waves per SIMD cover each other's LDS round trips, and a chunk's 36 KiB of weights (the
// larger part of the LDS fill) is staged once per 512 pixels instead of once per 256.  The write-out of a finished tile
// is then no longer hidden under the other group's MFMAs.
// P8: ONE consumer group (waves 0-3, every tile, write-out right after the tile) and EIGHT producer waves (4-11): the
// measured critical path of a step is the producers' staging + their stalls at vector-memory issue, so the
// four waves that otherwise only write tiles out and wait become producers.  Measured r02o (DIAMOND_WS_P8=1): the K loop
// gets ~12 % faster, the now exposed write-out costs ~14 %: 287 vs 280 us on the 64x64 conv, 9.5k vs 9.9k frames/s.
// On the 32-cout instance (DIAMOND_WS_P8=2; twice the staging per MFMA) it is a wash: 64x64 Cin 32 152 vs 147 us,
// Cin 16 71 vs 80 us, 32x32 37 vs 39.5 us, 16x16 19 vs 22 us.
template <bool B8_, int NCB_, int TAPS_ = 9, bool JOINT_ = false, bool P8_ = false>
struct WsGeom {
  static constexpr bool P8 = P8_;
  static constexpr int NPT = P8_ ? 512 : 256;  // producer threads
  static constexpr bool B8 = B8_;
  static constexpr int NCB = NCB_;
  static constexpr int TAPS = TAPS_;
  static constexpr bool JOINT = JOINT_;
  static constexpr int COUT = 32 * NCB_;
  static constexpr int NCW = JOINT_ ? 8 : 4;  // consumer waves per tile
  static constexpr int NPH = NCW / NCB_;
  static constexpr int SUB = B8_ ? 2 * NPH : NPH / 2;
  static constexpr int TS = B8_ ? 8 : 16;
  static constexpr int PW = TS + 2;
  static constexpr int PPS = PW * PW;
  static constexpr int NPP = SUB * PPS;
  static constexpr int ITEMS = (NPP * 4 + NPT - 1) / NPT;
  static constexpr int W_UNITS = TAPS_ * 2 * 2 * COUT;      // 16-byte units of one chunk's weights
  static constexpr int WU = (W_UNITS + NPT - 1) / NPT;      // units per producer thread
  static constexpr int BUF_UNITS = NPP * 4 + W_UNITS;       // one {patch, weights} buffer, 16-byte units
  static constexpr int CIN_MAX = NCB_ == 2 ? 128 : 64;
  static constexpr int TAB_SLOTS = JOINT_ ? 3 : 4;  // tile generations whose tables can be alive at once (>= 3)
  static constexpr int TAB_FLOATS = TAB_SLOTS * SUB * CIN_MAX;
  // raw fp32 patch of one chunk as the LDS-DMA writes it: unit (it * 256 + tid), the last item only where it has pixels
  static constexpr int RAW_UNITS = (ITEMS - 1) * 256 + ((NPP * 4 - (ITEMS - 1) * 256 + 63) / 64) * 64;
  static constexpr int SMEM_BASE = 2 * BUF_UNITS * 16 + 2 * TAB_FLOATS * 4;
  static constexpr bool DOUBLE_STAGE = NCB_ == 2 && !JOINT_;          // two activation register sets only where they fit
  // source pixel offsets of a tile cached in registers (one per item) or recomputed at every chunk: recomputing frees
  // ITEMS registers (JOINT needs them) but costs ~18 VALU per item and chunk (measured r02e: 8x8 levels +25 %)
  static constexpr bool CACHE_GOFF = !JOINT_;
  // The chunk weights (36 KiB per step, L2 -> registers -> LDS) are copied by the consumer group that is NOT computing
  // the current tile (it only writes its finished tile out and otherwise waits at the barriers) instead of by the
  // producers: measured r02i, the weight loads cost the producers -- the critical path of every step, they stall at
  // ISSUE while the vector-memory queue is backed up -- 15 % of the kernel.  JOINT has no idle group.
#ifndef WS_W_DMA
#define WS_W_DMA 1  // with WS_W_BY_IDLE: the idle group moves the weights with LDS-DMA (global_load_lds, no registers)
#endif
#ifndef WS_W_BY_IDLE
#define WS_W_BY_IDLE 1  // r02j: through registers 346 vs 296 us (spills, the group arrives late at the barrier); r02p: with LDS-DMA (WS_W_DMA) +1.4 % end to end
#endif
  static constexpr bool W_BY_IDLE = WS_W_BY_IDLE && !JOINT_ && !P8_;
  // REV: tiles with an odd index inside their image walk the K chunks in DESCENDING order, so that at every tile
  // boundary of the persistent walk the last chunk of one tile is the first chunk of the next and its 36 KiB of weights
  // are already in LDS: one weight fetch in `nchunks` saved (a quarter of the weight bytes at Cin = 64, ALL but the
  // first at Cin = 16).  The kernel runs at the CU's memory-path ceiling (~17 GB/s per CU whatever the mix of
  // activations / weights / outputs), half of the bytes are weights.  The order depends only on the tile's
  // position inside its image, never on the batch: results stay bitwise independent of the launch configuration.
#ifndef WS_REV
#define WS_REV 0  // measured r02r: correct (bitwise tpw tests pass) but no gain (10.0k vs 10.2k frames/s, conv 269 vs 266 us)
#endif
  static constexpr bool REV = WS_REV && W_BY_IDLE && SUB == 1;
  // A_DMA: the ACTIVATIONS, too, are moved by the idle consumer group with LDS-DMA (raw fp32 patch of element e + 2 into a
  // two-slot LDS ring while element e is computed); the producers then never touch global memory: they read the raw patch
  // from LDS, normalise / activate / split it and write the h/l patch.  Needs 2 x RAW_UNITS x 16 more bytes of LDS.
#ifndef WS_A_DMA
#define WS_A_DMA 0  // measured r02x: correct (108 conv tests) but the idle group becomes the critical path: 294 vs 264 us, 9.1k vs 9.8k frames/s
#endif
  static constexpr bool A_DMA = WS_A_DMA && WS_W_DMA && W_BY_IDLE && DOUBLE_STAGE && SUB == 1 && !P8_ &&
                                (SMEM_BASE + 2 * RAW_UNITS * 16 <= 160 * 1024);
  static constexpr int SMEM_BYTES = SMEM_BASE + (A_DMA ? 2 * RAW_UNITS * 16 : 0);
};

struct WsTile {
  int n, y0, x0;
  bool valid;
};

template <class G>
__device__ __forceinline__ WsTile ws_subtile(const dmd_conv_params& p, int tile, int s) {
  const int tx = p.W / G::TS, per_img = tx * (p.H / G::TS);
  const int gs = tile * G::SUB + s;
  WsTile t;
  t.valid = gs < p.N * per_img;
  const int g2 = t.valid ? gs : 0;
  t.n = g2 / per_img;
  const int r = g2 - t.n * per_img;
  const int ty = r / tx;
  t.y0 = ty * G::TS;
  t.x0 = (r - ty * tx) * G::TS;
  return t;
}

__device__ __forceinline__ float ws_silu(float t) {
  return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
}

#ifndef WS_TRACE
#define WS_TRACE 0  // development: per-step s_memtime stamps of workgroup WS_TRACE_WG into a device buffer (tools/debug/ws_trace.py)
#endif
#if WS_TRACE
#define WS_TRACE_WG 37
#define WS_TRACE_N 4096
__device__ unsigned long long ws_trace_buf[3][WS_TRACE_N];
__device__ int ws_trace_cnt[3];
#define WS_STAMP(role_, tag_, step_)                                                                                  \
  do {                                                                                                                 \
    if (blockIdx.x == WS_TRACE_WG && (threadIdx.x & 255) == 0 && ws_ti < WS_TRACE_N) {                                  \
      ws_trace_buf[role_][ws_ti] = (__builtin_readcyclecounter() << 16) | ((unsigned long long)(tag_) << 12) | ((step_) & 0xfff); \
      ws_trace_cnt[role_] = ++ws_ti;                                                                                   \
    }                                                                                                                  \
  } while (0)
extern "C" int dmd_ws_trace_dump(unsigned long long* host, int* counts) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host, HIP_SYMBOL(ws_trace_buf), sizeof(unsigned long long) * 3 * WS_TRACE_N);
  hipMemcpyFromSymbol(counts, HIP_SYMBOL(ws_trace_cnt), sizeof(int) * 3);
  int zero[3] = {0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(ws_trace_cnt), zero, sizeof(zero));
  return WS_TRACE_N;
}
#else
#define WS_STAMP(role_, tag_, step_) do {} while (0)
#endif

#ifndef WS_PIPE
#define WS_PIPE 0  // 1: tap-level software pipeline of the fragment reads. Measured r02b: 5 % SLOWER than the compiler's tap-by-tap order (367 vs 350 us on the 64x64 residual conv)
#endif

// one tap's MFMA operands of a consumer wave: weight pieces (A) and the 4 pixel blocks' activation pieces (B)
struct WsFrag {
  h8 ah, al, bh[4], bl[4];
};

template <class G>
__device__ __forceinline__ void ws_load_frag(const u32x4* buf, int tt, int wunit, const int (&pixbase)[4], const int (&posh)[3],
                                             WsFrag& f) {
  const int tap = tt;                     // index into the chunk's weights
  const int win = G::TAPS == 9 ? tt : 4;  // window of the 3x3 patch geometry (4 = centre)
  const int dy = win / 3, dx = win % 3;
  const int toff = dy * G::PW + dx;
  f.ah = __builtin_bit_cast(h8, buf[(tap * 2 + 0) * 2 * G::COUT + wunit]);
#pragma unroll
  for (int b = 0; b < 4; ++b) f.bh[b] = __builtin_bit_cast(h8, buf[(pixbase[b] + toff) * 4 + posh[dx]]);
#pragma unroll
  for (int b = 0; b < 4; ++b) f.bl[b] = __builtin_bit_cast(h8, buf[(pixbase[b] + toff) * 4 + (posh[dx] ^ 2)]);
  f.al = __builtin_bit_cast(h8, buf[(tap * 2 + 1) * 2 * G::COUT + wunit]);
}

__device__ __forceinline__ void ws_mfma_frag(const WsFrag& f, f32x16 (&acc)[4]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.bh[b], acc[b], 0, 0, 0);
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.bl[b], acc[b], 0, 0, 0);
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al, f.bh[b], acc[b], 0, 0, 0);
}

// phase fence of the tap pipeline: nothing is scheduled across it, so the reads of tap t + 1 stay in the phase whose MFMAs
// are tap t's (left alone, the scheduler sinks every read to just before its first use to save registers and the wave
// stalls on each LDS round trip)
__device__ __forceinline__ void ws_sched_tap(bool) { __builtin_amdgcn_sched_barrier(0); }

template <class G>
__global__ __launch_bounds__(768, 3) void conv_f16ws_kernel(const dmd_conv_params p, int ntiles, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* bufs = (u32x4*)smem_raw;  // [2][BUF_UNITS]: patch [NPP][4] then weights [9][2][2][64]
  float* tab_a = (float*)(bufs + 2 * G::BUF_UNITS);  // [slot][SUB][CIN_MAX]
  float* tab_b = tab_a + G::TAB_FLOATS;
  u32x4* rawring = (u32x4*)(tab_b + G::TAB_FLOATS);  // [2][RAW_UNITS] (A_DMA only)
  static_assert(G::SMEM_BYTES <= 160 * 1024, "LDS budget");

#ifndef WS_ABL
#define WS_ABL 0  // development only (WRONG results): 1 = producers idle in steady state, 2 = no activation loads, 4 = no store_S, 16 = no MFMA loop,
                  // 32 = no weight global loads, 64 = no weight LDS writes either, 128 = no epilogue global stores / residual loads
#endif
  // 0, 1: consumer groups (even / odd tiles), 2: producer (staging); P8: 0 = the consumer group, 2 = producers (threads 256..767)
  const int role = G::P8 ? (threadIdx.x >> 8 ? 2 : 0) : (int)(threadIdx.x >> 8);
#if WS_TRACE
  int ws_ti = 0;  // next trace slot of this role's stamping thread (stores only: no load on the stamping path)
#endif
  const int tid = (G::P8 && threadIdx.x >= 256) ? (int)threadIdx.x - 256 : (int)(threadIdx.x & 255);  // index inside the role
  const int lane = tid & 63, wave = tid >> 6;
  const int up = p.upsample;
  const int Hs = p.H >> up, Ws = p.W >> up;
  const int C0 = p.src[0].C;
  const int C1 = p.nsrc > 1 ? p.src[1].C : 0;
  const int nch0 = C0 >> 4;
  const int nchunks = (C0 + C1) >> 4;
  const int tile0 = blockIdx.x * tiles_per_wg;
  const int nmy = min(tiles_per_wg, ntiles - tile0);
  const int S = nmy * nchunks;  // chunk stream of this workgroup
  // Workgroups walk their tile range from different starting points: with one image per workgroup all CUs would
  // otherwise touch the same (row, column) offsets of 1 MiB-strided images at the same time (same low address
  // bits -> the same HBM channels).
  const int rot = (blockIdx.x * 7) % nmy;
#define WS_TILE(k) (tile0 + (((k) + rot) >= nmy ? (k) + rot - nmy : (k) + rot))
  const int tiles_per_img = (p.H / G::TS) * (p.W / G::TS);
  // K chunk (16 input channels) processed by stream element e
  auto chunk_of = [&](int e) -> int {
    const int k = e / nchunks, ck = e - k * nchunks;
    if (!G::REV) return ck;
    return ((WS_TILE(k) % tiles_per_img) & 1) ? nchunks - 1 - ck : ck;
  };

  if (role == 2) {
    // =================================== PRODUCER ===================================
#ifdef WS_PRODUCER_PRIO
    __builtin_amdgcn_s_setprio(WS_PRODUCER_PRIO);
#endif
    const int q = tid & 3;
    int ipos[G::ITEMS];  // (sub << 16) | (py << 8) | px; -1: no item (beyond the patch)
    // 8-byte unit index of the h half-quad of item `it` in a patch (recomputed where needed: registers are scarce)
    auto loff_of = [&](int it) { return ((it * (G::NPT / 4) + (tid >> 2)) * 4 + (((q >> 1) + ((ipos[it] & 0xff) >> 1)) & 3)) * 2 + (q & 1); };
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
      const int id = it * G::NPT + tid;
      const int pp = id >> 2;
      const bool ok = pp < G::NPP;
      const int s = G::SUB == 1 ? 0 : (ok ? pp / G::PPS : 0);
      const int rem = pp - s * G::PPS;
      const int py = rem / G::PW, px = rem - py * G::PW;
      ipos[it] = ok ? ((s << 16) | (py << 8) | px) : -1;
    }
    int gk = -1;          // tile whose descriptors / tables are current
    int tab_n[G::SUB];    // images whose tables are current, and their slot
#pragma unroll
    for (int s2 = 0; s2 < G::SUB; ++s2) tab_n[s2] = -1;
    int tab_slot = -1;
    WsTile ti[G::SUB];    // sub-tiles of tile gk (wave-uniform)
    int goff[G::CACHE_GOFF ? G::ITEMS : 1];  // CACHE_GOFF: source pixel index per item for tile gk (0 outside the image)
    unsigned gzero = 0;                      // CACHE_GOFF: bit it = item is conv zero padding / outside the tensor
    const u32x4* wglob = (const u32x4*)p.w_f16;
    auto item_source = [&](int it, bool& inb) -> int {
      const int s = ipos[it] >> 16, py = (ipos[it] >> 8) & 0xff, px = ipos[it] & 0xff;
      WsTile t = ti[0];
#pragma unroll
      for (int kk = 1; kk < G::SUB; ++kk)
        if (s == kk) t = ti[kk];
      const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
      const bool window = G::TAPS == 9 || (py >= 1 && py <= G::TS && px >= 1 && px <= G::TS);  // 1x1: no halo needed
      inb = ipos[it] >= 0 && window && t.valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      return inb ? ((t.n * Hs + (iy >> up)) * Ws + (ix >> up)) : 0;
    };

    auto setup_tile = [&](int k) {  // descriptors + normalisation tables of tile k
      const int tile = WS_TILE(k);
#pragma unroll
      for (int s = 0; s < G::SUB; ++s) ti[s] = ws_subtile<G>(p, tile, s);
      if (G::CACHE_GOFF) {
        unsigned gz = 0;
#pragma unroll
        for (int it = 0; it < G::ITEMS; ++it) {
          bool inb;
          goff[G::CACHE_GOFF ? it : 0] = item_source(it, inb);
          gz |= (inb ? 0u : 1u) << it;
        }
        gzero = gz;
      }
      // tables: all tiles of one image share them -- rebuild only when an image of the tile changes
      bool rebuild = false;
#pragma unroll
      for (int s2 = 0; s2 < G::SUB; ++s2) {
        const int nn = ti[s2].valid ? ti[s2].n : -2;
        rebuild |= nn != tab_n[s2];
        tab_n[s2] = nn;
      }
      if (rebuild) tab_slot = tab_slot + 1 >= G::TAB_SLOTS ? 0 : tab_slot + 1;
      if (rebuild && tid < C0 + C1) {  // one channel per thread; visible to the other producers after the next barrier
        const int c = tid;
        const int si = c < C0 ? 0 : 1;
        const dmd_conv_src& sc = p.src[si];
        const int cl = si ? c - C0 : c;
        const int slot = tab_slot;
#pragma unroll
        for (int s = 0; s < G::SUB; ++s) {
          float m = 0.f, a = 1.f, ad = 0.f;
          if (sc.prologue != DMD_PROLOGUE_NONE && ti[s].valid)
            norm_entry(sc.norm, ti[s].n, cl, sc.C, (double)DMD_GN_GROUP * Hs * Ws, &m, &a, &ad);
          tab_a[(slot * G::SUB + s) * G::CIN_MAX + c] = a;
          tab_b[(slot * G::SUB + s) * G::CIN_MAX + c] = ad - m * a;
        }
      }
      gk = k;
    };

    // activation loads of stream element e into a register set (+ which items are conv zero padding)
    auto issue_S = [&](int e, auto& st, unsigned& zmask, int& slot_out) {
      const int k = e / nchunks, ck = chunk_of(e);
      if (k != gk) setup_tile(k);
      slot_out = tab_slot;
      const int si = ck < nch0 ? 0 : 1;
      const dmd_conv_src& sc = p.src[si];
      const int c0 = (si ? ck - nch0 : ck) * 16 + 4 * q;
      const f32x4* base4 = (const f32x4*)sc.x + (c0 >> 2);  // 16-byte units: index = pixel * (C / 4)
      const unsigned cq = (unsigned)sc.C >> 2;
      unsigned gz = gzero;
#pragma unroll
      for (int it = 0; it < G::ITEMS; ++it) {
        int go;
        if (G::CACHE_GOFF) {
          go = goff[G::CACHE_GOFF ? it : 0];
        } else {
          bool inb;
          go = item_source(it, inb);
          gz |= (inb ? 0u : 1u) << it;
        }
#if WS_ABL & 2
        st[it] = (f32x4){(float)go, 1.f, 2.f, (float)c0};
#elif WS_ABL & 512
        st[it] = base4[((unsigned)go & 0x3fffu) * cq];  // timing proxy: same loads folded into a 4 MB (L2-resident) window
#elif WS_ABL & 1024
        st[it] = base4[((unsigned)go & 0xfffffu) * cq];  // ... into a 256 MB window (Infinity-Cache-sized)
#else
        st[it] = base4[(unsigned)go * cq];
#endif
      }
      zmask = gz;
    };
    // normalise / activate / split element e from its register set into patch buffer e & 1
    // mode (wave-uniform, hoisted out of the item loop as a compile-time tag): 0 = no prologue, 1 = norm, 2 = norm + SiLU
    auto store_S_mode = [&](auto mode_tag, int e, const auto& st, unsigned zmask, int slot) {
      constexpr int MODE = decltype(mode_tag)::value;
      const int ck = chunk_of(e);
      const int cc = ck * 16 + 4 * q;
      uint2* pb = (uint2*)(bufs + (e & 1) * G::BUF_UNITS);
      // single-image tiles: the (a, b) rows of this thread's channel quad are the same for every item -> ONE pair of
      // LDS reads per chunk instead of one (with its lgkmcnt stall) per item
      f32x4 ta0 = (f32x4){1.f, 1.f, 1.f, 1.f}, tb0 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (MODE != 0 && G::SUB == 1) {
        ta0 = *(const f32x4*)(tab_a + slot * G::CIN_MAX + cc);
        tb0 = *(const f32x4*)(tab_b + slot * G::CIN_MAX + cc);
      }
#pragma unroll
      for (int it = 0; it < G::ITEMS; ++it) {
        f32x4 v = st[it];
        if (MODE != 0) {
          f32x4 ta = ta0, tb = tb0;
          if (G::SUB > 1) {
            const int s = ipos[it] >> 16;
            ta = *(const f32x4*)(tab_a + (slot * G::SUB + s) * G::CIN_MAX + cc);
            tb = *(const f32x4*)(tab_b + (slot * G::SUB + s) * G::CIN_MAX + cc);
          }
#pragma unroll
          for (int el = 0; el < 4; ++el) {
            float t = __builtin_fmaf(v[el], ta[el], tb[el]);
            if (MODE == 2) t = ws_silu(t);
            v[el] = t;
          }
        }
        h4 hv, lv;
        const bool zero = (zmask >> it) & 1;  // conv zero padding is applied AFTER the activation (blocks.py:143-144)
#pragma unroll
        for (int el = 0; el < 4; ++el) {
          const float x = zero ? 0.f : v[el];  // no clamp: out-of-range operands turn into NaN outputs (header)
          const _Float16 h = (_Float16)x;
          hv[el] = h;
          lv[el] = (_Float16)(x - (float)h);
        }
        if ((it + 1) * G::NPT <= G::NPP * 4 || ipos[it] >= 0) {  // only the last item row can fall beyond the patch
          const int lo = loff_of(it);
          pb[lo] = __builtin_bit_cast(uint2, hv);
          pb[lo ^ 4] = __builtin_bit_cast(uint2, lv);
        }
      }
    };
    auto store_S = [&](int e, const auto& st, unsigned zmask, int slot) {
#if WS_ABL & 4
      if (e > 1) return;
#endif
      const int ck = chunk_of(e);
      const int prologue = p.src[ck < nch0 ? 0 : 1].prologue;
      if (prologue == DMD_PROLOGUE_NORM_SILU)
        store_S_mode(std::integral_constant<int, 2>{}, e, st, zmask, slot);
      else if (prologue == DMD_PROLOGUE_NORM)
        store_S_mode(std::integral_constant<int, 1>{}, e, st, zmask, slot);
      else
        store_S_mode(std::integral_constant<int, 0>{}, e, st, zmask, slot);
    };

    f32x4 stage0[G::ITEMS], stage1[G::DOUBLE_STAGE ? G::ITEMS : 1];
    unsigned zm0 = 0, zm1 = 0;
    int sl0 = 0, sl1 = 0;
#ifndef WS_WPF
#define WS_WPF 0  // 1: weights fetched TWO steps ahead through two register sets (double-stage geometries only)
#endif
    constexpr bool WPF2 = WS_WPF && G::DOUBLE_STAGE;
    u32x4 wst[G::WU], wst2[WPF2 ? G::WU : 1];
    auto load_Wr = [&](int e, auto& ws) {
      const int ck = chunk_of(e);
      const u32x4* w = wglob + (size_t)ck * G::W_UNITS + tid;
#pragma unroll
      for (int i = 0; i < G::WU; ++i)
        if (G::W_UNITS % G::NPT == 0 || tid + G::NPT * i < G::W_UNITS) {
#if WS_ABL & 32
          ws[i] = (u32x4){(unsigned)(size_t)w, 0x3c003c00u, 0u, (unsigned)i};
#else
          ws[i] = w[G::NPT * i];
#endif
        }
    };
    auto store_Wr = [&](int e, const auto& ws) {
#if WS_ABL & 64
      if (e > 1) return;
#endif
      u32x4* wl = bufs + (e & 1) * G::BUF_UNITS + G::NPP * 4;
#pragma unroll
      for (int i = 0; i < G::WU; ++i)
        if (G::W_UNITS % G::NPT == 0 || tid + G::NPT * i < G::W_UNITS) wl[tid + G::NPT * i] = ws[i];
    };
    auto load_W = [&](int e) {
      if (!G::W_BY_IDLE) load_Wr(e, wst);
    };
    auto store_W = [&](int e) {
      if (!G::W_BY_IDLE) store_Wr(e, wst);
    };

#ifndef WS_TRIPLE
#define WS_TRIPLE 0  // 1: THREE activation register sets (loads three steps ahead) where the weights are moved by the idle group
#endif
    constexpr bool TRIPLE = WS_TRIPLE && G::DOUBLE_STAGE && G::W_BY_IDLE;
    f32x4 stage2[TRIPLE ? G::ITEMS : 1];
    unsigned zm2 = 0;
    int sl2 = 0;
    if (G::A_DMA) {
      // the raw patch of element e was DMA'd into rawring slot e & 1 by the idle consumer group (two steps before it is
      // consumed); here: LDS -> registers -> prologue math -> h/l patch of buffer e & 1
      auto stage_from_lds = [&](int e) {  // requires: descriptors / tables of element e's tile are current AND visible
        const u32x4* rs = rawring + (e & 1) * G::RAW_UNITS;
        f32x4 st[G::ITEMS];
#pragma unroll
        for (int it = 0; it < G::ITEMS; ++it)
          st[it] = (it * 256 + tid < G::RAW_UNITS) ? __builtin_bit_cast(f32x4, rs[it * 256 + tid]) : (f32x4){0.f, 0.f, 0.f, 0.f};
        store_S(e, st, gzero, tab_slot);
        // the next element's tile: its tables go to a fresh slot and become visible at the barrier that ends this step
        if (e + 1 < S && (e + 1) / nchunks != gk) setup_tile((e + 1) / nchunks);
      };
      __syncthreads();  // B(-1): the raw patches of elements 0 and 1 have landed; the tables of tile 0 need one more barrier
      setup_tile(0);
      __syncthreads();  // B(-1'): tables visible
      stage_from_lds(0);
      __syncthreads();  // B0
      for (int j = 0; j < S; ++j) {
        if (j + 1 < S) stage_from_lds(j + 1);
        __syncthreads();
      }
    } else if (TRIPLE) {
      // element e lives in register set e % 3; step j stores element j + 1 and re-issues its set for element j + 4
      issue_S(0, stage0, zm0, sl0);
      if (S > 1) issue_S(1, stage1, zm1, sl1);
      if (S > 2) issue_S(2, stage2, zm2, sl2);
      __syncthreads();  // B(-1)
      store_S(0, stage0, zm0, sl0);
      if (S > 3) issue_S(3, stage0, zm0, sl0);
      __syncthreads();  // B0
      for (int j = 0; j < S; j += 3) {
        if (j + 1 < S) {
          store_S(j + 1, stage1, zm1, sl1);
          if (j + 4 < S) issue_S(j + 4, stage1, zm1, sl1);
        }
        __syncthreads();
        if (j + 1 < S) {
          if (j + 2 < S) {
            store_S(j + 2, stage2, zm2, sl2);
            if (j + 5 < S) issue_S(j + 5, stage2, zm2, sl2);
          }
          __syncthreads();
          if (j + 2 < S) {
            if (j + 3 < S) {
              store_S(j + 3, stage0, zm0, sl0);
              if (j + 6 < S) issue_S(j + 6, stage0, zm0, sl0);
            }
            __syncthreads();
          }
        }
      }
    } else if (G::DOUBLE_STAGE && WPF2) {
      // as below, with the weights of element e fetched two steps before they are copied into LDS (even elements
      // through wst, odd ones through wst2): an L2 round trip under load is longer than one step's staging work
      issue_S(0, stage0, zm0, sl0);
      load_Wr(0, wst);
      if (S > 1) issue_S(1, stage1, zm1, sl1);
      if (S > 1) load_Wr(1, wst2);
      __syncthreads();  // B(-1)
      store_S(0, stage0, zm0, sl0);
      if (S > 2) issue_S(2, stage0, zm0, sl0);
      store_Wr(0, wst);
      if (S > 2) load_Wr(2, wst);
      __syncthreads();  // B0
      for (int j = 0; j < S; j += 2) {
        if (j + 1 < S) {
          store_S(j + 1, stage1, zm1, sl1);
          if (j + 3 < S) issue_S(j + 3, stage1, zm1, sl1);
          store_Wr(j + 1, wst2);
          if (j + 3 < S) load_Wr(j + 3, wst2);
        }
        __syncthreads();
        if (j + 1 < S) {
          if (j + 2 < S) {
            store_S(j + 2, stage0, zm0, sl0);
            if (j + 4 < S) issue_S(j + 4, stage0, zm0, sl0);
            store_Wr(j + 2, wst);
            if (j + 4 < S) load_Wr(j + 4, wst);
          }
          __syncthreads();
        }
      }
    } else if (G::DOUBLE_STAGE) {
      // fill element 0; elements 1 and 2 in flight
      issue_S(0, stage0, zm0, sl0);
      load_W(0);
      if (S > 1) issue_S(1, stage1, zm1, sl1);
      __syncthreads();  // B(-1): the tables written by setup_tile are visible to all producers
      store_S(0, stage0, zm0, sl0);
      if (S > 2) issue_S(2, stage0, zm0, sl0);
      store_W(0);
      __syncthreads();  // B0: buffer 0 = element 0
      // step j: consumers compute element j, producers fill element j + 1 (register set (j + 1) & 1)
      for (int j = 0; j < S; j += 2) {
#if WS_ABL & 1
        __syncthreads();
        if (j + 1 < S) __syncthreads();
        continue;
#endif
        WS_STAMP(2, 0, j);
        if (j + 1 < S) {
          load_W(j + 1);
          WS_STAMP(2, 1, j);
          store_S(j + 1, stage1, zm1, sl1);
          WS_STAMP(2, 2, j);
          if (j + 3 < S) issue_S(j + 3, stage1, zm1, sl1);
          WS_STAMP(2, 3, j);
          store_W(j + 1);
          WS_STAMP(2, 4, j);
        }
        __syncthreads();
        WS_STAMP(2, 5, j);
        if (j + 1 < S) {
          if (j + 2 < S) {
            load_W(j + 2);
            WS_STAMP(2, 1, j + 1);
            store_S(j + 2, stage0, zm0, sl0);
            WS_STAMP(2, 2, j + 1);
            if (j + 4 < S) issue_S(j + 4, stage0, zm0, sl0);
            WS_STAMP(2, 3, j + 1);
            store_W(j + 2);
            WS_STAMP(2, 4, j + 1);
          }
          __syncthreads();
          WS_STAMP(2, 5, j + 1);
        }
      }
    } else {
      // one register set: activations are fetched one step ahead only
      issue_S(0, stage0, zm0, sl0);
      load_W(0);
      __syncthreads();  // B(-1)
      store_S(0, stage0, zm0, sl0);
      if (S > 1) issue_S(1, stage0, zm0, sl0);
      store_W(0);
      __syncthreads();  // B0
      for (int j = 0; j < S; ++j) {
        if (j + 1 < S) {
          load_W(j + 1);
          store_S(j + 1, stage0, zm0, sl0);
          if (j + 2 < S) issue_S(j + 2, stage0, zm0, sl0);
          store_W(j + 1);
        }
        __syncthreads();
      }
    }
  } else {
    // =================================== CONSUMER ===================================
#ifndef WS_CONSUMER_PRIO
#define WS_CONSUMER_PRIO 3
#endif
    // static priority: the MFMA waves win issue arbitration against the co-resident staging wave of their SIMD
    __builtin_amdgcn_s_setprio(WS_CONSUMER_PRIO);
    const int cwave = G::JOINT ? (int)(threadIdx.x >> 6) : wave;  // JOINT: both consumer groups share the tile
    const int cb = cwave % G::NCB;  // 32-cout block == GroupNorm group
    const int ph = cwave / G::NCB;  // 128-pixel part of the tile
    const int n31 = lane & 31, g = lane >> 5;
    int pixbase[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      if (G::B8) {
        const int s = ph * 2 + (blk >> 1);
        const int row = (blk & 1) * 4 + (n31 >> 3);
        pixbase[blk] = s * G::PPS + row * G::PW + (n31 & 7);
      } else {
        const int row = (ph & 1) * 8 + blk * 2 + (n31 >> 4);
        pixbase[blk] = (ph >> 1) * G::PPS + row * G::PW + (n31 & 15);
      }
    }
    const int col = G::B8 ? (n31 & 7) : (n31 & 15);
    int posh[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) posh[dx] = (g + ((col + dx) >> 1)) & 3;
    const int wunit = G::NPP * 4 + g * G::COUT + cb * 32 + n31;  // weights follow the patch inside a buffer

    f32x16 acc[4];

    // ---- write-out state of this group's finished tile ----
    // lane owns couts cb*32 + 8 qd + 4 g + (0..3), qd = 0..3, of pixel n31 of each 32-pixel block
    int pixoff[4];     // 16-byte unit offset of (pixel, cb*32 + 4 g) per block, -1: sub-tile outside the tensor
    int stat_slot[2];  // out_stats slot per statistics tile of this wave, -1: none
    double ssum[2], ssq[2];
    int pending = 0;   // next block of the finished tile to write (4 = nothing pending)
    pending = 4;
    auto epi_begin = [&](int k) {
      const int tile = WS_TILE(k);
      WsTile ti[G::SUB];
#pragma unroll
      for (int s = 0; s < G::SUB; ++s) ti[s] = ws_subtile<G>(p, tile, s);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        WsTile t = ti[0];
        int oy, ox;
        if (G::B8) {
          const int s = ph * 2 + (blk >> 1);
#pragma unroll
          for (int kk = 1; kk < G::SUB; ++kk)
            if (s == kk) t = ti[kk];
          oy = t.y0 + (blk & 1) * 4 + (n31 >> 3);
          ox = t.x0 + (n31 & 7);
        } else {
#pragma unroll
          for (int kk = 1; kk < G::SUB; ++kk)
            if ((ph >> 1) == kk) t = ti[kk];
          oy = t.y0 + (ph & 1) * 8 + blk * 2 + (n31 >> 4);
          ox = t.x0 + (n31 & 15);
        }
        pixoff[blk] = t.valid ? (((t.n * p.H + oy) * p.W + ox) * (G::COUT / 4) + cb * 8 + g) : -1;  // 16-byte units
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        WsTile t = ti[0];
        int T, tt;
        if (G::B8) {
          const int s = ph * 2 + kk;
#pragma unroll
          for (int k2 = 1; k2 < G::SUB; ++k2)
            if (s == k2) t = ti[k2];
          const int tx8 = p.W / 8;
          T = tx8 * (p.H / 8);
          tt = (t.y0 / 8) * tx8 + t.x0 / 8;
        } else {
#pragma unroll
          for (int k2 = 1; k2 < G::SUB; ++k2)
            if ((ph >> 1) == k2) t = ti[k2];
          const int tx16 = p.W / 16;
          T = tx16 * (p.H / 8);
          tt = (t.y0 / 8 + (ph & 1)) * tx16 + t.x0 / 16;
        }
        stat_slot[kk] = (t.valid && (G::B8 || kk == 0)) ? ((t.n * G::NCB + cb) * T + tt) : -1;
        ssum[kk] = 0.0;
        ssq[kk] = 0.0;
      }
      pending = 0;
    };
    // blocks [pending, pending + count) of the finished tile: bias, residual, store, statistics
    auto epi_blocks = [&](int count) {
      const int first = pending, last = min(4, pending + count);
      f32x4 bias[4];  // re-read per call (L1/L2 resident): not worth 16 registers across the MFMA loop
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bias[qd] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias[qd] = *(const f32x4*)(p.bias + cb * 32 + 8 * qd + 4 * g);
      }
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        if (blk < first || blk >= last) continue;  // uniform; keeps acc[] statically indexed
        if (G::NCB == 1 && p.out_nchw) {
          // few-channel NCHW output (conv_out: 64 -> 3, weights zero-padded to 32 couts): the real channels are
          // couts 0..3 = quad 0 of the k-group-0 lanes; consecutive lanes = consecutive pixels of a plane
          if (pixoff[blk] >= 0 && g == 0) {
            const int pixel = pixoff[blk] >> 3;  // cb == 0, g == 0
            const int HW = p.H * p.W;
            const int n = pixel / HW, rem = pixel - n * HW;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < p.Cout) p.out[((size_t)n * p.Cout + c) * HW + rem] = acc[blk][c] + bias[0][c];
          }
          continue;
        }
        if (pixoff[blk] >= 0) {
          float* op = p.out + (size_t)pixoff[blk] * 4;
#if WS_ABL & 256
          // timing proxy (WRONG results): the same bytes as fully coalesced 1 KiB-per-instruction accesses
          const size_t cbase = (size_t)__shfl(pixoff[blk], 0, 64) * 4 + (size_t)lane * 4;
#endif
          f32x4 rv[4];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
#if WS_ABL & 256
            rv[qd] = p.residual ? *(const f32x4*)(p.residual + cbase + 256 * qd) : (f32x4){0.f, 0.f, 0.f, 0.f};
#else
            rv[qd] = (p.residual && !(WS_ABL & 128)) ? *(const f32x4*)(p.residual + (size_t)pixoff[blk] * 4 + 8 * qd) : (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
          }
          float fs = 0.f, fq = 0.f;  // fp32 over the lane's 16 values of this block, fp64 across
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 v = (f32x4){acc[blk][4 * qd], acc[blk][4 * qd + 1], acc[blk][4 * qd + 2], acc[blk][4 * qd + 3]};
            v += bias[qd];
            v += rv[qd];
#if WS_ABL & 256
            *(f32x4*)(p.out + cbase + 256 * qd) = v;
#else
            if (!(WS_ABL & 128) || v[0] == 1.2345e30f) *(f32x4*)(op + 8 * qd) = v;
#endif
            fs += (v[0] + v[1]) + (v[2] + v[3]);
            // Sum of squares as an fma chain into its own register, NOT as in-place squares of v: with the squares
            // written over v's registers (`v_mul_f32 v48, v48, v48` right behind the `global_store_dwordx4 v[48:51]`),
            // a slice executed while the other consumer group's MFMAs run on the same SIMD lost one lane's
            // contribution of a block now and then (sum and stored outputs exact, sum of squares short by ~16 values)
            // -- found by tests/test_gpu_tpw.py (tiles_per_wg >= 2), reproduced and bisected in tools/debug/r02*.
            fq = __builtin_fmaf(v[0], v[0], fq);
            fq = __builtin_fmaf(v[1], v[1], fq);
            fq = __builtin_fmaf(v[2], v[2], fq);
            fq = __builtin_fmaf(v[3], v[3], fq);
          }
          const int slot = G::B8 ? (blk >> 1) : 0;
          ssum[slot] += (double)fs;
          ssq[slot] += (double)fq;
        }
      }
      pending = last;
      if (last == 4 && first < 4 && p.out_stats) {
#pragma unroll
        for (int kk = 0; kk < (G::B8 ? 2 : 1); ++kk) {
          const double a = dmd_wave_sum(ssum[kk]);
          const double b = dmd_wave_sum(ssq[kk]);
          if (lane == 0 && stat_slot[kk] >= 0) {
            double* o = p.out_stats + (size_t)stat_slot[kk] * 2;
            o[0] = a;
            o[1] = b;
          }
        }
      }
    };
    const int blocks_per_step = nchunks >= 4 ? 1 : (nchunks >= 2 ? 2 : 4);

    // ---- weight copy of the idle group (W_BY_IDLE): element e's 36 KiB into buffer e & 1 ----
    u32x4 cw[(G::W_BY_IDLE && !WS_W_DMA) ? G::WU : 1];
    const u32x4* cwglob = (const u32x4*)p.w_f16;
    auto cons_load_W = [&](int ck, int wbuf) {
#if WS_ABL & 32
      if (wbuf >= 0) return;  // ablation: the weights are never moved (LDS keeps whatever it held)
#endif
      const u32x4* w = cwglob + (size_t)ck * G::W_UNITS + tid;
#if WS_W_DMA
      // LDS-DMA: lane l of a wave lands at (wave-uniform base) + 16 l; round i moves units [256 i, 256 i + 256)
      u32x4* wl = (u32x4*)bufs + wbuf * G::BUF_UNITS + G::NPP * 4 + wave * 64;
#pragma unroll
      for (int i = 0; i < G::WU; ++i)
        if (G::W_UNITS % 256 == 0 || 256 * i + wave * 64 < G::W_UNITS)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + 256 * i),
                                           (__attribute__((address_space(3))) void*)(wl + 256 * i), 16, 0, 0);
#else
#pragma unroll
      for (int i = 0; i < G::WU; ++i)
        if (G::W_UNITS % 256 == 0 || tid + 256 * i < G::W_UNITS) cw[(G::W_BY_IDLE && !WS_W_DMA) ? i : 0] = w[256 * i];
#endif
    };
    auto cons_store_W = [&](int wbuf) {
#if WS_W_DMA
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA writes have landed before the step's barrier
      (void)wbuf;
#else
      u32x4* wl = (u32x4*)bufs + wbuf * G::BUF_UNITS + G::NPP * 4;
#pragma unroll
      for (int i = 0; i < G::WU; ++i)
        if (G::W_UNITS % 256 == 0 || tid + 256 * i < G::W_UNITS) wl[tid + 256 * i] = cw[(G::W_BY_IDLE && !WS_W_DMA) ? i : 0];
#endif
    };

    // ---- A_DMA: raw activation patch of stream element E into rawring slot E & 1 (same (item, lane) -> patch position
    //      mapping as the producers' `ipos`; conv zero padding is applied by the producers) ----
    auto cons_dma_acts = [&](int E) {
      const int k = E / nchunks, ck = chunk_of(E);
      const WsTile t = ws_subtile<G>(p, WS_TILE(k), 0);
      const int si = ck < nch0 ? 0 : 1;
      const dmd_conv_src& sc = p.src[si];
      const int c0 = (si ? ck - nch0 : ck) * 16 + 4 * (tid & 3);
      u32x4* rs = rawring + (E & 1) * G::RAW_UNITS + wave * 64;
#pragma unroll
      for (int it = 0; it < G::ITEMS; ++it) {
        if (it * 256 + wave * 64 >= G::RAW_UNITS) continue;  // wave-uniform: the last item has pixels in wave 0 only
        const int pp = it * 64 + (tid >> 2);
        const int py = pp / G::PW, px = pp - py * G::PW;
        const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
        const bool window = G::TAPS == 9 || (py >= 1 && py <= G::TS && px >= 1 && px <= G::TS);
        const bool inb = pp < G::NPP && window && t.valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const int go = inb ? ((t.n * Hs + (iy >> up)) * Ws + (ix >> up)) : 0;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc.x + (size_t)(unsigned)go * sc.C + c0),
                                         (__attribute__((address_space(3))) void*)(rs + it * 256), 16, 0, 0);
      }
    };

    if (G::A_DMA) {
      if (role == 1) {  // group 1 is idle during tile 0: raw patches of elements 0 and 1 + the first chunk's weights
        cons_dma_acts(0);
        if (S > 1) cons_dma_acts(1);
        cons_load_W(chunk_of(0), 0);
        cons_store_W(0);
      }
      __syncthreads();  // B(-1)
      __syncthreads();  // B(-1')
    } else {
      __syncthreads();  // B(-1)
      if (G::W_BY_IDLE && role == 1) {  // group 1 is idle during tile 0: it provides the first chunk's weights
        cons_load_W(chunk_of(0), 0);
        cons_store_W(0);
      }
    }
    __syncthreads();  // B0
    // weight buffer of the current step (== j & 1 unless a tile boundary re-used the previous step's weights, REV)
    int wb = 0, cur_ck = chunk_of(0);
    // step j -> j + 1: which chunk comes next, can its weights stay where they are, and where do they live
    auto next_weights = [&](int jn, int& ck_n, bool& reuse, int& wb_n) {
      ck_n = chunk_of(jn);
      reuse = G::REV && ck_n == cur_ck;
      wb_n = reuse ? wb : (wb ^ 1);
    };
    int j = 0;
    for (int k = 0; k < nmy; ++k) {
      if (G::JOINT || G::P8 || (k & 1) == role) {
        // ---- this group's tile: MFMA only ----
        if (pending < 4) epi_blocks(4);  // (only if the other group's tile had too few steps to finish the write-out)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
        for (int ck = 0; ck < nchunks; ++ck, ++j) {
          const u32x4* buf = bufs + (j & 1) * G::BUF_UNITS;   // patch of this step
          const u32x4* bufw = bufs + wb * G::BUF_UNITS;        // weights of this step
          WS_STAMP(role, 0, j);
#if WS_PIPE
          // Software pipeline over the taps: the 10 fragment reads of tap t + 1 are issued between the 12 MFMAs of tap t
          // (two fragment sets), so an LDS round trip (100+ cycles under the producers' write bursts) is covered by a
          // whole tap of matrix work (384 cycles) instead of stalling the wave before every MFMA group.  Per
          // accumulator the products are still added in the order ah*bh, ah*bl, al*bh, tap by tap: results are
          // bit-identical to the unpipelined loop.
          constexpr int NT = (WS_ABL & 16) ? 0 : G::TAPS;
          WsFrag fa, fb;
          static_assert(!G::REV, "WS_PIPE reads weights and patch from one buffer: build with -DWS_REV=0");
          if (NT > 0) ws_load_frag<G>(buf, 0, wunit, pixbase, posh, fa);
#pragma unroll
          for (int tt = 0; tt < NT; tt += 2) {
            if (tt + 1 < NT) ws_load_frag<G>(buf, tt + 1, wunit, pixbase, posh, fb);
            ws_mfma_frag(fa, acc);
            ws_sched_tap(tt + 1 < NT);
            if (tt + 1 < NT) {
              if (tt + 2 < NT) ws_load_frag<G>(buf, tt + 2, wunit, pixbase, posh, fa);
              ws_mfma_frag(fb, acc);
              ws_sched_tap(tt + 2 < NT);
            }
          }
#else
#pragma unroll
          for (int tt = 0; tt < ((WS_ABL & 16) ? 0 : G::TAPS); ++tt) {
            const int tap = tt;                       // index into the chunk's weights
            const int win = G::TAPS == 9 ? tt : 4;    // window of the 3x3 patch geometry (4 = centre)
            const int dy = win / 3, dx = win % 3;
            h8 bh[4], bl[4];
            const int toff = dy * G::PW + dx;
            const h8 ah = __builtin_bit_cast(h8, bufw[(tap * 2 + 0) * 2 * G::COUT + wunit]);
            bh[0] = __builtin_bit_cast(h8, buf[(pixbase[0] + toff) * 4 + posh[dx]]);
            bh[1] = __builtin_bit_cast(h8, buf[(pixbase[1] + toff) * 4 + posh[dx]]);
            bl[0] = __builtin_bit_cast(h8, buf[(pixbase[0] + toff) * 4 + (posh[dx] ^ 2)]);
            bl[1] = __builtin_bit_cast(h8, buf[(pixbase[1] + toff) * 4 + (posh[dx] ^ 2)]);
            const h8 al = __builtin_bit_cast(h8, bufw[(tap * 2 + 1) * 2 * G::COUT + wunit]);
            bh[2] = __builtin_bit_cast(h8, buf[(pixbase[2] + toff) * 4 + posh[dx]]);
            bh[3] = __builtin_bit_cast(h8, buf[(pixbase[3] + toff) * 4 + posh[dx]]);
            bl[2] = __builtin_bit_cast(h8, buf[(pixbase[2] + toff) * 4 + (posh[dx] ^ 2)]);
            bl[3] = __builtin_bit_cast(h8, buf[(pixbase[3] + toff) * 4 + (posh[dx] ^ 2)]);
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
          }
#endif
          WS_STAMP(role, 1, j);
          __syncthreads();  // B(j + 1)
          WS_STAMP(role, 2, j);
          if (j + 1 < S) {
            int ck_n, wb_n;
            bool reuse;
            next_weights(j + 1, ck_n, reuse, wb_n);
            cur_ck = ck_n;
            wb = wb_n;
          }
        }
        epi_begin(k);  // written out while the other group computes the next tile
        if (G::JOINT || G::P8) epi_blocks(4);  // ... or right away: every consumer wave is needed for the next tile
      } else {
        // ---- the other group's tile: write our finished tile out, a slice per chunk step ----
        for (int ck = 0; ck < nchunks; ++ck, ++j) {
          WS_STAMP(role, 8, j);
          int ck_n = 0, wb_n = wb ^ 1;
          bool reuse = false;
          if (j + 1 < S) next_weights(j + 1, ck_n, reuse, wb_n);
          const bool wnext = G::W_BY_IDLE && j + 1 < S && !reuse;  // this (idle) group copies the next step's weights
          if (G::A_DMA && j + 2 < S) cons_dma_acts(j + 2);          // ... and the raw activations two steps ahead
          if (wnext) cons_load_W(ck_n, wb_n);
          if (pending < 4) epi_blocks(blocks_per_step);
          if (wnext || (G::A_DMA && j + 2 < S)) cons_store_W(wb_n);  // vmcnt(0): the DMA writes have landed
          WS_STAMP(role, 9, j);
          __syncthreads();  // B(j + 1)
          WS_STAMP(role, 10, j);
          if (j + 1 < S) {
            cur_ck = ck_n;
            wb = wb_n;
          }
        }
      }
    }
    if (pending < 4) epi_blocks(4);  // tail: the last tile(s) of the range
  }
}

template <class G>
static int launch_f16ws(const dmd_conv_params& p, int ntiles, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16ws_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::SMEM_BYTES);
    DMD_CHECK_ARG(e == hipSuccess, "conv_f16ws: hipFuncSetAttribute(%d bytes): %s", G::SMEM_BYTES, hipGetErrorString(e));
    attr_set = true;
  }
  // persistent: one 768-thread workgroup per CU (LDS-limited), contiguous tile ranges (neighbouring tiles share
  // halo rows and, inside one image, the normalisation statistics)
  static int ncu = 0;  // compute units of the current device (MI355X: 256), queried once
  if (ncu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu = n;
  }
  const int tpw = (ntiles + ncu - 1) / ncu;
  const int nwg = (ntiles + tpw - 1) / tpw;
  hipLaunchKernelGGL((conv_f16ws_kernel<G>), dim3(nwg), dim3(768), G::SMEM_BYTES, st, p, ntiles, tpw);
  return 0;
}

int dmd_launch_conv_f16ws(const dmd_conv_params& p, hipStream_t st) {
  const bool b8 = p.W % 16 != 0;
  const int sub8 = p.N * (p.H / 8) * (p.W / 8), t16 = p.N * (p.H / 16) * (p.W / 16);
  static const int joint = getenv("DIAMOND_WS_JOINT") ? atoi(getenv("DIAMOND_WS_JOINT")) : 0;
  static const int p8 = getenv("DIAMOND_WS_P8") ? atoi(getenv("DIAMOND_WS_P8")) : 0;
  if ((p8 & 1) && p.taps == 9 && p.CoutPad == 64 && !b8) return launch_f16ws<WsGeom<false, 2, 9, false, true>>(p, t16, st);
  if ((p8 & 2) && p.taps == 9 && p.CoutPad == 32 && !b8) return launch_f16ws<WsGeom<false, 1, 9, false, true>>(p, (t16 + 1) / 2, st);
  if (joint && p.taps == 9 && p.CoutPad == 64 && !b8 && t16 >= 512) return launch_f16ws<WsGeom<false, 2, 9, true>>(p, (t16 + 1) / 2, st);
  if (p.taps == 9) {
    if (p.CoutPad == 64) return b8 ? launch_f16ws<WsGeom<true, 2, 9>>(p, (sub8 + 3) / 4, st) : launch_f16ws<WsGeom<false, 2, 9>>(p, t16, st);
    return b8 ? launch_f16ws<WsGeom<true, 1, 9>>(p, (sub8 + 7) / 8, st) : launch_f16ws<WsGeom<false, 1, 9>>(p, (t16 + 1) / 2, st);
  }
  if (p.CoutPad == 64) return b8 ? launch_f16ws<WsGeom<true, 2, 1>>(p, (sub8 + 3) / 4, st) : launch_f16ws<WsGeom<false, 2, 1>>(p, t16, st);
  return b8 ? launch_f16ws<WsGeom<true, 1, 1>>(p, (sub8 + 7) / 8, st) : launch_f16ws<WsGeom<false, 1, 1>>(p, (t16 + 1) / 2, st);
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

// 1: the parameters run on conv_f16ws_kernel
extern "C" int dmd_conv2d_f16x2_eligible(const dmd_conv_params* p) {
  if (!p || (p->precision & 0xff) != DMD_PRECISION_F16X2 || !p->w_f16) return 0;
  if (p->stride != 1 || p->residual_norm.stats || (p->taps != 9 && p->taps != 1)) return 0;
  if (p->taps == 1 && p->upsample) return 0;
  // few-channel NCHW head (conv_out): Cout <= 4 zero-padded to 32, no residual / statistics
  const bool head = p->out_nchw && p->Cout <= 4 && p->CoutPad == 32 && !p->residual && !p->out_stats;
  if (!head && ((p->Cout != 64 && p->Cout != 32) || p->CoutPad != p->Cout || p->out_nchw)) return 0;
  int cin = 0;
  for (int i = 0; i < p->nsrc; ++i) cin += p->src[i].C;
  if (cin > (p->CoutPad == 64 ? 128 : 64)) return 0;
  const bool a16 = p->H % 16 == 0 && p->W % 16 == 0;
  const bool b8 = p->W % 16 != 0;
  return (a16 || b8) ? 1 : 0;
}

