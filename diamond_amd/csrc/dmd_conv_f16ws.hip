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
//     chunk step, and moves the next step's 36 KiB of pre-split weights L2 -> LDS with LDS-DMA (no registers).
//     A CU can only store ~10 B/clk, so a 64 KiB tile takes longer to write than a chunk takes to compute: with
//     a single consumer group that write sits on the critical path.
//   * waves 8-11 = PRODUCERS: global loads, GroupNorm/FiLM (one fma) + SiLU (v_exp/v_rcp) + h/l split, LDS
//     writes of the halo'd patch [patch pixel][4 x 16 B] = {h[0:8], h[8:16], l[0:8], l[8:16]} (slot rotated by
//     (px >> 1) -> conflict-free ds_read_b128 for every tap, tools/lds_sim.py); they run one chunk ahead of the
//     consumers through a double-buffered {patch, weights} LDS pair, two chunks ahead for the activation loads.
//   * the workgroup is persistent: it walks a contiguous range of tiles as ONE stream of chunks,
//     so the producers prefetch the next tile's first chunks while the consumers finish the
//     current tile -- no per-tile pipeline fill.
// One s_barrier per chunk separates "consumers read buffer j, producers fill buffer j + 1".
//
// Round 3 (what the ISA of the round-2 kernel showed):
//   * PRODUCERS.  hipcc's s_waitcnt insertion lost track of the two register sets across the loop's branches and
//     emitted `s_waitcnt vmcnt(0)` both in front of the re-issue of a set and inside the staging of the other one: every
//     step waited for the loads issued ONE step earlier, i.e. the "two chunks ahead" prefetch was one chunk deep and the
//     load latency sat on the critical path of every step ("issueS" in profiles/r02_ws_timeline_trace.txt).  The
//     activation loads are now inline-asm `global_load_dwordx4` the compiler does not count, with hand-counted
//     `s_waitcnt vmcnt(N)` per staged item (N = loads that may stay in flight: the rest of this set + the whole newer
//     set; tail steps issue dummy loads so that N is one compile-time constant).  Every such load is awaited and its result
//     "used" -- an unused result would leave its registers free for reuse while the load is in flight -- and there is ONE
//     instance of the staging code (several instances make hipcc copy pending registers where their paths meet).
//     tools/asm_lint.py checks in the emitted .s that nothing touches a destination register between a load and its wait
//     (tests/test_boundary.py runs it).
//   * CONSUMERS.  The compiler's tap-by-tap order issued each fragment read right in front of its first use
//     (an exposed LDS round trip per MFMA group) and placed MFMAs on the SAME accumulator back to back.  The K loop is
//     now a hand-ordered software pipeline over HALF-taps (6 MFMAs = 2 pixel blocks x {w_h x_h, w_h x_l, w_l x_h}):
//     the 4-6 `ds_read_b128` of half-tap u + 1 are issued one per MFMA under half-tap u, accumulators alternate, and
//     the order is pinned with sched_barrier.  The pipeline runs ACROSS the chunk barrier: a chunk's last half-tap is
//     held back in registers and executed after the barrier, under the first fragment reads of the next chunk.
//     Per accumulator the products are still added tap by tap as w_h x_h, w_h x_l, w_l x_h: results are bit-identical
//     to the round-2 kernel.
#include <type_traits>

#include "dmd_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef DMD_LAB
#ifndef WS_ABL
#define WS_ABL 0
#endif
#else
#undef WS_ABL
#define WS_ABL 0
#endif
// WS_ABL (DMD_LAB builds only, WRONG results: timing proxies for profiles/): 2 = no activation global loads,
// 16 = no MFMA loop, 32 = no weight movement, 128 = no epilogue global stores / residual loads.

// NCB = 32-output-channel blocks of the convolution (2: Cout = 64, the U-Net; 1: Cout = 32, the reward/end model and
// the first actor-critic blocks).  The workgroup's consumer group is always 4 waves = NCB cout blocks x NPH pixel
// halves of 128 pixels, so a Cout = 32 tile is 512 pixels (two 16x16 patches / eight 8x8 patches).
// TAPS = 9 (3x3, pad 1) or 1 (1x1: the same halo'd patch geometry, only the centre window is loaded and read).
template <bool B8_, int NCB_, int TAPS_ = 9>
struct WsGeom {
  static constexpr int NPT = 256;  // producer threads
  static constexpr bool B8 = B8_;
  static constexpr int NCB = NCB_;
  static constexpr int TAPS = TAPS_;
  static constexpr int COUT = 32 * NCB_;
  static constexpr int NPH = 4 / NCB_;
  static constexpr int SUB = B8_ ? 2 * NPH : NPH / 2;
  static constexpr int TS = B8_ ? 8 : 16;
  static constexpr int PW = TS + 2;
  static constexpr int PPS = PW * PW;
  static constexpr int NPP = SUB * PPS;
  static constexpr int ITEMS = (NPP * 4 + NPT - 1) / NPT;
  static constexpr int W_UNITS = TAPS_ * 2 * 2 * COUT;      // 16-byte units of one chunk's weights
  static constexpr int WU = (W_UNITS + 255) / 256;          // DMA rounds of the idle consumer group
  // LDS: [patch 0][patch 1][weights 0][weights 1][tables].  A fragment read is `ds_read_b128 v, vaddr offset:imm` with
  // the tap window and the pixel block in the immediate; the consumers keep 7 address registers and move them between
  // the buffer pairs once per chunk
  static constexpr int PATCH_BYTES = NPP * 64;
  static constexpr int W_BYTES = W_UNITS * 16;
  static constexpr int W_BASE = 2 * PATCH_BYTES;
  static constexpr int BUF_BYTES = PATCH_BYTES + W_BYTES;
  static constexpr int CIN_MAX = NCB_ == 2 ? 128 : 64;
  static constexpr int TAB_SLOTS = 4;  // tile generations whose tables can be alive at once (>= 3)
  static constexpr int TAB_FLOATS = TAB_SLOTS * SUB * CIN_MAX;
  static constexpr int SMEM_BYTES = 2 * BUF_BYTES + 2 * TAB_FLOATS * 4;
  static constexpr bool DOUBLE_STAGE = NCB_ == 2;  // two activation register sets only where they fit
  // hand-counted inline-asm activation loads (header: PRODUCERS) where there are two register sets to keep apart and the
  // producers have registers to spare; the single-set geometries (11-13 items per thread, at the register cap) keep
  // compiler-counted loads: with one set in flight hipcc's waits are the exact ones
  static constexpr bool ASM_LOADS = DOUBLE_STAGE;
  static constexpr int HALVES = TAPS_ * 2;         // half-taps of one chunk
  // patch byte offset of a consumer wave's 32-pixel block `blk` relative to its block 0
  static constexpr int blk_off(int blk) { return (B8_ ? (blk >> 1) * PPS + (blk & 1) * 4 * PW : blk * 2 * PW) * 64; }
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

// compile-time counted loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
__device__ __forceinline__ void ws_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ws_for<I + 1, N>(f);
  }
}

// ---- activation loads hipcc does not count (header: PRODUCERS) ----
// "=&v": the destination never overlaps the address pair.  Nothing may read or move `dst` before ws_await names it.
__device__ __forceinline__ void ws_aload(f32x4& dst, const f32x4* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; `v` is usable afterwards
template <int N>
__device__ __forceinline__ void ws_await(f32x4& v) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%1) ; await %0" : "+v"(v) : "n"(N) : "memory");
}

// marks every element of a register set as used at this point (no instruction)
template <int N, int M>
__device__ __forceinline__ void ws_use_all(f32x4 (&st)[M]) {
  static_assert(N <= M, "register set");
#pragma unroll
  for (int it = 0; it < N; ++it) asm volatile("; drained %0" : : "v"(st[it]));
}

// ---- consumer fragments ----
struct WsA {
  h8 h, l;  // weight pieces of one tap (MFMA A operand: rows = couts)
};
struct WsB {
  h8 h0, h1, l0, l1;  // activation pieces of the two 32-pixel blocks of a half-tap (MFMA B operand)
};

// LDS byte addresses of a consumer wave's fragments in the buffer pair it currently reads (lane-dependent part; the tap
// window and the pixel block go into the instruction's immediate offset).  ws_addr_flip moves them to the other pair.
struct WsAddr {
  int ph[3], pl[3];  // patch unit of (pixel block 0, window column dx): h piece, l piece (= h ^ 32)
  int w;             // this lane's unit inside a (tap, piece) weight row
};

template <class G>
__device__ __forceinline__ void ws_addr_move(WsAddr& ad, int dpar) {  // dpar = new parity - current parity (wave-uniform)
  const int dp = dpar * G::PATCH_BYTES, dw = dpar * G::W_BYTES;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    ad.ph[dx] += dp;
    ad.pl[dx] += dp;
  }
  ad.w += dw;
}

template <class G, int TAP>
struct WsWin {
  static constexpr int win = G::TAPS == 9 ? TAP : 4;  // window of the 3x3 patch geometry (4 = centre)
  static constexpr int dy = win / 3, dx = win % 3;
  static constexpr int off = (dy * G::PW + dx) * 64;  // bytes
};

template <class G, int TAP, int BLK, bool LOW>
__device__ __forceinline__ h8 ws_read_b(const unsigned char* lds, const WsAddr& ad) {
  using Wn = WsWin<G, TAP>;
  return *(const h8*)(lds + (LOW ? ad.pl[Wn::dx] : ad.ph[Wn::dx]) + (Wn::off + G::blk_off(BLK)));
}
template <class G, int TAP, bool LOW>
__device__ __forceinline__ h8 ws_read_a(const unsigned char* lds, const WsAddr& ad) {
  return *(const h8*)(lds + ad.w + (TAP * 2 + (LOW ? 1 : 0)) * 2 * G::COUT * 16);
}

// One half-tap: 6 MFMAs on the accumulators of pixel blocks (P0, P0 + 1) with operands (a, b), and -- one per MFMA, in
// the order of first use -- the fragment reads of the NEXT half-tap (tap NT, blocks NP0, NP0 + 1) through `ad`:
//   NEXT == 0: nothing to prefetch;  1: the 4 activation pieces;  2: the 4 activation pieces, then the weight pieces of
//   tap NT + 1 into `an` (issued in the first half of a tap for the next tap);  3: weight pieces of tap NT first, then the 4
//   activation pieces (first reads of a chunk, issued under the held-back last half-tap of the previous chunk).
// sched_barrier(0) after every (MFMA, read) pair pins this order (left alone the scheduler sinks every read to just in
// front of its first use and the wave waits out an LDS round trip per MFMA group).
template <class G, int P0, int NEXT, int NT, int NP0>
__device__ __forceinline__ void ws_halftap(f32x16 (&acc)[4], const WsA& a, const WsB& b, const unsigned char* lds, const WsAddr& ad,
                                           WsA& an, WsB& bn) {
#if WS_ABL & 16
  return;
#endif
  constexpr int P1 = P0 + 1, NP1 = NP0 + 1;
  acc[P0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h0, acc[P0], 0, 0, 0);
  if constexpr (NEXT == 3) an.h = ws_read_a<G, NT, false>(lds, ad);
  if constexpr (NEXT == 1 || NEXT == 2) bn.h0 = ws_read_b<G, NT, NP0, false>(lds, ad);
  __builtin_amdgcn_sched_barrier(0);
  acc[P1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h1, acc[P1], 0, 0, 0);
  if constexpr (NEXT == 3) bn.h0 = ws_read_b<G, NT, NP0, false>(lds, ad);
  if constexpr (NEXT == 1 || NEXT == 2) bn.h1 = ws_read_b<G, NT, NP1, false>(lds, ad);
  __builtin_amdgcn_sched_barrier(0);
  acc[P0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.l0, acc[P0], 0, 0, 0);
  if constexpr (NEXT == 3) bn.h1 = ws_read_b<G, NT, NP1, false>(lds, ad);
  if constexpr (NEXT == 1 || NEXT == 2) bn.l0 = ws_read_b<G, NT, NP0, true>(lds, ad);
  __builtin_amdgcn_sched_barrier(0);
  acc[P1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.l1, acc[P1], 0, 0, 0);
  if constexpr (NEXT == 3) bn.l0 = ws_read_b<G, NT, NP0, true>(lds, ad);
  if constexpr (NEXT == 1 || NEXT == 2) bn.l1 = ws_read_b<G, NT, NP1, true>(lds, ad);
  __builtin_amdgcn_sched_barrier(0);
  acc[P0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b.h0, acc[P0], 0, 0, 0);
  if constexpr (NEXT == 3) bn.l1 = ws_read_b<G, NT, NP1, true>(lds, ad);
  if constexpr (NEXT == 2) an.h = ws_read_a<G, NT + 1, false>(lds, ad);
  __builtin_amdgcn_sched_barrier(0);
  acc[P1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b.h1, acc[P1], 0, 0, 0);
  if constexpr (NEXT == 3) an.l = ws_read_a<G, NT, true>(lds, ad);
  if constexpr (NEXT == 2) an.l = ws_read_a<G, NT + 1, true>(lds, ad);
  __builtin_amdgcn_sched_barrier(0);
}

// half-taps [0, HALVES - 1) of one chunk: on entry (a, b) hold the operands of half-tap 0 (and, TAPS > 1, nothing else);
// on exit (a, b) hold the operands of the LAST half-tap, which the caller executes after the chunk's barrier
template <class G>
__device__ __forceinline__ void ws_chunk_body(f32x16 (&acc)[4], WsA& a, WsB& b, const unsigned char* lds, const WsAddr& ad) {
  WsA an = a;
  WsB bn = b;
  ws_for<0, G::HALVES - 1>([&](auto uc) {
    constexpr int U = decltype(uc)::value;
    constexpr int T = U / 2, H = U % 2;
    constexpr int NU = U + 1, NT = NU / 2, NH = NU % 2;
    // first half of a tap: also fetch the next tap's weight pieces (if there is a next tap in this chunk)
    constexpr int NEXT = (H == 0 && T + 1 < G::TAPS) ? 2 : 1;
    ws_halftap<G, 2 * H, NEXT, NT, 2 * NH>(acc, a, b, lds, ad, an, bn);
    b = bn;
    if (H == 1) a = an;  // the next half-tap starts a new tap
  });
}

template <class G>
__global__ __launch_bounds__(768, 3) void conv_f16ws_kernel(const dmd_conv_params p, int ntiles, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // [patch 0][patch 1][weights 0][weights 1]: patch [NPP][4 x 16 B], weights [TAPS][h|l][2][COUT] x 16 B
  float* tab_a = (float*)(smem_raw + 2 * G::BUF_BYTES);  // [slot][SUB][CIN_MAX]
  float* tab_b = tab_a + G::TAB_FLOATS;
  static_assert(G::SMEM_BYTES <= 160 * 1024, "LDS budget");

  // 0, 1: consumer groups (even / odd tiles), 2: producer (staging).  readfirstlane: wave-uniform by construction, and
  // the compiler must know it (scalar branches and scalar loop counters instead of exec-masked ones)
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = (int)(threadIdx.x & 255);  // index inside the role
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  if (role == 2) {
    // =================================== PRODUCER ===================================
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
    int goff[G::ITEMS];   // source pixel index per item for tile gk (0 outside the image)
    unsigned gzero = 0;   // bit it = item is conv zero padding / outside the tensor
    int gtile = 0;        // tile index of tile gk
    auto item_source = [&](int it, bool& inb) -> int {
      const int s = ipos[it] >> 16, py = (ipos[it] >> 8) & 0xff, px = ipos[it] & 0xff;
      // sub-tile of this item: recomputed from its index where a tile has several (selecting from the ti[] registers
      // by a per-lane index turns the array into scratch memory)
      const WsTile t = G::SUB == 1 ? ti[0] : ws_subtile<G>(p, gtile, s);
      const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
      const bool window = G::TAPS == 9 || (py >= 1 && py <= G::TS && px >= 1 && px <= G::TS);  // 1x1: no halo needed
      inb = ipos[it] >= 0 && window && t.valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      return inb ? ((t.n * Hs + (iy >> up)) * Ws + (ix >> up)) : 0;
    };

    auto setup_tile = [&](int k) {  // descriptors + normalisation tables of tile k
      const int tile = WS_TILE(k);
      gtile = tile;
#pragma unroll
      for (int s = 0; s < G::SUB; ++s) ti[s] = ws_subtile<G>(p, tile, s);
      unsigned gz = 0;
#pragma unroll
      for (int it = 0; it < G::ITEMS; ++it) {
        bool inb;
        goff[it] = item_source(it, inb);
        gz |= (inb ? 0u : 1u) << it;
      }
      gzero = gz;
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

    // ITEMS activation loads of stream element e into a register set (+ which items are conv zero padding).  ASM_LOADS:
    // an element beyond the stream loads the LAST element's data once more (same code path, no branch), so that the
    // hand-counted vmcnt of stage_S is one compile-time constant per item; drain_S at the end awaits and "uses" those
    // results (an asm load whose result is never used would leave its registers free for reuse while it is in flight).
    auto issue_S = [&](int e_, auto& st, unsigned& zmask, int& slot_out) {
      const int e = G::ASM_LOADS ? min(e_, S - 1) : e_;
      const int k = e / nchunks, ck = e - k * nchunks;
      if (k != gk) setup_tile(k);
      slot_out = tab_slot;
      const int si = ck < nch0 ? 0 : 1;
      const dmd_conv_src& sc = p.src[si];
      const int c0 = (si ? ck - nch0 : ck) * 16 + 4 * q;
      const f32x4* base4 = (const f32x4*)sc.x + (c0 >> 2);  // 16-byte units: index = pixel * (C / 4)
      const unsigned cq = (unsigned)sc.C >> 2;
      ws_for<0, G::ITEMS>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
#if WS_ABL & 2
        const f32x4* src = (const f32x4*)p.src[0].x;
#else
        const f32x4* src = base4 + (size_t)((unsigned)goff[it] * cq);
#endif
        if constexpr (G::ASM_LOADS)
          ws_aload(st[it], src);
        else
          st[it] = *src;
      });
      zmask = gzero;
    };
    // normalise / activate / split element e from its register set into patch e & 1.  ONE instance of this code for every
    // prologue (the tables hold a = 1, b = 0 where there is no normalisation; SiLU is a wave-uniform select): with
    // several instances hipcc assigns the pending registers differently per instance and copies them -- before the
    // wait -- where the paths meet.  NEWER (ASM_LOADS) = loads issued after this set's: they may stay in flight.
    auto stage_S = [&](int e, auto& st, unsigned zmask, int slot) {
      constexpr int NEWER = G::DOUBLE_STAGE ? G::ITEMS : 0;
      const int ck = e % nchunks;
      const bool silu = p.src[ck < nch0 ? 0 : 1].prologue == DMD_PROLOGUE_NORM_SILU;
      const int cc = ck * 16 + 4 * q;
      uint2* pb = (uint2*)(smem_raw + (e & 1) * G::PATCH_BYTES);
      // single-image tiles: the (a, b) rows of this thread's channel quad are the same for every item -> ONE pair of
      // LDS reads per chunk instead of one (with its lgkmcnt stall) per item
      f32x4 ta0 = (f32x4){1.f, 1.f, 1.f, 1.f}, tb0 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (G::SUB == 1) {
        ta0 = *(const f32x4*)(tab_a + slot * G::CIN_MAX + cc);
        tb0 = *(const f32x4*)(tab_b + slot * G::CIN_MAX + cc);
      }
      ws_for<0, G::ITEMS>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        if constexpr (G::ASM_LOADS) ws_await<NEWER + G::ITEMS - 1 - it>(st[it]);
        f32x4 v = st[it];
        f32x4 ta = ta0, tb = tb0;
        if (G::SUB > 1) {
          const int s = ipos[it] >> 16;
          ta = *(const f32x4*)(tab_a + (slot * G::SUB + s) * G::CIN_MAX + cc);
          tb = *(const f32x4*)(tab_b + (slot * G::SUB + s) * G::CIN_MAX + cc);
        }
        h4 hv, lv;
        const bool zero = (zmask >> it) & 1;  // conv zero padding is applied AFTER the activation (blocks.py:143-144)
#pragma unroll
        for (int el = 0; el < 4; ++el) {
          const float t = __builtin_fmaf(v[el], ta[el], tb[el]);
          const float u = silu ? ws_silu(t) : t;
          const float x = zero ? 0.f : u;  // no clamp: out-of-range operands turn into NaN outputs (header)
          const _Float16 h = (_Float16)x;
          hv[el] = h;
          lv[el] = (_Float16)(x - (float)h);
        }
        if ((it + 1) * G::NPT <= G::NPP * 4 || ipos[it] >= 0) {  // only the last item row can fall beyond the patch
          const int lo = loff_of(it);
          pb[lo] = __builtin_bit_cast(uint2, hv);
          pb[lo ^ 4] = __builtin_bit_cast(uint2, lv);
        }
      });
    };

    f32x4 stage0[G::ITEMS], stage1[G::DOUBLE_STAGE ? G::ITEMS : 1];
    unsigned zm0 = 0, zm1 = 0;
    int sl0 = 0, sl1 = 0;
    if constexpr (G::DOUBLE_STAGE) {
      // Invariant in front of every stage_S(e): outstanding = [the ITEMS loads of element e] [the ITEMS loads of
      // element e + 1, real or dummy], in this order: every stage_S(e) is followed by issue_S(e + 2) into the same set.
      issue_S(0, stage0, zm0, sl0);
      issue_S(1, stage1, zm1, sl1);
      __syncthreads();  // B(-1): the tables written by setup_tile are visible to all producers
      stage_S(0, stage0, zm0, sl0);
      issue_S(2, stage0, zm0, sl0);
      __syncthreads();  // B0: buffer 0 = element 0
      // step j: consumers compute element j, producers fill element j + 1 (register set (j + 1) & 1)
      for (int j = 0; j < S; j += 2) {
        if (j + 1 < S) {
          stage_S(j + 1, stage1, zm1, sl1);
          issue_S(j + 3, stage1, zm1, sl1);
        }
        __syncthreads();
        if (j + 1 < S) {
          if (j + 2 < S) {
            stage_S(j + 2, stage0, zm0, sl0);
            issue_S(j + 4, stage0, zm0, sl0);
          }
          __syncthreads();
        }
      }
      // the two dummy sets of the tail (elements S and S + 1): land, and are "used" here
      if constexpr (G::ASM_LOADS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ws_use_all<G::ITEMS>(stage0);
        ws_use_all<G::ITEMS>(stage1);
      }
    } else {
      // one register set: activations are fetched one step ahead only
      issue_S(0, stage0, zm0, sl0);
      __syncthreads();  // B(-1)
      stage_S(0, stage0, zm0, sl0);
      if (S > 1) issue_S(1, stage0, zm0, sl0);
      __syncthreads();  // B0
      for (int j = 0; j < S; ++j) {
        if (j + 1 < S) {
          stage_S(j + 1, stage0, zm0, sl0);
          if (j + 2 < S) issue_S(j + 2, stage0, zm0, sl0);
        }
        __syncthreads();
      }
    }
  } else {
    // =================================== CONSUMER ===================================
    // static priority: the MFMA waves win issue arbitration against the co-resident staging wave of their SIMD
    __builtin_amdgcn_s_setprio(3);
    const int cb = wave % G::NCB;  // 32-cout block == GroupNorm group
    const int ph = wave / G::NCB;  // 128-pixel part of the tile
    const int n31 = lane & 31, g = lane >> 5;
    WsAddr ad;
    {
      const int col = G::B8 ? (n31 & 7) : (n31 & 15);
      // block 0 of this wave: A16: rows (ph & 1) * 8 + {0, 1} of patch ph >> 1; B8: rows 0..3 of patch 2 ph
      const int pixbase = G::B8 ? (ph * 2 * G::PPS + (n31 >> 3) * G::PW + col)
                                : ((ph >> 1) * G::PPS + ((ph & 1) * 8 + (n31 >> 4)) * G::PW + col);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        ad.ph[dx] = (pixbase * 4 + ((g + ((col + dx) >> 1)) & 3)) * 16;
        ad.pl[dx] = ad.ph[dx] ^ 32;
      }
      ad.w = G::W_BASE + (g * G::COUT + cb * 32 + n31) * 16;
    }
    int apar = 0;  // buffer pair `ad` points to
    const unsigned char* lds = smem_raw;

    f32x16 acc[4];

    // ---- write-out state of this group's finished tile ----
    // lane owns couts cb*32 + 8 qd + 4 g + (0..3), qd = 0..3, of pixel n31 of each 32-pixel block
    int pixoff[4];     // 16-byte unit offset of (pixel, cb*32 + 4 g) per block, -1: sub-tile outside the tensor
    constexpr int NSTAT = G::B8 ? 2 : 1;  // statistics tiles of this wave's 128 pixels (one per 8 rows x 8 | 16 columns)
    int stat_slot[NSTAT];  // out_stats slot per statistics tile of this wave, -1: none
    double ssum[NSTAT], ssq[NSTAT];
    int pending = 4;   // next block of the finished tile to write (4 = nothing pending)
    auto epi_begin = [&](int k) {
      const int tile = WS_TILE(k);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        // sub-tile of this block (wave-uniform index; computed, not selected from a register array)
        const WsTile t = ws_subtile<G>(p, tile, G::B8 ? ph * 2 + (blk >> 1) : (ph >> 1));
        int oy, ox;
        if (G::B8) {
          oy = t.y0 + (blk & 1) * 4 + (n31 >> 3);
          ox = t.x0 + (n31 & 7);
        } else {
          oy = t.y0 + (ph & 1) * 8 + blk * 2 + (n31 >> 4);
          ox = t.x0 + (n31 & 15);
        }
        pixoff[blk] = t.valid ? (((t.n * p.H + oy) * p.W + ox) * (G::COUT / 4) + cb * 8 + g) : -1;  // 16-byte units
      }
#pragma unroll
      for (int kk = 0; kk < NSTAT; ++kk) {
        const WsTile t = ws_subtile<G>(p, tile, G::B8 ? ph * 2 + kk : (ph >> 1));
        int T, tt;
        if (G::B8) {
          const int tx8 = p.W / 8;
          T = tx8 * (p.H / 8);
          tt = (t.y0 / 8) * tx8 + t.x0 / 8;
        } else {
          const int tx16 = p.W / 16;
          T = tx16 * (p.H / 8);
          tt = (t.y0 / 8 + (ph & 1)) * tx16 + t.x0 / 16;
        }
        stat_slot[kk] = t.valid ? ((t.n * G::NCB + cb) * T + tt) : -1;
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
          f32x4 rv[4];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            rv[qd] = (p.residual && !(WS_ABL & 128)) ? *(const f32x4*)(p.residual + (size_t)pixoff[blk] * 4 + 8 * qd) : (f32x4){0.f, 0.f, 0.f, 0.f};
          float fs = 0.f, fq = 0.f;  // fp32 over the lane's 16 values of this block, fp64 across
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 v = (f32x4){acc[blk][4 * qd], acc[blk][4 * qd + 1], acc[blk][4 * qd + 2], acc[blk][4 * qd + 3]};
            v += bias[qd];
            v += rv[qd];
            if (!(WS_ABL & 128) || v[0] == 1.2345e30f) *(f32x4*)(op + 8 * qd) = v;
            fs += (v[0] + v[1]) + (v[2] + v[3]);
            // Sum of squares as an fma chain into its own register, NOT as in-place squares of v: with the squares
            // written over v's registers (`v_mul_f32 v48, v48, v48` right behind the `global_store_dwordx4 v[48:51]`),
            // a slice executed while the other consumer group's MFMAs run on the same SIMD lost one lane's
            // contribution of a block now and then (sum and stored outputs exact, sum of squares short by ~16 values)
            // -- found by tests/test_gpu_tpw.py (tiles_per_wg >= 2), round 2.
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
        for (int kk = 0; kk < NSTAT; ++kk) {
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

    // ---- weight copy of the idle group: chunk ck's pre-split weights into buffer wbuf by LDS-DMA ----
    // (lane l of a wave lands at (wave-uniform base) + 16 l; round i moves units [256 i, 256 i + 256))
    const u32x4* cwglob = (const u32x4*)p.w_f16;
    auto cons_load_W = [&](int ck, int wbuf) {
#if WS_ABL & 32
      if (wbuf >= 0) return;  // ablation: the weights are never moved (LDS keeps whatever it held)
#endif
      const u32x4* w = cwglob + (size_t)ck * G::W_UNITS + tid;
      u32x4* wl = (u32x4*)(smem_raw + G::W_BASE + wbuf * G::W_BYTES) + wave * 64;
#pragma unroll
      for (int i = 0; i < G::WU; ++i)
        if (G::W_UNITS % 256 == 0 || 256 * i + wave * 64 < G::W_UNITS)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + 256 * i),
                                           (__attribute__((address_space(3))) void*)(wl + 256 * i), 16, 0, 0);
    };
    auto cons_land_W = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };  // the DMA writes have landed before the step's barrier

    __syncthreads();  // B(-1)
    if (role == 1) {  // group 1 is idle during tile 0: it provides the first chunk's weights
      cons_load_W(0, 0);
      cons_land_W();
    }
    __syncthreads();  // B0
    int j = 0;
    for (int k = 0; k < nmy; ++k) {
      if ((k & 1) == role) {
        // ---- this group's tile: fragment reads + MFMAs only ----
        if (pending < 4) epi_blocks(4);  // (only if the other group's tile had too few steps to finish the write-out)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
        // operands of the tile's first half-tap (the one exposed LDS round trip per tile)
        ws_addr_move<G>(ad, (j & 1) - apar);
        apar = j & 1;
        WsA a, an;
        WsB b, bn;
        a.h = ws_read_a<G, 0, false>(lds, ad);
        b.h0 = ws_read_b<G, 0, 0, false>(lds, ad);
        b.h1 = ws_read_b<G, 0, 1, false>(lds, ad);
        b.l0 = ws_read_b<G, 0, 0, true>(lds, ad);
        b.l1 = ws_read_b<G, 0, 1, true>(lds, ad);
        a.l = ws_read_a<G, 0, true>(lds, ad);
        an = a;
        bn = b;
        constexpr int LH = (G::HALVES - 1) % 2;
        for (int ck = 0; ck < nchunks; ++ck, ++j) {
          ws_chunk_body<G>(acc, a, b, lds, ad);
          __syncthreads();  // B(j + 1): every fragment of buffer j is in registers; buffer j + 1 is complete
          // the held-back last half-tap, under the first fragment reads of the next chunk (other buffer pair).  After
          // the tile's last chunk these reads fetch the other group's first fragments (or, at the end of the stream,
          // stale LDS) and are simply dropped: unconditional, so that the accumulators stay in one set of registers
          ws_addr_move<G>(ad, 1 - 2 * apar);
          apar ^= 1;
          ws_halftap<G, 2 * LH, 3, 0, 0>(acc, a, b, lds, ad, an, bn);
          a = an;
          b = bn;
        }
        epi_begin(k);  // written out while the other group computes the next tile
      } else {
        // ---- the other group's tile: write our finished tile out, a slice per chunk step, and move the weights ----
        for (int ck = 0; ck < nchunks; ++ck, ++j) {
          const bool wnext = j + 1 < S;  // this (idle) group copies the next step's weights
          if (wnext) cons_load_W((j + 1) % nchunks, (j + 1) & 1);
          if (pending < 4) epi_blocks(blocks_per_step);
          if (wnext) cons_land_W();
          __syncthreads();  // B(j + 1)
        }
      }
    }
    if (pending < 4) epi_blocks(4);  // tail: the last tile(s) of the range
  }
#undef WS_TILE
}

template <class G>
static int launch_f16ws(const dmd_conv_params& p, int ntiles, hipStream_t st) {
  // per device: the LDS attribute and the CU count belong to the device the launch goes to
  static bool attr_set[DMD_MAX_DEVICES] = {};
  static int ncus[DMD_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DMD_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16ws_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::SMEM_BYTES);
    DMD_CHECK_ARG(e == hipSuccess, "conv_f16ws: hipFuncSetAttribute(%d bytes): %s", G::SMEM_BYTES, hipGetErrorString(e));
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncus[dev] = n;
    attr_set[dev] = true;
  }
  // persistent: one 768-thread workgroup per CU (LDS-limited), contiguous tile ranges (neighbouring tiles share
  // halo rows and, inside one image, the normalisation statistics)
  const int ncu = ncus[dev];
  const int tpw = (ntiles + ncu - 1) / ncu;
  const int nwg = (ntiles + tpw - 1) / tpw;
  hipLaunchKernelGGL((conv_f16ws_kernel<G>), dim3(nwg), dim3(768), G::SMEM_BYTES, st, p, ntiles, tpw);
  return 0;
}

int dmd_launch_conv_f16ws(const dmd_conv_params& p, hipStream_t st) {
  const bool b8 = p.W % 16 != 0;
  const int sub8 = p.N * (p.H / 8) * (p.W / 8), t16 = p.N * (p.H / 16) * (p.W / 16);
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
