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
// Round 3 (what the ISA of the round-2 kernel showed, HISTORY.md §3):
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
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
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
// WS_WRITEOUT_PRIO (DMD_LAB builds only, same results): s_setprio of a consumer group while it WRITES ITS TILE OUT (3 while it computes; the
// shipped library keeps 3 throughout): does the computing group's MFMA chain gain from winning arbitration against the write-out wave?
#if defined(DMD_LAB) && defined(WS_WRITEOUT_PRIO)
#define WS_PRIO_COMPUTE() __builtin_amdgcn_s_setprio(3)
#define WS_PRIO_WRITEOUT() __builtin_amdgcn_s_setprio(WS_WRITEOUT_PRIO)
#else
#define WS_PRIO_COMPUTE() do {} while (0)
#define WS_PRIO_WRITEOUT() do {} while (0)
#endif
// WS_ABL (DMD_LAB builds only, WRONG results: timing proxies for profiles/): 2 = no activation global loads, 4 = no staging arithmetic,
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
  // tables a, b (y = a x + b), c, d (exponent argument of the sigmoid, c x + d) + the bias row of the convolution.
  // The 8-patch geometry has no room for c, d (8 table rows per slot): it derives them from a, b per item.
  static constexpr bool CD_TABLES = !(B8_ && NCB_ == 1);
  static constexpr int SMEM_BYTES = 2 * BUF_BYTES + (CD_TABLES ? 4 : 2) * TAB_FLOATS * 4 + COUT * 4;
  static constexpr bool DOUBLE_STAGE = NCB_ == 2;  // two activation register sets only where they fit
  // hand-counted inline-asm activation loads (header: PRODUCERS) where there are two register sets to keep apart and the
  // producers have registers to spare; the single-set geometries (11-13 items per thread, at the register cap) keep
  // compiler-counted loads: with one set in flight hipcc's waits are the exact ones
  static constexpr bool ASM_LOADS = DOUBLE_STAGE;
  static constexpr int HALVES = TAPS_ * 2;         // half-taps of one chunk
  // address registers of the l pieces (otherwise h ^ 32 at every l read: the 13-item geometry is 3 registers short)
  static constexpr bool PL_REGS = !(B8_ && NCB_ == 1);
  // patch byte offset of a consumer wave's 32-pixel block `blk` relative to its block 0
  static constexpr int blk_off(int blk) { return (B8_ ? (blk >> 1) * PPS + (blk & 1) * 4 * PW : blk * 2 * PW) * 64; }
  static constexpr bool PROJ = false;  // fused skip projection (WsGeomProj)
  static constexpr int PROJ_KS = 1;
};

// PROJECTION.  The up path's ResBlocks compute  proj(cat(x, skip)) + conv2(...)  (blocks.py:133,147) with a 128 -> 64
// 1x1 projection: on its own that is an HBM-bound pass of 805 MB per launch at the 64x64 level (read 128 channels, write
// 64, read them back as conv2's residual) and 11 % of the imagination step.  This geometry adds the projection to the
// write-out of conv2 instead: the write-out wave of a 32-pixel x 32-cout block fetches the block's 128 raw input
// channels one chunk step ahead (16 x 16 B per lane -- the NHWC row IS the K-contiguous B operand, as in
// dmd_conv1x1.hip), splits them in registers and runs 8 k-steps x {w_h x_h, w_h x_l, w_l x_h} into the block's
// accumulators in front of its stores; the projection's 32 KiB of pre-split weights stay in LDS for the lifetime of the
// workgroup.  Same MACs as the separate launch, 536 MB less HBM traffic, no residual read.
struct WsGeomProj : WsGeom<false, 2, 9> {
  static constexpr bool PROJ = true;
  static constexpr int PROJ_KS = 8;                               // 16-channel k-steps: two 64-channel sources
  static constexpr int PROJ_BYTES = PROJ_KS * 2 * 2 * COUT * 16;  // [k-step][h|l][k group][cout] x 16 B
  static constexpr bool PL_REGS = false;  // at the register cap: the l pieces' addresses are h ^ 32 at every read
  static constexpr int SMEM_BYTES = WsGeom<false, 2, 9>::SMEM_BYTES + PROJ_BYTES;
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

// The chunk barrier.  Not __syncthreads(): its fence makes hipcc drain every vector-memory operation it knows of
// (`s_waitcnt vmcnt(0)`) in front of s_barrier, i.e. the write-out group would wait for the acknowledgement of its
// global STORES at every step.  LDS traffic of this wave is complete (lgkmcnt(0)); LDS-DMA writes are awaited by hand
// where they are issued (cons_land_W); global stores simply stay in flight.
#ifndef WS_SYNCTHREADS
#define WS_SYNCTHREADS 0
#endif
__device__ __forceinline__ void ws_barrier() {
#if WS_SYNCTHREADS
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// WS_TRACE (DMD_LAB builds only): s_memtime stamps of workgroup WS_TRACE_WG, one stream per role, read back with
// tools/ws_trace.py: which role waits for which in a chunk step.
#if defined(DMD_LAB) && defined(WS_TRACE)
#define WS_TRACE_WG (gridDim.x > 37 ? 37 : 0)
#define WS_TRACE_N 4096
__device__ unsigned long long ws_trace_buf[3][WS_TRACE_N];
__device__ int ws_trace_cnt[3];
#define WS_TRACE_DECL int ws_ti = 0
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
#define WS_TRACE_DECL
#define WS_STAMP(role_, tag_, step_) do {} while (0)
#endif

// ---- the inline-assembly helpers.  tests/simt (the host build of these sources for the SIMT interpreter) defines
//      WS_HOST_HELPERS and supplies C++ spellings of the same operations: loads are synchronous there, waits are no-ops ----
#ifndef WS_HOST_HELPERS
// ---- activation loads hipcc does not count (header: PRODUCERS) ----
// "=&v": the destination never overlaps the address pair.  Nothing may read or move `dst` before ws_await names it.
__device__ __forceinline__ void ws_aload(f32x4& dst, const f32x4* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
}
// the same with a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit address arithmetic per load
// (FIRST: the base may come straight out of v_readfirstlane -- a VALU write of an SGPR needs 5 wait states before a
//  vector-memory instruction reads it, and hipcc pads nothing inside an asm statement)
template <bool FIRST>
__device__ __forceinline__ void ws_aload(f32x4& dst, const void* base, unsigned voff) {
  if constexpr (FIRST)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(base) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(base) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; `v` is usable afterwards
template <int N>
__device__ __forceinline__ void ws_await(f32x4& v) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%1) ; await %0" : "+v"(v) : "n"(N) : "memory");
}

// l pieces of two operands: fp16(x - h) as ONE mixed-precision fma per element (v_fma_mix reads the fp16 h piece
// directly and rounds the exact fp32 difference to fp16) instead of cvt_f32_f16 + sub + cvt_f16_f32
__device__ __forceinline__ unsigned ws_low_pair(float x0, float x1, unsigned h01) {
  unsigned l01;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(l01)
      : "v"(x0), "v"(x1), "v"(h01));
  return l01;
}
// `t` = an opaque copy of itself: the compiler may not reason about its value (no instruction)
#define WS_OPAQUE(t) asm volatile("" : "+v"(t))
// wait until at most n vector-memory operations of this wave are outstanding (a literal)
#define WS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif

// a pointer the compiler must treat as wave-uniform (it is: kernel arguments and the stream position)
__device__ __forceinline__ const char* ws_uniform_ptr(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
// wave64 sum of a double on the DPP network (no LDS round trips, unlike __shfl_xor): quad swaps and row mirrors leave
// the 16-lane row total in every lane of a row, row_bcast:15 / row_bcast:31 chain the four rows; total in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double ws_dpp_add(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return x + __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double ws_wave_sum_lane63(double x) {
  x = ws_dpp_add<0xB1, 0xf>(x);   // quad_perm [1, 0, 3, 2]
  x = ws_dpp_add<0x4E, 0xf>(x);   // quad_perm [2, 3, 0, 1]
  x = ws_dpp_add<0x141, 0xf>(x);  // row_half_mirror
  x = ws_dpp_add<0x140, 0xf>(x);  // row_mirror
  x = ws_dpp_add<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3
  x = ws_dpp_add<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3
  return x;
}

#ifndef WS_HOST_HELPERS
// one wait for a whole register set: the loads were issued two chunk steps earlier and have landed long before; what
// matters is that the compiler may then interleave the staging arithmetic of ALL items (a staging wave is bound by the
// latency of its dependent fma -> exp -> add -> rcp -> mul -> cvt chains, tools/probe/simd_share_probe.hip: ~10 cycles
// per instruction with the 4 chains of one item in flight, 4 with enough of them)
template <int N, int M>
__device__ __forceinline__ void ws_await_set(f32x4 (&st)[M]) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  static_assert(M == 6 || M == 7, "register sets of the double-staged geometries");
  if constexpr (M == 6)
    asm volatile("s_waitcnt vmcnt(%6) ; await %0 %1 %2 %3 %4 %5"
                 : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]), "+v"(st[5]) : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%7) ; await %0 %1 %2 %3 %4 %5 %6"
                 : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]), "+v"(st[5]), "+v"(st[6]) : "n"(N) : "memory");
}

// marks every element of a register set as used at this point (no instruction)
template <int N, int M>
__device__ __forceinline__ void ws_use_all(f32x4 (&st)[M]) {
  static_assert(N <= M, "register set");
#pragma unroll
  for (int it = 0; it < N; ++it) asm volatile("; drained %0" : : "v"(st[it]));
}
#endif

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
    if (G::PL_REGS) ad.pl[dx] += dp;
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
  return *(const h8*)(lds + (LOW ? (G::PL_REGS ? ad.pl[Wn::dx] : (ad.ph[Wn::dx] ^ 32)) : ad.ph[Wn::dx]) + (Wn::off + G::blk_off(BLK)));
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
  DMD_DYNAMIC_LDS(unsigned char, smem_raw);
  // [patch 0][patch 1][weights 0][weights 1]: patch [NPP][4 x 16 B], weights [TAPS][h|l][2][COUT] x 16 B
  float* tab_a = (float*)(smem_raw + 2 * G::BUF_BYTES);  // [slot][SUB][CIN_MAX]
  float* tab_b = tab_a + G::TAB_FLOATS;
  float* tab_c = tab_b + G::TAB_FLOATS;  // (CD_TABLES)
  float* tab_d = tab_c + G::TAB_FLOATS;
  float* bias_lds = tab_b + (G::CD_TABLES ? 3 : 1) * G::TAB_FLOATS;  // [COUT]
  static_assert(G::SMEM_BYTES <= 160 * 1024, "LDS budget");
  unsigned char* proj_lds = (unsigned char*)(bias_lds + G::COUT);  // (PROJ) the projection's weights
  if constexpr (G::PROJ) {
    const u32x4* pwg = (const u32x4*)p.proj_w_f16;
    for (int u = (int)threadIdx.x; u < G::PROJ_BYTES / 16; u += 768) ((u32x4*)proj_lds)[u] = pwg[u];  // visible after B(-1)
  }

  // 0, 1: consumer groups (even / odd tiles), 2: producer (staging).  readfirstlane: wave-uniform by construction, and
  // the compiler must know it (scalar branches and scalar loop counters instead of exec-masked ones)
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = (int)(threadIdx.x & 255);  // index inside the role
  WS_TRACE_DECL;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  WS_STAMP(role, 13, 0);  // kernel entry
  const int up = p.upsample;
  const int Hs = p.H >> up, Ws = p.W >> up;
  // valid extent (include/diamond_hip.h): conv-input coordinates (= output coordinates: stride 1) and stored-source ones
  const int Hv = p.valid_h ? p.valid_h : p.H, Wv = p.valid_w ? p.valid_w : p.W;
  const int Hvs = Hv >> up, Wvs = Wv >> up;
  const int C0 = p.src[0].C;
  const int C1 = p.nsrc > 1 ? p.src[1].C : 0;
  const int nch0 = C0 >> 4;
  const int nchunks = (C0 + C1) >> 4;
  const int tile0 = blockIdx.x * tiles_per_wg;
  const int nmy = min(tiles_per_wg, ntiles - tile0);
  const int S = nmy * nchunks;  // chunk stream of this workgroup
  // (Workgroup i runs on XCD i % 8; giving the workgroups of one XCD NEIGHBOURING ranges -- shared halo rows in one L2 --
  // was measured: 134-135 vs 130-131 us per launch over the bench window, i.e. slower; plain order stays.)
  // Workgroups walk their tile range from different starting points: with one image per workgroup all CUs would
  // otherwise touch the same (row, column) offsets of 1 MiB-strided images at the same time (same low address
  // bits -> the same HBM channels).
  const int rot = (blockIdx.x * 7) % nmy;
  // (Walking the range backwards in every other launch -- so that a layer starts with the tiles its producer wrote last, the
  // part of a 268 MB tensor still in the 256 MB Infinity Cache -- was measured in round 4: no difference, 125.5 vs 125.4 us per
  // launch over the bench window; a 64-image launch whose tensors fit the cache entirely runs the same time per image.)
#define WS_TILE(k) (tile0 + (((k) + rot) >= nmy ? (k) + rot - nmy : (k) + rot))

  if (role == 2) {
    // =================================== PRODUCER ===================================
    const int q = tid & 3;
    // patch position of item `it` of this thread: (sub << 16) | (py << 8) | px; -1: no item (beyond the patch).  Kept in
    // registers, except in the 13-item geometry (eight 8x8 patches x 32 couts, at the register cap), which recomputes it
    // from an OPAQUE copy of the thread index wherever it is used (divisions by constants)
    constexpr bool IPOS_REGS = !(G::B8 && G::NCB == 1);
    auto ipos_calc = [&](int it, int t) -> int {
      const int pp = (it * G::NPT + t) >> 2;
      const bool ok = pp < G::NPP;
      const int s = G::SUB == 1 ? 0 : (ok ? pp / G::PPS : 0);
      const int rem = pp - s * G::PPS;
      const int py = rem / G::PW, px = rem - py * G::PW;
      return ok ? ((s << 16) | (py << 8) | px) : -1;
    };
    int ipos_[IPOS_REGS ? G::ITEMS : 1];
#pragma unroll
    for (int it = 0; it < (IPOS_REGS ? G::ITEMS : 1); ++it) ipos_[it] = ipos_calc(it, tid);
    auto ipos = [&](int it) -> int {
      if (IPOS_REGS) return ipos_[IPOS_REGS ? it : 0];
      int t = tid;
      WS_OPAQUE(t);
      return ipos_calc(it, t);
    };
    // 8-byte unit index of the h half-quad of item `it` in a patch.  The 11-13-item geometries are at the register cap:
    // there the offsets are recomputed at every chunk from an OPAQUE copy of the thread index (left alone, hipcc hoists
    // one loop-invariant address register per item and piece and spills something else)
    auto loff_of = [&](int it) {
      int t = tid;
      if (G::SUB > 1) WS_OPAQUE(t);
      const int qq = t & 3;
      return ((it * (G::NPT / 4) + (t >> 2)) * 4 + (((qq >> 1) + ((ipos(it) & 0xff) >> 1)) & 3)) * 2 + (qq & 1);
    };
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
      const int ip = ipos(it);
      const int s = ip >> 16, py = (ip >> 8) & 0xff, px = ip & 0xff;
      // sub-tile of this item: recomputed from its index where a tile has several (selecting from the ti[] registers
      // by a per-lane index turns the array into scratch memory)
      const WsTile t = G::SUB == 1 ? ti[0] : ws_subtile<G>(p, gtile, s);
      const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
      const bool window = G::TAPS == 9 || (py >= 1 && py <= G::TS && px >= 1 && px <= G::TS);  // 1x1: no halo needed
      inb = ip >= 0 && window && t.valid && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
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
            norm_entry(sc.norm, ti[s].n, cl, sc.C, (double)DMD_GN_GROUP * Hvs * Wvs, &m, &a, &ad);
          const float bb = ad - m * a;
          const bool silu = sc.prologue == DMD_PROLOGUE_NORM_SILU;
          const int ti_ = (slot * G::SUB + s) * G::CIN_MAX + c;
          tab_a[ti_] = a;
          tab_b[ti_] = bb;
          // sigmoid(t) = 1 / (1 + 2^(c x + d)), (c, d) = -log2(e) (a, b); no SiLU: 2^-126 -> the factor is exactly 1
          if (G::CD_TABLES) {
            tab_c[ti_] = silu ? -1.4426950408889634f * a : 0.f;
            tab_d[ti_] = silu ? -1.4426950408889634f * bb : -126.f;
          }
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
      // source bytes of item `it` = wave-uniform base (source + this chunk's 64 bytes) + pixel * C * 4 + this lane's quad;
      // 24-bit multiply: pixel indices and C * 4 are far below 2^24, the product fits 32 bits (launch check)
      const char* base = ws_uniform_ptr((const char*)sc.x + (si ? ck - nch0 : ck) * 64);
      const unsigned cbytes = (unsigned)sc.C * 4u;
      ws_for<0, G::ITEMS>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
#if WS_ABL & 2
        const unsigned voff = 16u * q;
#else
        const unsigned voff = __umul24((unsigned)goff[it], cbytes) + 16u * q;
#endif
        if constexpr (G::ASM_LOADS)
          ws_aload<it == 0>(st[it], base, voff);
        else
          st[it] = *(const f32x4*)(base + voff);
      });
      zmask = gzero;
    };
    // normalise / activate / split element e from its register set into patch e & 1.  ONE instance of this code for every
    // prologue (the tables hold a = 1, b = 0 where there is no normalisation and an exponent of -126 where there is no
    // SiLU): with several instances hipcc assigns the pending registers differently per instance and copies them --
    // before the wait -- where the paths meet.  34 VALU instructions per item (54 in round 2): every VALU instruction
    // of a staging wave delays the MFMA issue of the consumer wave on its SIMD by a few cycles (HISTORY.md §3).
    // NEWER (ASM_LOADS) = loads issued after this set's: they may stay in flight.
    // e_next >= 0: the register set is re-issued for that element (two steps ahead) as soon as the set has been read -- right
    // after the first arithmetic phase where one batch covers the whole set: the loads then leave one by one while the
    // vector-memory queue drains, instead of as a burst at the end of the step
    auto stage_S = [&](int e, auto& st, unsigned& zm_set, int& sl_set, int e_next) {
      const unsigned zmask = zm_set;
      const int slot = sl_set;
      auto reissue = [&]() __attribute__((always_inline)) {
        if (e_next >= 0) issue_S(e_next, st, zm_set, sl_set);
      };
      constexpr int NEWER = G::DOUBLE_STAGE ? G::ITEMS : 0;
      const int ck = e % nchunks;
      const int cc = ck * 16 + 4 * q;
      uint2* pb = (uint2*)(smem_raw + (e & 1) * G::PATCH_BYTES);
      // single-image tiles: the table rows of this thread's channel quad are the same for every item -> ONE set of
      // LDS reads per chunk instead of one (with its lgkmcnt stall) per item
      const bool silu_ck = p.src[ck < nch0 ? 0 : 1].prologue == DMD_PROLOGUE_NORM_SILU;  // (only without c, d tables)
      f32x4 ta0, tb0, tc0, td0;
      if (G::SUB == 1) {
        ta0 = *(const f32x4*)(tab_a + slot * G::CIN_MAX + cc);
        tb0 = *(const f32x4*)(tab_b + slot * G::CIN_MAX + cc);
        tc0 = *(const f32x4*)(tab_c + slot * G::CIN_MAX + cc);
        td0 = *(const f32x4*)(tab_d + slot * G::CIN_MAX + cc);
      }
      if constexpr (G::ASM_LOADS) ws_await_set<NEWER>(st);
      WS_STAMP(2, 7, e - 1);
      // PHASE-wise over batches of items (fma | exp | add | rcp | mul + select | split), pinned with sched_barrier: every
      // phase is BATCH x 4 independent instructions.  Item by item (the compiler's choice: fewest live registers) the
      // wave runs one dependent chain of quarter-rate ops after the other and takes ~10 cycles per instruction.
      // by the registers the geometry has left (several sub-tiles: four table rows per ITEM are live as well)
      constexpr int BATCH = G::SUB == 1 ? G::ITEMS : 1;
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      ws_for<0, (G::ITEMS + BATCH - 1) / BATCH>([&](auto bc) {
        constexpr int I0 = decltype(bc)::value * BATCH;
        constexpr int NB = (I0 + BATCH <= G::ITEMS) ? BATCH : G::ITEMS - I0;
        float t[NB][4], u[NB][4];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int it = I0 + i;
          f32x4 ta = ta0, tb = tb0, tc = tc0, td = td0;
          if (G::SUB > 1) {
            const int to = (slot * G::SUB + (ipos(it) >> 16)) * G::CIN_MAX + cc;
            ta = *(const f32x4*)(tab_a + to);
            tb = *(const f32x4*)(tab_b + to);
            if (G::CD_TABLES) {
              tc = *(const f32x4*)(tab_c + to);
              td = *(const f32x4*)(tab_d + to);
            } else {
#pragma unroll
              for (int el = 0; el < 4; ++el) {
                tc[el] = silu_ck ? -1.4426950408889634f * ta[el] : 0.f;
                td[el] = silu_ck ? -1.4426950408889634f * tb[el] : -126.f;
              }
            }
          }
#if defined(DMD_LAB) && defined(WS_PK_MATH)
          // (lab build, round 5: the two fmas as packed fp32 -- v_pk_fma_f32 handles two values per lane in one VALU slot; the same
          //  IEEE fma per value: bit-identical.  Measured: profiles/r05x_pk_math_*.txt)
          typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            const f2v x2 = {st[it][2 * e2], st[it][2 * e2 + 1]};
            const f2v t2 = __builtin_elementwise_fma(x2, (f2v){ta[2 * e2], ta[2 * e2 + 1]}, (f2v){tb[2 * e2], tb[2 * e2 + 1]});
            const f2v u2 = __builtin_elementwise_fma(x2, (f2v){tc[2 * e2], tc[2 * e2 + 1]}, (f2v){td[2 * e2], td[2 * e2 + 1]});
            t[i][2 * e2] = t2[0], t[i][2 * e2 + 1] = t2[1];
            u[i][2 * e2] = u2[0], u[i][2 * e2 + 1] = u2[1];
          }
#else
#pragma unroll
          for (int el = 0; el < 4; ++el) {
            // y = a v + b (GroupNorm / FiLM; a = 1, b = 0 without a prologue); SiLU = y / (1 + 2^(c v + d))
            t[i][el] = __builtin_fmaf(st[it][el], ta[el], tb[el]);
            u[i][el] = __builtin_fmaf(st[it][el], tc[el], td[el]);
          }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BATCH == G::ITEMS) {
          reissue();
          __builtin_amdgcn_sched_barrier(0);
        }
#if !(WS_ABL & 4)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int el = 0; el < 4; ++el) u[i][el] = __builtin_amdgcn_exp2f(u[i][el]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int el = 0; el < 4; ++el) u[i][el] = 1.0f + u[i][el];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int el = 0; el < 4; ++el) u[i][el] = __builtin_amdgcn_rcpf(u[i][el]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const bool zero = (zmask >> (I0 + i)) & 1;  // conv zero padding is applied AFTER the activation (blocks.py:143-144)
#if defined(DMD_LAB) && defined(WS_PK_MATH)
          typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            const f2v y2 = (f2v){t[i][2 * e2], t[i][2 * e2 + 1]} * (f2v){u[i][2 * e2], u[i][2 * e2 + 1]};  // v_pk_mul_f32
            u[i][2 * e2] = zero ? 0.f : y2[0];
            u[i][2 * e2 + 1] = zero ? 0.f : y2[1];
          }
#else
#pragma unroll
          for (int el = 0; el < 4; ++el) {
            const float y = t[i][el] * u[i][el];
            u[i][el] = zero ? 0.f : y;  // no clamp: out-of-range operands turn into NaN outputs (header)
          }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int it = I0 + i;
          const float* x = u[i];
          const unsigned h01 = __builtin_bit_cast(unsigned, (h2){(_Float16)x[0], (_Float16)x[1]});
          const unsigned h23 = __builtin_bit_cast(unsigned, (h2){(_Float16)x[2], (_Float16)x[3]});
          const unsigned l01 = ws_low_pair(x[0], x[1], h01);
          const unsigned l23 = ws_low_pair(x[2], x[3], h23);
          if ((it + 1) * G::NPT <= G::NPP * 4 || ipos(it) >= 0) {  // only the last item row can fall beyond the patch
            const int lo = loff_of(it);
            pb[lo] = (uint2){h01, h23};
            pb[lo ^ 4] = (uint2){l01, l23};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (BATCH != G::ITEMS) reissue();
    };

    f32x4 stage0[G::ITEMS], stage1[G::DOUBLE_STAGE ? G::ITEMS : 1];
    unsigned zm0 = 0, zm1 = 0;
    int sl0 = 0, sl1 = 0;
    if constexpr (G::DOUBLE_STAGE) {
      // Invariant in front of every stage_S(e): outstanding = [the ITEMS loads of element e] [the ITEMS loads of
      // element e + 1, real or dummy], in this order: every stage_S(e) is followed by issue_S(e + 2) into the same set.
      issue_S(0, stage0, zm0, sl0);
      issue_S(1, stage1, zm1, sl1);
      WS_STAMP(2, 14, 0);
      ws_barrier();  // B(-1): the tables written by setup_tile are visible to all producers
      WS_STAMP(2, 15, 0);
      stage_S(0, stage0, zm0, sl0, 2);
      WS_STAMP(2, 12, 0);
      ws_barrier();  // B0: buffer 0 = element 0
      // step j: consumers compute element j, producers fill element j + 1 (register set (j + 1) & 1)
      for (int j = 0; j < S; j += 2) {
        WS_STAMP(2, 0, j);
        if (j + 1 < S) {
          stage_S(j + 1, stage1, zm1, sl1, j + 3);
          WS_STAMP(2, 1, j);
          WS_STAMP(2, 2, j);
        }
        ws_barrier();
        WS_STAMP(2, 3, j);
        if (j + 1 < S) {
          WS_STAMP(2, 0, j + 1);
          if (j + 2 < S) {
            stage_S(j + 2, stage0, zm0, sl0, j + 4);
            WS_STAMP(2, 1, j + 1);
            WS_STAMP(2, 2, j + 1);
          }
          ws_barrier();
          WS_STAMP(2, 3, j + 1);
        }
      }
      // the two dummy sets of the tail (elements S and S + 1): land, and are "used" here
      if constexpr (G::ASM_LOADS) {
        WS_WAIT_VM(0);
        ws_use_all<G::ITEMS>(stage0);
        ws_use_all<G::ITEMS>(stage1);
      }
    } else {
      // one register set: activations are fetched one step ahead only
      issue_S(0, stage0, zm0, sl0);
      ws_barrier();  // B(-1)
      stage_S(0, stage0, zm0, sl0, S > 1 ? 1 : -1);
      ws_barrier();  // B0
      for (int j = 0; j < S; ++j) {
        WS_STAMP(2, 0, j);
        if (j + 1 < S) {
          stage_S(j + 1, stage0, zm0, sl0, j + 2 < S ? j + 2 : -1);
          WS_STAMP(2, 1, j);
          WS_STAMP(2, 2, j);
        }
        ws_barrier();
        WS_STAMP(2, 3, j);
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
    // 16-byte unit offset of (pixel, cb*32 + 4 g) per block, -1: sub-tile outside the tensor.  16x16 patches: the four
    // blocks of a wave are rows 2 blk + {0, 1} of ONE sub-tile; 8x8 patches: blocks 2 s + {0, 1} are rows 0-3 / 4-7 of
    // sub-tile s: one base register per sub-tile + a uniform stride
    constexpr int NPO = G::B8 ? 2 : 1;
    int pixoff_[NPO];
    auto pixoff_of = [&](int blk) {
      const int base = pixoff_[G::B8 ? (blk >> 1) : 0];
      const int rows = G::B8 ? (blk & 1) * 4 : blk * 2;
      return base < 0 ? -1 : base + rows * p.W * (G::COUT / 4);
    };
    // bit blk: this lane's pixel of block blk lies outside the valid extent (its output is stored, but stays out of the
    // GroupNorm partial sums); always 0 without a valid extent.  (PROJ launches have none: eligibility.)
    int dead = 0;
    constexpr int NSTAT = G::B8 ? 2 : 1;  // statistics tiles of this wave's 128 pixels (one per 8 rows x 8 | 16 columns)
    int stat_slot[NSTAT];  // out_stats slot per statistics tile of this wave, -1: none
    // per-lane partial sums of a statistics tile: fp64 across a 16x16 geometry's four blocks; the 8x8 geometries (two blocks
    // = 32 values per lane and tile, and short of registers) keep them in fp32 -- fp64 from the wave reduction on
    using StatAcc = std::conditional_t<G::B8 || G::PROJ, float, double>;  // (PROJ: at the register cap; 4 fp32 block sums per lane)
    StatAcc ssum[NSTAT], ssq[NSTAT];
    int pending = 4;   // next block of the finished tile to write (4 = nothing pending)
    // residual of the NEXT block to write, fetched a chunk step ahead (zeros without a residual): the write-out wave
    // otherwise sits out an HBM round trip between its DMA issue and its stores, at every step
    // (32-cout geometries: no registers to spare -- fetched where it is used)
    constexpr bool RES_PREFETCH = G::NCB == 2 && !G::PROJ;  // (PROJ: the projection is the residual)
    f32x4 rnext[4];
    auto res_prefetch = [&](int blk) __attribute__((always_inline)) {
      const int po = pixoff_of(blk);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        rnext[qd] = (p.residual && po >= 0 && !(WS_ABL & 128)) ? *(const f32x4*)(p.residual + (size_t)po * 4 + 8 * qd) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // (PROJ) raw input channels of the block being projected at this lane's pixel: unit u (= 16-channel k-step u of
    // cat(proj_x[0], proj_x[1])) = xq[2 u], xq[2 u + 1] = channels 16 u + 8 g + (0..7).  All 8 units of a block are fetched
    // one chunk step ahead, ROLLING: unit u of block s + 1 is requested into unit u's registers right behind the MFMAs
    // that consumed unit u of block s, so every request is a whole block of MFMAs + the stores + the barrier old when
    // it is needed (under load a vector-memory round trip takes ~3,000 cycles here, L2 hit or not: the first version --
    // source 0 a step ahead, source 1 inside the step -- had two exposed round trips per step and ran at half speed,
    // profiles/r03_ws_timeline_trace.txt).
    f32x4 xq[G::PROJ ? 2 * G::PROJ_KS : 1];
    auto proj_fetch_unit = [&](int blk, auto uc) __attribute__((always_inline)) {
      if constexpr (G::PROJ) {
        constexpr int u = decltype(uc)::value;
        const int po = pixoff_of(blk);
        // uniform base (SGPR pair) + 32-bit byte offset: po = pixel * (COUT / 4) + cb * 8 + g  ->  pixel * 256 + 32 g
        const unsigned boff = po < 0 ? 0u : ((((unsigned)(po - (cb * 8 + g)) >> 4) << 8) + 32u * g);
        const char* xs = (const char*)p.proj_x[u / 4] + boff + (u % 4) * 64;
        xq[2 * u] = *(const f32x4*)xs;
        xq[2 * u + 1] = *(const f32x4*)(xs + 16);
      }
    };
    // (PROJ) k-step u of the projection (operands in xq) into the accumulators of block BLK
    auto proj_mfma_unit = [&](f32x16& dacc, auto uc) __attribute__((always_inline)) {
      if constexpr (G::PROJ) {
        constexpr int ks = decltype(uc)::value;
        // (this lane's unit inside a weight row is the one of the 3x3 weights: derived from ad.w, no register of its own)
        const unsigned char* pw = proj_lds + (ad.w - G::W_BASE - apar * G::W_BYTES);
        const f32x4 x0 = xq[2 * ks], x1 = xq[2 * ks + 1];
        const unsigned h01 = __builtin_bit_cast(unsigned, (h2){(_Float16)x0[0], (_Float16)x0[1]});
        const unsigned h23 = __builtin_bit_cast(unsigned, (h2){(_Float16)x0[2], (_Float16)x0[3]});
        const unsigned h45 = __builtin_bit_cast(unsigned, (h2){(_Float16)x1[0], (_Float16)x1[1]});
        const unsigned h67 = __builtin_bit_cast(unsigned, (h2){(_Float16)x1[2], (_Float16)x1[3]});
        const h8 xh = __builtin_bit_cast(h8, (u32x4){h01, h23, h45, h67});
        const h8 xl = __builtin_bit_cast(h8, (u32x4){ws_low_pair(x0[0], x0[1], h01), ws_low_pair(x0[2], x0[3], h23),
                                                     ws_low_pair(x1[0], x1[1], h45), ws_low_pair(x1[2], x1[3], h67)});
        const h8 wh = *(const h8*)(pw + (ks * 2 + 0) * 2 * G::COUT * 16);
        const h8 wl = *(const h8*)(pw + (ks * 2 + 1) * 2 * G::COUT * 16);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, dacc, 0, 0, 0);
      }
    };
    auto epi_begin = [&](int k) {
      const int tile = WS_TILE(k);
#pragma unroll
      for (int s = 0; s < NPO; ++s) {
        // sub-tile (wave-uniform index; computed, not selected from a register array) and this lane's pixel of its block 0
        const WsTile t = ws_subtile<G>(p, tile, G::B8 ? ph * 2 + s : (ph >> 1));
        int oy, ox;
        int n31e = n31;
        if constexpr (G::PROJ) {  // at the register cap: recomputed per tile from an OPAQUE copy of the thread index
          int te = tid;           // (left alone, hipcc keeps the row term in a register of its own and spills it)
          WS_OPAQUE(te);
          n31e = te & 31;
        }
        if (G::B8) {
          oy = t.y0 + (n31e >> 3);
          ox = t.x0 + (n31e & 7);
        } else {
          oy = t.y0 + (ph & 1) * 8 + (n31e >> 4);
          ox = t.x0 + (n31e & 15);
        }
        pixoff_[s] = t.valid ? (((t.n * p.H + oy) * p.W + ox) * (G::COUT / 4) + cb * 8 + g) : -1;  // 16-byte units
        if constexpr (!G::PROJ) {
          if (s == 0) dead = 0;
#pragma unroll
          for (int b = 0; b < (G::B8 ? 2 : 4); ++b) {
            const int blk = G::B8 ? 2 * s + b : b;
            const int row = oy + (G::B8 ? b * 4 : b * 2);
            dead |= (row >= Hv || ox >= Wv) ? (1 << blk) : 0;
          }
        }
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
        ssum[kk] = 0;
        ssq[kk] = 0;
      }
      pending = 0;
      if (RES_PREFETCH) res_prefetch(0);
    };
    // blocks [pending, pending + count) of the finished tile: bias, residual, store, statistics
    // `land`: wait for this wave's LDS-DMA (issued before the call) after the residual loads and BEFORE the stores of the
    // first block, so that no store is older than the wait (the stores then stay in flight across the barrier)
    auto epi_blocks = [&](int count, bool land = false) {
      const int first = pending, last = min(4, pending + count);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        if (blk < first || blk >= last) continue;  // uniform; keeps acc[] statically indexed
        if (G::NCB == 1 && p.out_nchw) {
          // few-channel NCHW output (conv_out: 64 -> 3, weights zero-padded to 32 couts): the real channels are
          // couts 0..3 = quad 0 of the k-group-0 lanes; consecutive lanes = consecutive pixels of a plane
          if (pixoff_of(blk) >= 0 && g == 0) {
            const int pixel = pixoff_of(blk) >> 3;  // cb == 0, g == 0
            const int HW = p.H * p.W;
            const int n = pixel / HW, rem = pixel - n * HW;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < p.Cout) p.out[((size_t)n * p.Cout + c) * HW + rem] = acc[blk][c];
          }
          continue;
        }
        const int po = pixoff_of(blk);
        if (po >= 0) {
          float* op = p.out + (size_t)po * 4;
          // the residual of this block was fetched one step ago (res_prefetch at the end of the previous block / epi_begin)
          if (!RES_PREFETCH && !G::PROJ) res_prefetch(blk);
          if constexpr (G::PROJ) {
            // (stream tail only -- the steady state is proj_step below) the block's skip projection into its accumulators
            ws_for<0, 4>([&](auto bc) {
              if (decltype(bc)::value == blk) {
                ws_for<0, G::PROJ_KS>([&](auto uc) { proj_fetch_unit(blk, uc); });
                ws_for<0, G::PROJ_KS>([&](auto uc) {
                  proj_mfma_unit(acc[blk], uc);
                  __builtin_amdgcn_sched_barrier(0);
                });
              }
            });
          }
          f32x4 v[4];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            v[qd] = (f32x4){acc[blk][4 * qd], acc[blk][4 * qd + 1], acc[blk][4 * qd + 2], acc[blk][4 * qd + 3]};
            if (!G::PROJ) {
              // element by element, and this file is compiled with -fno-slp-vectorize (build.sh): as ONE vector add -- or re-packed by
              // the SLP vectoriser -- these become v_pk_add_f32 (+ v_mov shuffles), and packed fp32 arithmetic in a write-out / staging
              // wave costs the co-resident consumer wave more MFMA issue than the scalar instructions (round 5: the staging's fmas
              // as v_pk_fma_f32 measured 9-13 % SLOWER, profiles/r05x_*; these adds un-packed: profiles/r05y_*, r05z_*).  NOT inline
              // assembly: the hazard recogniser does not look into it, and accumulators fresh out of the fused projection's MFMAs
              // read that way gave batch-dependent frames (11 GPU tests caught the first version of this change).
#pragma unroll
              for (int r = 0; r < 4; ++r) v[qd][r] += rnext[qd][r];
            }
          }
          if (land) {  // (the waits hipcc emits for `rnext` leave the younger LDS-DMA in flight; this one does not)
            WS_WAIT_VM(0);
            land = false;
          }
          float fs = 0.f, fq = 0.f;  // fp32 over the lane's 16 values of this block, fp64 across
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            if (!(WS_ABL & 128) || v[qd][0] == 1.2345e30f) *(f32x4*)(op + 8 * qd) = v[qd];
            fs += (v[qd][0] + v[qd][1]) + (v[qd][2] + v[qd][3]);
            // Sum of squares as an fma chain into its own register, NOT as in-place squares of v: with the squares
            // written over v's registers (`v_mul_f32 v48, v48, v48` right behind the `global_store_dwordx4 v[48:51]`),
            // a slice executed while the other consumer group's MFMAs run on the same SIMD lost one lane's
            // contribution of a block now and then (sum and stored outputs exact, sum of squares short by ~16 values)
            // -- found by tests/test_gpu_tpw.py (tiles_per_wg >= 2), round 2.
            fq = __builtin_fmaf(v[qd][0], v[qd][0], fq);
            fq = __builtin_fmaf(v[qd][1], v[qd][1], fq);
            fq = __builtin_fmaf(v[qd][2], v[qd][2], fq);
            fq = __builtin_fmaf(v[qd][3], v[qd][3], fq);
          }
          if (!G::PROJ && ((dead >> blk) & 1)) {  // outside the valid extent: stored, not counted
            fs = 0.f;
            fq = 0.f;
          }
          if (RES_PREFETCH && blk + 1 < 4) res_prefetch(blk + 1);  // behind the stores: in flight across the barrier, used next step
          const int slot = G::B8 ? (blk >> 1) : 0;
          ssum[slot] += (StatAcc)fs;
          ssq[slot] += (StatAcc)fq;
        }
      }
      if (land) WS_WAIT_VM(0);  // (no block with stores in this call)
      pending = last;
      if (last == 4 && first < 4 && p.out_stats) {
#pragma unroll
        for (int kk = 0; kk < NSTAT; ++kk) {
          const double a = ws_wave_sum_lane63((double)ssum[kk]);
          const double b = ws_wave_sum_lane63((double)ssq[kk]);
          if (lane == 63 && stat_slot[kk] >= 0) {
            double* o = p.out_stats + (size_t)stat_slot[kk] * 2;
            o[0] = a;
            o[1] = b;
          }
        }
      }
    };
    // (PROJ) chunk step S of the write-out: block S's projection (its operands were requested a step ago), the rolling
    // requests for block S + 1, then bias / store / statistics like epi_blocks.  The weight DMA of the step was issued
    // just before: it is older than the 16 rolling loads, so `vmcnt(16)` = "the DMA has landed" (loads return in order).
    auto proj_step = [&](auto sc, bool wnext) __attribute__((always_inline)) {
      if constexpr (G::PROJ) {
        constexpr int blk = decltype(sc)::value;
        const int po = pixoff_of(blk);
        // (hipcc waits for the operands with vmcnt(0) behind the step's LDS-DMA, guide: "while a glds is in flight ...":
        //  the DMA's round trip sits in front of the MFMAs instead of under them.  Hand-counted inline-asm loads with
        //  `s_waitcnt vmcnt(23)` per unit were measured and changed nothing -- 64.1 vs 64.0 ms over the bench window:
        //  the write-out wave is then bound by the issue of its 24 MFMAs and 25 vector-memory instructions between the
        //  other group's, not by that wait -- so the loads stay compiler-tracked.)
        ws_for<0, G::PROJ_KS>([&](auto uc) {
          proj_mfma_unit(acc[blk], uc);
          if constexpr (blk + 1 < 4) proj_fetch_unit(blk + 1, uc);
          __builtin_amdgcn_sched_barrier(0);  // unit by unit (hoisting the weight reads / splits of all units costs 60+ registers)
        });
        if (wnext) {
          if constexpr (blk + 1 < 4)
            WS_WAIT_VM(16);
          else
            WS_WAIT_VM(0);
        }
        if (po >= 0) {
          float* op = p.out + (size_t)po * 4;
          float fs = 0.f, fq = 0.f;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4 v = (f32x4){acc[blk][4 * qd], acc[blk][4 * qd + 1], acc[blk][4 * qd + 2], acc[blk][4 * qd + 3]};
            if (!(WS_ABL & 128) || v[0] == 1.2345e30f) *(f32x4*)(op + 8 * qd) = v;
            fs += (v[0] + v[1]) + (v[2] + v[3]);
            fq = __builtin_fmaf(v[0], v[0], fq);  // (an fma chain into its own register: see epi_blocks)
            fq = __builtin_fmaf(v[1], v[1], fq);
            fq = __builtin_fmaf(v[2], v[2], fq);
            fq = __builtin_fmaf(v[3], v[3], fq);
          }
          ssum[0] += (StatAcc)fs;
          ssq[0] += (StatAcc)fq;
        }
        pending = blk + 1;
        if constexpr (blk == 3) {
          if (p.out_stats) {
            const double a = ws_wave_sum_lane63((double)ssum[0]);
            const double b = ws_wave_sum_lane63((double)ssq[0]);
            if (lane == 63 && stat_slot[0] >= 0) {
              double* o = p.out_stats + (size_t)stat_slot[0] * 2;
              o[0] = a;
              o[1] = b;
            }
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
    auto cons_land_W = [&]() { WS_WAIT_VM(0); };  // the DMA writes have landed before the step's barrier

    // the bias row of the convolution: accumulators START from it (no bias loads / adds in the write-out)
    if (role == 0 && tid < G::COUT)
      bias_lds[tid] = (p.bias ? p.bias[tid] : 0.f) + ((G::PROJ && p.proj_bias) ? p.proj_bias[tid] : 0.f);
    WS_STAMP(role, 14, 0);
    ws_barrier();  // B(-1)
    WS_STAMP(role, 15, 0);
    if (role == 1) {  // group 1 is idle during tile 0: it provides the first chunk's weights
      cons_load_W(0, 0);
      cons_land_W();
    }
    WS_STAMP(role, 12, 0);
    ws_barrier();  // B0
    int j = 0;
    // The two groups take alternate tiles: group `role` computes tiles k = role, role + 2, ... and, while the other group
    // computes tile k + 1, writes tile k out.  The alternation is spelled out structurally (not as `(k & 1) == role`
    // inside one loop): PROJ's write-out requests are inline-asm loads, and on this skeleton every one of them is provably
    // awaited on every path (tools/asm_lint.py follows the control-flow graph).
    // group 1 has nothing to write during tile 0: it only provides the weights of tile 0's chunk steps
    int k = role;
    if (role == 1 && nmy > 0) {
      for (int ck = 0; ck < nchunks; ++ck, ++j) {
        const bool wnext = j + 1 < S;
        WS_STAMP(role, 8, j);
        if (wnext) {
          cons_load_W((j + 1) % nchunks, (j + 1) & 1);
          cons_land_W();
        }
        WS_STAMP(role, 10, j);
        ws_barrier();  // B(j + 1)
        WS_STAMP(role, 11, j);
      }
    }
    for (; k < nmy; k += 2) {
      // ---- this group's tile: fragment reads + MFMAs only ----
      // (only if the other group's tile had too few steps to finish the write-out; PROJ launches have 4 steps per tile)
      if (!G::PROJ && pending < 4) epi_blocks(4);
      WS_PRIO_COMPUTE();
      {
        // lane owns couts cb*32 + 8 qd + 4 g + (0..3), qd = 0..3, of its pixel of each block
        f32x4 bq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bq[qd] = *(const f32x4*)(bias_lds + cb * 32 + 8 * qd + 4 * g);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[blk][r] = bq[r >> 2][r & 3];
      }
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
        WS_STAMP(role, 4, j);
        ws_chunk_body<G>(acc, a, b, lds, ad);
        WS_STAMP(role, 5, j);
        ws_barrier();  // B(j + 1): every fragment of buffer j is in registers; buffer j + 1 is complete
        WS_STAMP(role, 6, j);
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
      if (k + 1 >= nmy) break;
      WS_PRIO_WRITEOUT();
      // ---- the other group's tile: write our finished tile out, a slice per chunk step, and move the weights ----
      if constexpr (G::PROJ) {
        // 4 chunk steps per tile (eligibility), block s of the finished tile in step s: straight-line code, so that
        // hipcc sees which accumulators are dead and counts the loads in flight exactly
        ws_for<0, 4>([&](auto sc) {
          const bool wnext = j + 1 < S;
          WS_STAMP(role, 8, j);
          // block 0's operands: requested here, not in epi_begin -- nothing separates the two in time (the tile's last
          // MFMAs are right in front of this step), and every request and its wait then sit in one straight line
          if constexpr (decltype(sc)::value == 0) ws_for<0, G::PROJ_KS>([&](auto uc) { proj_fetch_unit(0, uc); });
          if (wnext) cons_load_W((j + 1) % nchunks, (j + 1) & 1);
          WS_STAMP(role, 9, j);
          proj_step(sc, wnext);  // (a finished tile is always waiting here: epi_begin was the previous statement)
          WS_STAMP(role, 10, j);
          ws_barrier();  // B(j + 1)
          WS_STAMP(role, 11, j);
          ++j;
        });
      } else {
        for (int ck = 0; ck < nchunks; ++ck, ++j) {
          const bool wnext = j + 1 < S;  // this (idle) group copies the next step's weights
          WS_STAMP(role, 8, j);
          if (wnext) cons_load_W((j + 1) % nchunks, (j + 1) & 1);
          WS_STAMP(role, 9, j);
          if (pending < 4)
            epi_blocks(blocks_per_step, wnext);
          else if (wnext)
            cons_land_W();
          WS_STAMP(role, 10, j);
          ws_barrier();  // B(j + 1)
          WS_STAMP(role, 11, j);
        }
      }
    }
    WS_STAMP(role, 13, 1);
    if (pending < 4) epi_blocks(4);  // tail: the last tile(s) of the range
    WS_STAMP(role, 13, 2);
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

extern "C" int dmd_conv2d_proj_eligible(const dmd_conv_params* p);
int dmd_launch_conv_f16ws(const dmd_conv_params& p, hipStream_t st) {
  const bool b8 = p.W % 16 != 0;
  const int sub8 = p.N * (p.H / 8) * (p.W / 8), t16 = p.N * (p.H / 16) * (p.W / 16);
  if (p.proj_nsrc) {
    DMD_CHECK_ARG(dmd_conv2d_proj_eligible(&p), "conv: the fused skip projection needs a split-fp16 3x3 stride-1 launch with Cout 64, "
                  "H, W multiples of 16, two 64-channel projection sources and no other residual (dmd_conv2d_proj_eligible)");
    return launch_f16ws<WsGeomProj>(p, t16, st);
  }
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
  // the producers address a source as base + 32-bit byte offset formed by a 24-bit multiply (pixel index x C * 4)
  const long long src_pixels = (long long)p->N * (p->H >> p->upsample) * (p->W >> p->upsample);
  if (src_pixels >= (1ll << 24)) return 0;
  for (int i = 0; i < p->nsrc; ++i)
    if (src_pixels * p->src[i].C * 4 >= (1ll << 32)) return 0;
  const bool a16 = p->H % 16 == 0 && p->W % 16 == 0;
  const bool b8 = p->W % 16 != 0;
  return (a16 || b8) ? 1 : 0;
}

// 1: the proj_* fields of these parameters can be fused into the launch (WsGeomProj)
extern "C" int dmd_conv2d_proj_eligible(const dmd_conv_params* p) {
  if (!p || !dmd_conv2d_f16x2_eligible(p)) return 0;
  if (p->taps != 9 || p->upsample || p->Cout != 64 || p->CoutPad != 64 || p->out_nchw || p->residual) return 0;
  if (p->H % 16 != 0 || p->W % 16 != 0 || p->valid_h || p->valid_w) return 0;
  // 4 chunk steps per tile = one 32-pixel block of the finished tile per step (the kernel has no catch-up path)
  if (p->nsrc != 1 || p->src[0].C != 64) return 0;
  if ((long long)p->N * p->H * p->W * 256 >= (1ll << 32)) return 0;  // 32-bit byte offsets into the projection sources
  if (p->proj_nsrc != 2 || !p->proj_w_f16 || !p->proj_x[0] || !p->proj_x[1] || p->proj_C[0] != 64 || p->proj_C[1] != 64) return 0;
  return 1;
}
