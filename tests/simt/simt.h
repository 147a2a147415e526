// TEST INFRASTRUCTURE -- never part of the product.
//
// A small SIMT interpreter for the host: the kernels of diamond_amd/csrc/*.hip, compiled UNCHANGED as host C++ (the shim
// tests/simt/include/hip/hip_runtime.h stands in for the HIP headers), run one workgroup at a time with one FIBER per lane.
// Wave-level operations (MFMA, DPP, readlane, readfirstlane, __shfl_xor) and __syncthreads / s_barrier are rendezvous points: a lane deposits its
// operands and yields; the last lane of the wave to arrive computes the operation for all 64 and everybody resumes.  That
// executes the kernels' real index arithmetic, LDS layouts, synchronisation structure and MFMA operand layouts on a CPU,
// so `pytest -m "not gpu"` can hold them against the oracle; it says nothing about speed, memory ordering hazards or the
// hardware's accumulation order inside an MFMA (results agree with the GPU to rounding, not bit for bit).
//
// Only tests/ builds or loads this (tests/simt/build.sh -> tests/simt/_build/libdiamond_simt.so); diamond_amd/native.py
// loads libdiamond_hip.so and nothing else.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

namespace simt {

struct u3 {
  unsigned x, y, z;
};

struct Lane {
  u3 tid;
  int lane;  // 0..63 within its wave
  int wave;
};

struct Block {
  u3 bid, bdim, gdim;
  unsigned char* dyn_lds;
};

extern thread_local Lane* g_lane;
extern thread_local Block* g_block;

// __syncthreads
void barrier();

// One wave-level operation.  `fn` sees the operands of all 64 lanes (`in + lane * in_stride`; lanes that have left the
// kernel hold zeros and are clear in `live`) and writes every lane's result.  `imm` are the instruction's immediates: they
// and `opcode` must agree across the lanes of a wave (a divergent call site aborts with a diagnostic).
typedef void (*wave_fn)(const unsigned char* in, size_t in_stride, unsigned char* out, size_t out_stride, uint64_t live,
                        const int* imm);
void wave_op(int opcode, const void* in, size_t in_bytes, void* out, size_t out_bytes, wave_fn fn, const int* imm, int n_imm);

// blocks run one after the other on up to SIMT_THREADS host threads (default: the hardware concurrency, at most 8)
void launch(u3 grid, u3 block, size_t dyn_lds_bytes, const std::function<void()>& body);

enum { OP_SHFL_XOR = 1, OP_READLANE, OP_DPP, OP_MFMA_16x16x4_F32, OP_MFMA_16x16x16_F16, OP_MFMA_16x16x32_F16, OP_MFMA_32x32x16_F16,
       OP_MFMA_32x32x8_F16, OP_BALLOT, OP_READFIRSTLANE };

// ---- the wave-level operations the kernels use ---------------------------------------------------------------------------------
void fn_shfl_xor(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);
void fn_readlane(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);
void fn_dpp(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);
void fn_readfirstlane(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);
void fn_mfma_16x16x4_f32(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);
void fn_mfma_16x16xK_f16(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);
void fn_mfma_32x32xK_f16(const unsigned char*, size_t, unsigned char*, size_t, uint64_t, const int*);

template <class T>
inline T shfl_xor(T v, int mask, int width = 64) {
  static_assert(sizeof(T) <= 8, "shfl_xor: up to 8 bytes");
  const int imm[3] = {mask, width, (int)sizeof(T)};
  T r;
  wave_op(OP_SHFL_XOR, &v, sizeof(T), &r, sizeof(T), fn_shfl_xor, imm, 3);
  return r;
}

inline int readlane(int v, int src) {
  const int imm[1] = {src};
  int r;
  wave_op(OP_READLANE, &v, 4, &r, 4, fn_readlane, imm, 1);
  return r;
}

// v_readfirstlane_b32: the value of the lowest lane still in the kernel
template <class T>
inline T readfirstlane(T v) {
  static_assert(sizeof(T) == 4, "readfirstlane: 32-bit values");
  T r;
  wave_op(OP_READFIRSTLANE, &v, 4, &r, 4, fn_readfirstlane, nullptr, 0);
  return r;
}

// global_load_lds (LDS-DMA): every lane moves `size` bytes from ITS global address to the wave's LDS base + lane * size (+ offset);
// synchronous here (on the device it is in flight until the issuing wave's vmcnt says otherwise)
inline void global_load_lds(const void* gptr, void* lds_base, int size, int offset) {
  memcpy((unsigned char*)lds_base + offset + (size_t)g_lane->lane * size, gptr, (size_t)size);
}

inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int imm[4] = {ctrl, row_mask, bank_mask, bound_ctrl ? 1 : 0};
  const int in[2] = {old, src};
  int r;
  wave_op(OP_DPP, in, 8, &r, 4, fn_dpp, imm, 4);
  return r;
}

// v_perm_b32 D = perm({S0, S1}, sel): selector 0-3 = byte of S1, 4-7 = byte of S0, 0x0c = 0x00, >= 0x0d = 0xff
inline unsigned perm(unsigned s0, unsigned s1, unsigned sel) {
  const uint64_t v = ((uint64_t)s0 << 32) | s1;
  unsigned d = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned s = (sel >> (8 * i)) & 0xff;
    unsigned b;
    if (s < 8) b = (unsigned)(v >> (8 * s)) & 0xff;
    else if (s == 0x0c) b = 0;
    else if (s >= 0x0d) b = 0xff;
    else {
      fprintf(stderr, "simt: v_perm_b32 selector 0x%x is not modelled\n", s);
      abort();
    }
    d |= b << (8 * i);
  }
  return d;
}

// MFMA: A (8 or 16 bytes of fp16, or one float), B likewise, C / D 4 or 16 floats per lane
template <class A, class C>
inline C mfma(int opcode, wave_fn fn, int k, A a, A b, C c) {
  struct {
    unsigned char a[16], b[16];
    float c[16];
  } in;
  static_assert(sizeof(A) <= 16 && sizeof(C) <= 64, "mfma operand sizes");
  memset(&in, 0, sizeof(in));
  memcpy(in.a, &a, sizeof(A));
  memcpy(in.b, &b, sizeof(A));
  memcpy(in.c, &c, sizeof(C));
  const int imm[3] = {k, (int)sizeof(A), (int)sizeof(C)};
  C d;
  wave_op(opcode, &in, sizeof(in), &d, sizeof(C), fn, imm, 3);
  return d;
}
template <class A, class C>
inline C mfma_f32_16x16x4f32(A a, A b, C c, int, int, int) {
  static_assert(sizeof(A) == 4 && sizeof(C) == 16, "v_mfma_f32_16x16x4_f32 operands");
  return mfma(OP_MFMA_16x16x4_F32, fn_mfma_16x16x4_f32, 4, a, b, c);
}
template <class A, class C>
inline C mfma_f32_16x16x16f16(A a, A b, C c, int, int, int) {
  static_assert(sizeof(A) == 8 && sizeof(C) == 16, "v_mfma_f32_16x16x16_f16 operands");
  return mfma(OP_MFMA_16x16x16_F16, fn_mfma_16x16xK_f16, 16, a, b, c);
}
template <class A, class C>
inline C mfma_f32_16x16x32_f16(A a, A b, C c, int, int, int) {
  static_assert(sizeof(A) == 16 && sizeof(C) == 16, "v_mfma_f32_16x16x32_f16 operands");
  return mfma(OP_MFMA_16x16x32_F16, fn_mfma_16x16xK_f16, 32, a, b, c);
}
template <class A, class C>
inline C mfma_f32_32x32x16_f16(A a, A b, C c, int, int, int) {
  static_assert(sizeof(A) == 16 && sizeof(C) == 64, "v_mfma_f32_32x32x16_f16 operands");
  return mfma(OP_MFMA_32x32x16_F16, fn_mfma_32x32xK_f16, 16, a, b, c);
}
template <class A, class C>
inline C mfma_f32_32x32x8f16(A a, A b, C c, int, int, int) {
  static_assert(sizeof(A) == 8 && sizeof(C) == 64, "v_mfma_f32_32x32x8_f16 operands");
  return mfma(OP_MFMA_32x32x8_F16, fn_mfma_32x32xK_f16, 8, a, b, c);
}

}  // namespace simt
