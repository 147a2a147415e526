// TEST INFRASTRUCTURE (see simt.h): fibers, the workgroup scheduler and the wave-level operations.
#include "simt.h"

#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace simt {

thread_local Lane* g_lane = nullptr;
thread_local Block* g_block = nullptr;

// ---- context switch (x86-64 System V: callee-saved registers + stack pointer) ----------------------------------------------------
extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size simt_switch,.-simt_switch
)");

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_LANES = 1024;
constexpr size_t SLOT = 128;  // bytes of operand / result staging per lane

enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Fiber {
  Lane lane;
  void* sp = nullptr;
  State state = DONE;
};

struct Wave {
  int arrived = 0, live = 0;
  int opcode = 0, n_imm = 0, in_bytes = 0, out_bytes = 0;
  int imm[4] = {0, 0, 0, 0};
  wave_fn fn = nullptr;
  alignas(16) unsigned char in[64 * SLOT];
  alignas(16) unsigned char out[64 * SLOT];
  uint64_t arrived_mask = 0;
};

struct Worker {
  unsigned char* stacks = nullptr;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  void* sched_sp = nullptr;
  Fiber* cur = nullptr;
  const std::function<void()>* body = nullptr;
  int nlanes = 0, nwaves = 0, done = 0;
  // dynamic LDS: the block ENDS at an inaccessible page, so a kernel that reaches beyond the bytes its launch asked for faults
  // here instead of reading / writing a neighbour's data unnoticed (on the device: another workgroup's LDS, or a memory fault)
  unsigned long long n_barriers = 0, n_ops[16] = {};  // SIMT_STATS: wave-level barriers / operations by opcode, of this worker
  unsigned char* lds_map = nullptr;
  static constexpr size_t LDS_MAP = 192 * 1024, PAGE = 4096;
  Block block;
  ~Worker() {
    if (stacks) munmap(stacks, STACK_BYTES * MAX_LANES);
    if (lds_map) munmap(lds_map, LDS_MAP + PAGE);
  }
};

thread_local Worker* tl_worker = nullptr;

[[noreturn]] void die(const char* msg) {
  Worker* w = tl_worker;
  if (w && w->cur)
    fprintf(stderr, "simt: %s (block %u,%u,%u thread %u lane %d of wave %d)\n", msg, w->block.bid.x, w->block.bid.y, w->block.bid.z,
            w->cur->lane.tid.x, w->cur->lane.lane, w->cur->lane.wave);
  else
    fprintf(stderr, "simt: %s\n", msg);
  abort();
}

void yield_to_scheduler() {
  Worker* w = tl_worker;
  Fiber* f = w->cur;
  simt_switch(&f->sp, w->sched_sp);
}

void fiber_main() {
  Worker* w = tl_worker;
  (*w->body)();
  Fiber* f = w->cur;
  f->state = DONE;
  yield_to_scheduler();
  die("a finished lane was resumed");
}

void complete_wave_op(Worker* w, Wave& wv, int wave_index) {
  ++w->n_ops[wv.opcode & 15];
  uint64_t live = wv.arrived_mask;
  wv.fn(wv.in, SLOT, wv.out, SLOT, live, wv.imm);
  wv.arrived = 0;
  wv.arrived_mask = 0;
  for (int l = 0; l < 64; ++l) {
    const int t = wave_index * 64 + l;
    if (t < w->nlanes && w->fibers[t].state == WAIT_WAVE) w->fibers[t].state = RUNNABLE;
  }
}

void prepare_fiber(Worker* w, int t) {
  Fiber& f = w->fibers[t];
  unsigned char* top = w->stacks + (size_t)(t + 1) * STACK_BYTES;
  void** sp = (void**)top;
  *--sp = nullptr;              // the entry function's "return address" (never used)
  *--sp = (void*)&fiber_main;   // popped by simt_switch's ret
  for (int i = 0; i < 6; ++i) *--sp = nullptr;
  f.sp = sp;
  f.state = RUNNABLE;
}

void run_block(Worker* w, u3 bid, u3 bdim, u3 gdim, size_t dyn_lds, const std::function<void()>& body) {
  const int n = (int)(bdim.x * bdim.y * bdim.z);
  if (n > MAX_LANES) die("more than 1024 threads per block");
  if (!w->stacks) {
    w->stacks = (unsigned char*)mmap(nullptr, STACK_BYTES * MAX_LANES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w->stacks == (unsigned char*)MAP_FAILED) die("mmap of the lane stacks failed");
    w->fibers.resize(MAX_LANES);
    w->waves.resize(MAX_LANES / 64);
  }
  if (!w->lds_map) {
    w->lds_map = (unsigned char*)mmap(nullptr, Worker::LDS_MAP + Worker::PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (w->lds_map == (unsigned char*)MAP_FAILED) die("mmap of the LDS block failed");
    if (mprotect(w->lds_map + Worker::LDS_MAP, Worker::PAGE, PROT_NONE) != 0) die("mprotect of the LDS guard page failed");
  }
  if (dyn_lds > 160 * 1024) die("a launch asks for more than 160 KiB of dynamic LDS");
  w->block.bid = bid;
  w->block.bdim = bdim;
  w->block.gdim = gdim;
  w->block.dyn_lds = w->lds_map + Worker::LDS_MAP - ((dyn_lds + 15) & ~(size_t)15);  // 16-byte aligned, ends at the guard page
  memset(w->block.dyn_lds, 0xff, dyn_lds);  // poison: whatever a kernel reads before it has written it is a NaN (fp32 / fp16 alike)
  w->body = &body;
  w->nlanes = n;
  w->nwaves = (n + 63) / 64;
  w->done = 0;
  g_block = &w->block;
  for (int t = 0; t < n; ++t) {
    Fiber& f = w->fibers[t];
    f.lane.tid = {(unsigned)t % bdim.x, ((unsigned)t / bdim.x) % bdim.y, (unsigned)t / (bdim.x * bdim.y)};
    f.lane.lane = t % 64;
    f.lane.wave = t / 64;
    prepare_fiber(w, t);
  }
  for (int v = 0; v < w->nwaves; ++v) {
    Wave& wv = w->waves[v];
    wv.arrived = 0;
    wv.arrived_mask = 0;
    wv.live = (v == w->nwaves - 1 && n % 64) ? n % 64 : 64;
    memset(wv.in, 0, sizeof(wv.in));
  }
  // SIMT_SCHEDULE: the order in which runnable waves / lanes are resumed (0: ascending, 1: descending, 2: odd waves first).  A
  // correctly synchronised kernel computes the same bits under every order; tests run a kernel under several and compare.
  int schedule = 0;
  if (const char* e = getenv("SIMT_SCHEDULE")) schedule = atoi(e);
  while (w->done < n) {
    bool progressed = false;
    for (int vi = 0; vi < w->nwaves; ++vi) {
      int v = vi;
      if (schedule == 1) v = w->nwaves - 1 - vi;
      else if (schedule == 2) v = (2 * vi + 1 < w->nwaves) ? 2 * vi + 1 : 2 * (vi - w->nwaves / 2);
      bool again = true;
      while (again) {
        again = false;
        for (int li = 0; li < 64; ++li) {
          const int l = schedule == 1 ? 63 - li : li;
          const int t = v * 64 + l;
          if (t >= n) continue;
          Fiber& f = w->fibers[t];
          if (f.state != RUNNABLE) continue;
          w->cur = &f;
          g_lane = &f.lane;
          simt_switch(&w->sched_sp, f.sp);
          progressed = true;
          if (f.state == DONE) {
            ++w->done;
            Wave& wv = w->waves[v];
            --wv.live;
            memset(wv.in + (size_t)l * SLOT, 0, SLOT);  // a lane that has left the kernel contributes zeros from now on
            // the lanes still in the kernel may all be waiting in a wave operation already
            if (wv.live > 0 && wv.arrived == wv.live) {
              complete_wave_op(w, wv, v);
              again = true;
            }
          } else if (f.state == RUNNABLE) {
            again = true;
          }
        }
        for (int l = 0; l < 64 && !again; ++l) {
          const int t = v * 64 + l;
          if (t < n && w->fibers[t].state == RUNNABLE) again = true;
        }
      }
    }
    // workgroup barrier: everybody still in the kernel waits at it
    int waiting = 0;
    for (int t = 0; t < n; ++t) waiting += w->fibers[t].state == WAIT_BLOCK;
    if (waiting && waiting == n - w->done) {
      ++w->n_barriers;
      for (int t = 0; t < n; ++t)
        if (w->fibers[t].state == WAIT_BLOCK) w->fibers[t].state = RUNNABLE;
      progressed = true;
    }
    if (!progressed) {
      int ww = 0, wb = 0;
      for (int t = 0; t < n; ++t) {
        ww += w->fibers[t].state == WAIT_WAVE;
        wb += w->fibers[t].state == WAIT_BLOCK;
      }
      fprintf(stderr, "simt: deadlock in block %u,%u,%u: %d lanes in a wave operation, %d at __syncthreads, %d finished of %d\n", bid.x, bid.y,
              bid.z, ww, wb, w->done, n);
      for (int v = 0; v < w->nwaves; ++v)
        fprintf(stderr, "  wave %d: live %d arrived %d opcode %d\n", v, w->waves[v].live, w->waves[v].arrived, w->waves[v].opcode);
      abort();
    }
  }
  w->cur = nullptr;
  g_lane = nullptr;
}

}  // namespace

void barrier() {
  Worker* w = tl_worker;
  Fiber* f = w->cur;
  Wave& wv = w->waves[f->lane.wave];
  if (wv.arrived) die("__syncthreads while other lanes of the wave wait in a wave-level operation (divergent control flow)");
  f->state = WAIT_BLOCK;
  yield_to_scheduler();
}

void wave_op(int opcode, const void* in, size_t in_bytes, void* out, size_t out_bytes, wave_fn fn, const int* imm, int n_imm) {
  Worker* w = tl_worker;
  Fiber* f = w->cur;
  Wave& wv = w->waves[f->lane.wave];
  if (in_bytes > SLOT || out_bytes > SLOT || n_imm > 4) die("wave operation: operand too large for the staging slot");
  if (wv.arrived == 0) {
    wv.opcode = opcode;
    wv.fn = fn;
    wv.n_imm = n_imm;
    wv.in_bytes = (int)in_bytes;
    wv.out_bytes = (int)out_bytes;
    for (int i = 0; i < n_imm; ++i) wv.imm[i] = imm[i];
  } else {
    bool same = wv.opcode == opcode && wv.n_imm == n_imm && wv.in_bytes == (int)in_bytes;
    for (int i = 0; same && i < n_imm; ++i) same = wv.imm[i] == imm[i];
    if (!same) die("the lanes of a wave reached DIFFERENT wave-level operations (divergent control flow around an MFMA / shuffle)");
  }
  memcpy(wv.in + (size_t)f->lane.lane * SLOT, in, in_bytes);
  wv.arrived_mask |= 1ull << f->lane.lane;
  ++wv.arrived;
  if (wv.arrived == wv.live) {
    complete_wave_op(w, wv, f->lane.wave);
    f->state = RUNNABLE;
  } else {
    f->state = WAIT_WAVE;
    yield_to_scheduler();
  }
  memcpy(out, wv.out + (size_t)f->lane.lane * SLOT, out_bytes);
}

// Persistent worker threads: a launch hands them (grid, block, body) and waits; lane stacks and the thread_local LDS arrays of
// the kernels are set up once per worker, not once per launch.
namespace {
std::atomic<unsigned long long> g_barriers{0}, g_ops[16];
struct Pool {
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> threads;
  uint64_t generation = 0;
  int active = 0, wanted = 0;
  bool stop = false;
  // the job
  u3 grid, block;
  size_t lds = 0, nblocks = 0;
  const std::function<void()>* body = nullptr;
  std::atomic<size_t> next{0};

  void work(int index) {
    static thread_local Worker worker;
    tl_worker = &worker;
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || (generation != seen && index < wanted); });
        if (stop) return;
        seen = generation;
      }
      for (;;) {
        const size_t b = next.fetch_add(1);
        if (b >= nblocks) break;
        const u3 bid = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))};
        run_block(&worker, bid, block, grid, lds, *body);
      }
      g_barriers += worker.n_barriers;
      worker.n_barriers = 0;
      for (int i = 0; i < 16; ++i) {
        g_ops[i] += worker.n_ops[i];
        worker.n_ops[i] = 0;
      }
      std::lock_guard<std::mutex> lk(m);
      if (--active == 0) cv_done.notify_all();
    }
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_job.notify_all();
    for (auto& t : threads) t.join();
  }
};
Pool& pool() {
  static Pool* p = new Pool;  // never destroyed: worker threads must not be joined from a static destructor at exit
  return *p;
}
std::mutex launch_mutex;  // one launch at a time (a stream)
}  // namespace

void launch(u3 grid, u3 block, size_t dyn_lds_bytes, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (!nblocks) return;
  int nthreads = 0;
  if (const char* e = getenv("SIMT_THREADS")) nthreads = atoi(e);
  if (nthreads <= 0) {
    nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads > 8) nthreads = 8;
    if (nthreads < 1) nthreads = 1;
  }
  if ((size_t)nthreads > nblocks) nthreads = (int)nblocks;
  std::lock_guard<std::mutex> one(launch_mutex);
  Pool& P = pool();
  std::unique_lock<std::mutex> lk(P.m);
  while ((int)P.threads.size() < nthreads) {
    const int index = (int)P.threads.size();
    P.threads.emplace_back([&P, index] { P.work(index); });
  }
  P.grid = grid;
  P.block = block;
  P.lds = dyn_lds_bytes;
  P.nblocks = nblocks;
  P.body = &body;
  P.next.store(0);
  P.wanted = nthreads;
  P.active = nthreads;
  ++P.generation;
  P.cv_job.notify_all();
  P.cv_done.wait(lk, [&] { return P.active == 0; });
  P.wanted = 0;
  if (getenv("SIMT_STATS")) {  // per launch: workgroup barriers and wave-level operations (by opcode, simt.h) actually executed
    fprintf(stderr, "simt: launch grid %u x %u x %u, block %u: %llu workgroup barriers; wave operations:", grid.x, grid.y, grid.z,
            block.x * block.y * block.z, (unsigned long long)g_barriers.exchange(0));
    static const char* names[16] = {"", "shfl_xor", "readlane", "dpp", "mfma16x16x4f32", "mfma16x16x16f16", "mfma16x16x32f16", "mfma32x32x16f16",
                                    "mfma32x32x8f16", "ballot", "readfirstlane", "", "", "", "", ""};
    for (int i = 1; i < 16; ++i) {
      const unsigned long long v = g_ops[i].exchange(0);
      if (v) fprintf(stderr, " %s %llu", names[i], v);
    }
    fprintf(stderr, "\n");
  } else {
    g_barriers = 0;
    for (int i = 0; i < 16; ++i) g_ops[i] = 0;
  }
}

// ---- wave-level operations -----------------------------------------------------------------------------------------------------
void fn_shfl_xor(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t, const int* imm) {
  const int mask = imm[0], width = imm[1], bytes = imm[2];
  for (int l = 0; l < 64; ++l) {
    int src = l ^ mask;
    if (src / width != l / width || src >= 64) src = l;  // out of the segment: own value
    memcpy(out + l * os, in + src * is, bytes);
  }
}

void fn_readfirstlane(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t live, const int*) {
  int first = 0;
  while (first < 63 && !((live >> first) & 1)) ++first;
  for (int l = 0; l < 64; ++l) memcpy(out + l * os, in + first * is, 4);
}

void fn_readlane(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t, const int* imm) {
  for (int l = 0; l < 64; ++l) memcpy(out + l * os, in + (imm[0] & 63) * is, 4);
}

// v_mov_b32_dpp (GFX9 controls).  in = {old, src}; a lane whose row / bank is masked off, or whose source lane does not exist
// (bound_ctrl clear), keeps `old`.
void fn_dpp(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t, const int* imm) {
  const int ctrl = imm[0], row_mask = imm[1], bank_mask = imm[2], bound_ctrl = imm[3];
  for (int l = 0; l < 64; ++l) {
    int old, v;
    memcpy(&old, in + l * is, 4);
    const int row = l / 16, rl = l % 16, bank = rl / 4;
    int src = -1;  // -1: no source lane
    if (ctrl >= 0x00 && ctrl <= 0xff) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = rl + (ctrl & 15); src = s < 16 ? row * 16 + s : -1; }       // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = rl - (ctrl & 15); src = s >= 0 ? row * 16 + s : -1; }       // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12f) src = row * 16 + ((rl - (ctrl & 15)) & 15);                                   // row_ror
    else if (ctrl == 0x140) src = row * 16 + (15 - rl);                                                                   // row_mirror
    else if (ctrl == 0x141) src = row * 16 + (rl & 8) + (7 - (rl & 7));                                                    // row_half_mirror
    else if (ctrl == 0x142) src = row > 0 ? (row - 1) * 16 + 15 : -1;                                                      // row_bcast:15
    else if (ctrl == 0x143) src = row >= 2 ? 31 : -1;                                                                      // row_bcast:31
    else {
      fprintf(stderr, "simt: dpp_ctrl 0x%x is not modelled\n", ctrl);
      abort();
    }
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> bank) & 1);
    if (!enabled) v = old;
    else if (src < 0) v = bound_ctrl ? 0 : old;
    else memcpy(&v, in + src * is + 4, 4);
    memcpy(out + l * os, &v, 4);
  }
}

namespace {
struct MfmaIn {
  unsigned char a[16], b[16];
  float c[16];
};
inline const MfmaIn* lane_in(const unsigned char* in, size_t is, int l) { return (const MfmaIn*)(in + l * is); }
}  // namespace

// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], k ascending (one fp32 fma per term).  The loops run i, k, j with j innermost so
// that the compiler vectorises over the columns; per element the order of the sum is unchanged.
namespace {
template <int M, int N, int K>
inline void mma(const float (&A)[M][K], const float (&B)[K][N], float (&D)[M][N]) {
  for (int i = 0; i < M; ++i)
    for (int k = 0; k < K; ++k) {
      const float a = A[i][k];
      for (int j = 0; j < N; ++j) D[i][j] = __builtin_fmaf(a, B[k][j], D[i][j]);
    }
}
}  // namespace

// v_mfma_f32_16x16x4_f32: lane l holds A[l % 16][l / 16], B[l / 16][l % 16], D[4 (l / 16) + r][l % 16] in register r
void fn_mfma_16x16x4_f32(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t, const int*) {
  float A[16][4], B[4][16], D[16][16];
  for (int l = 0; l < 64; ++l) {
    memcpy(&A[l % 16][l / 16], lane_in(in, is, l)->a, 4);
    memcpy(&B[l / 16][l % 16], lane_in(in, is, l)->b, 4);
    for (int r = 0; r < 4; ++r) D[4 * (l / 16) + r][l % 16] = lane_in(in, is, l)->c[r];
  }
  mma<16, 16, 4>(A, B, D);
  for (int l = 0; l < 64; ++l) {
    float d[4];
    for (int r = 0; r < 4; ++r) d[r] = D[4 * (l / 16) + r][l % 16];
    memcpy(out + l * os, d, 16);
  }
}

// K = 16 (4 halves per lane) or 32 (8 per lane): lane l holds A[l % 16][KV (l / 16) + v], B[KV (l / 16) + v][l % 16]
void fn_mfma_16x16xK_f16(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t, const int* imm) {
  const int K = imm[0], KV = K / 4;
  float A[16][32] = {}, B[32][16] = {}, D[16][16];
  for (int l = 0; l < 64; ++l) {
    const _Float16* a = (const _Float16*)lane_in(in, is, l)->a;
    const _Float16* b = (const _Float16*)lane_in(in, is, l)->b;
    for (int v = 0; v < KV; ++v) {
      A[l % 16][KV * (l / 16) + v] = (float)a[v];
      B[KV * (l / 16) + v][l % 16] = (float)b[v];
    }
    for (int r = 0; r < 4; ++r) D[4 * (l / 16) + r][l % 16] = lane_in(in, is, l)->c[r];
  }
  if (K == 32)
    mma<16, 16, 32>(A, B, D);
  else {  // the upper half of the K range is zero: stop at 16 (adding +0 products would not change a sum, this is for speed)
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < 16; ++k) {
        const float a = A[i][k];
        for (int j = 0; j < 16; ++j) D[i][j] = __builtin_fmaf(a, B[k][j], D[i][j]);
      }
  }
  for (int l = 0; l < 64; ++l) {
    float d[4];
    for (int r = 0; r < 4; ++r) d[r] = D[4 * (l / 16) + r][l % 16];
    memcpy(out + l * os, d, 16);
  }
}

// 32x32: lane l holds A[l % 32][KV (l / 32) + v], B[KV (l / 32) + v][l % 32] (KV = K / 2), D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32] in register r
void fn_mfma_32x32xK_f16(const unsigned char* in, size_t is, unsigned char* out, size_t os, uint64_t, const int* imm) {
  const int K = imm[0], KV = K / 2;
  float A[32][16] = {}, B[16][32] = {}, D[32][32];
  for (int l = 0; l < 64; ++l) {
    const _Float16* a = (const _Float16*)lane_in(in, is, l)->a;
    const _Float16* b = (const _Float16*)lane_in(in, is, l)->b;
    for (int v = 0; v < KV; ++v) {
      A[l % 32][KV * (l / 32) + v] = (float)a[v];
      B[KV * (l / 32) + v][l % 32] = (float)b[v];
    }
    for (int r = 0; r < 16; ++r) D[8 * (r / 4) + 4 * (l / 32) + r % 4][l % 32] = lane_in(in, is, l)->c[r];
  }
  if (K == 16)
    mma<32, 32, 16>(A, B, D);
  else {
    for (int i = 0; i < 32; ++i)
      for (int k = 0; k < K; ++k) {
        const float a = A[i][k];
        for (int j = 0; j < 32; ++j) D[i][j] = __builtin_fmaf(a, B[k][j], D[i][j]);
      }
  }
  for (int l = 0; l < 64; ++l) {
    float d[16];
    for (int r = 0; r < 16; ++r) d[r] = D[8 * (r / 4) + 4 * (l / 32) + r % 4][l % 32];
    memcpy(out + l * os, d, 64);
  }
}

}  // namespace simt
