/* tests/simt_emu/simt_emu.cpp — fiber scheduler + trivial host runtime of the SIMT interpreter.
 * See hip/hip_runtime.h in this directory for what this is (and is not). */
#include "hip/hip_runtime.h"
#include <mutex>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {

enum State { RUNNABLE, AT_BLOCK_BARRIER, AT_WAVE_BARRIER, DONE };

/* Context switch.  glibc's swapcontext / getcontext make one rt_sigprocmask system call each, which was a third of the
 * CPU tier's run time; on x86-64 the switch is six callee-saved registers and the stack pointer, so it is written out here.
 * Other hosts, and builds with -DSIMT_EMU_UCONTEXT (the AddressSanitizer build of tools/fuzz_asan.py: ASan follows
 * swapcontext, not a hand-made switch), use ucontext. */
#if defined(__x86_64__) && !defined(SIMT_EMU_UCONTEXT) && !defined(__SANITIZE_ADDRESS__)
#define SIMT_EMU_ASM_SWITCH 1
extern "C" void simt_emu_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".globl simt_emu_switch\n"
    ".type simt_emu_switch,@function\n"
    "simt_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size simt_emu_switch,.-simt_emu_switch\n");
struct Context { void* sp = nullptr; };
inline void ctx_switch(Context& from, Context& to) { simt_emu_switch(&from.sp, to.sp); }
inline void ctx_make(Context& c, char* stack, size_t size, void (*entry)())
{
  /* what simt_emu_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into entry with rsp = 8 (mod 16) as after a call */
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  void** sp = (void**)(top - 64);
  for (int i = 0; i < 6; i++) sp[i] = nullptr;
  sp[6] = (void*)entry;
  sp[7] = nullptr;                 /* entry never returns */
  c.sp = sp;
}
#else
struct Context { ucontext_t uc; };
inline void ctx_switch(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
inline void ctx_make(Context& c, char* stack, size_t size, void (*entry)())
{
  getcontext(&c.uc);
  c.uc.uc_stack.ss_sp = stack;
  c.uc.uc_stack.ss_size = size;
  c.uc.uc_link = nullptr;
  makecontext(&c.uc, entry, 0);
}
#endif

struct Fiber {
  Context ctx;
  char* stack = nullptr;
  State state = RUNNABLE;
  uint3 tid;
  int wave = 0, lane = 0;
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<unsigned long long> slots;   /* 64 per wave */
  Context sched;
  int current = -1;
  const std::function<void()>* body = nullptr;
};

Block* g_blk = nullptr;
const size_t kStack = 256 * 1024;
/* fiber stacks live as long as the launching host thread: a fresh 256 KB allocation per fiber and launch is an mmap /
 * page-fault / munmap cycle */
struct Stacks {
  std::vector<char*> v;
  ~Stacks() { for (char* p : v) free(p); }
};
thread_local Stacks g_stacks;
hipError_t g_last = hipSuccess;

void fiber_entry()
{
  Block* b = g_blk;
  (*b->body)();
  b->fibers[b->current].state = DONE;
  ctx_switch(b->fibers[b->current].ctx, b->sched);
}

void yield_to_scheduler()
{
  Block* b = g_blk;
  Fiber& f = b->fibers[b->current];
  ctx_switch(f.ctx, b->sched);
  /* resumed: restore the thread's built-ins */
  threadIdx = f.tid;
}

void run_block(Block& b)
{
  g_blk = &b;
  const int n = (int)b.fibers.size();
  /* Wave order.  Between two barriers the waves of a workgroup run in no particular order on the hardware; here they run one after
   * the other, in an order that changes with every scheduling round (start wave and direction), so that a missing barrier between
   * a wave's LDS / global write and another wave's read is not hidden by "wave 0 always runs first".  Lanes of one wave stay in
   * lane order.  SIMT_EMU_WAVE_ORDER=linear switches it off. */
  static const bool wave_shuffle = [] { const char* e = getenv("SIMT_EMU_WAVE_ORDER"); return !(e && !strcmp(e, "linear")); }();
  static thread_local unsigned round_no = 0;
  const int nwaves = (n + 63) / 64;
  for (;;) {
    bool progressed = false;
    int done = 0;
    const unsigned rn = wave_shuffle ? ++round_no * 2654435761u >> 8 : 0;
    const int w0 = (int)(rn % (unsigned)nwaves), dir = (rn >> 12) & 1;
    for (int k = 0; k < nwaves * 64; k++) {
      const int wk = k / 64, w = dir ? (w0 + nwaves - wk) % nwaves : (w0 + wk) % nwaves;
      const int i = w * 64 + k % 64;
      if (i >= n) continue;
      Fiber& f = b.fibers[i];
      if (f.state == DONE) { done++; continue; }
      if (f.state != RUNNABLE) continue;
      b.current = i;
      threadIdx = f.tid;
      ctx_switch(b.sched, f.ctx);
      progressed = true;
    }
    if (done == n) break;
    /* release wave barriers whose live lanes have all arrived */
    const int nw = (n + 63) / 64;
    for (int w = 0; w < nw; w++) {
      int live = 0, waiting = 0;
      for (int i = w * 64; i < n && i < (w + 1) * 64; i++) {
        if (b.fibers[i].state != DONE) live++;
        if (b.fibers[i].state == AT_WAVE_BARRIER) waiting++;
      }
      if (live && waiting == live) {
        for (int i = w * 64; i < n && i < (w + 1) * 64; i++)
          if (b.fibers[i].state == AT_WAVE_BARRIER) b.fibers[i].state = RUNNABLE;
        progressed = true;
      }
    }
    /* release the block barrier */
    {
      int live = 0, waiting = 0;
      for (int i = 0; i < n; i++) {
        if (b.fibers[i].state != DONE) live++;
        if (b.fibers[i].state == AT_BLOCK_BARRIER) waiting++;
      }
      if (live && waiting == live) {
        /* threads that have returned stay finished: like the hardware, a barrier only counts the waves still running */
        for (int i = 0; i < n; i++) if (b.fibers[i].state == AT_BLOCK_BARRIER) b.fibers[i].state = RUNNABLE;
        progressed = true;
      }
    }
    if (!progressed) {
      fprintf(stderr, "simt_emu: DEADLOCK in block (%u,%u,%u): a barrier or wave collective was reached by only part of its threads\n",
              blockIdx.x, blockIdx.y, blockIdx.z);
      for (int i = 0; i < n; i++)
        if (b.fibers[i].state != DONE)
          fprintf(stderr, "  thread %d state %d\n", i, (int)b.fibers[i].state);
      abort();
    }
  }
  g_blk = nullptr;
}

} // namespace

namespace simt {

void launch(dim3 grid, dim3 block, const std::function<void()>& body, const char* kernel_name)
{
  /* one launch at a time: __shared__ variables are statics and the thread built-ins globals here — host threads of different contexts (the ranks of
     tests/test_shard_ipc_emu.py) take turns */
  static std::mutex launch_mu;
  std::lock_guard<std::mutex> launch_lock(launch_mu);
  gridDim = grid; blockDim = block;
  const int n = (int)(block.x * block.y * block.z);
  Block b;
  b.body = &body;
  b.fibers.resize(n);
  b.slots.assign(((n + 63) / 64) * 64, 0);
  while ((int)g_stacks.v.size() < n) g_stacks.v.push_back((char*)malloc(kStack));
  for (int i = 0; i < n; i++) b.fibers[i].stack = g_stacks.v[i];
  /* Block order.  The hardware promises none; the default here is a different pseudo-random permutation for every launch, so a
   * kernel whose blocks lean on each other's results without saying so fails in the CPU tier.  SIMT_EMU_ORDER=linear | reverse |
   * shuffle (default).  One kernel is exempt and always runs in index order: k_intra, whose workgroups WAIT for each other by
   * design (k_intra.hip: on inter pictures the CTBs without a dependency go by workgroup index, the dependent ones by an atomic
   * ticket; it counts on the dispatcher starting lower indices first, and has a spin bound for the day it does not) — blocks run
   * one after the other here, so a waiting block must come after the one it waits for.  SIMT_EMU_INORDER=<substring> names
   * another such kernel. */
  static const int order_env = [] { const char* e = getenv("SIMT_EMU_ORDER"); return !e || !*e || !strcmp(e, "shuffle") ? 2 : (!strcmp(e, "reverse") ? 1 : 0); }();
  static const char* inorder_env = getenv("SIMT_EMU_INORDER");
  const bool in_order = kernel_name && (strstr(kernel_name, "k_intra<") || (inorder_env && *inorder_env && strstr(kernel_name, inorder_env)));
  const int order_mode = in_order ? 0 : order_env;
  static thread_local uint64_t launch_no = 0;
  const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
  /* an affine permutation i -> (a * i + c) mod total with gcd(a, total) = 1 */
  uint64_t a = 1, c = 0;
  if (order_mode == 2 && total > 1) {
    uint64_t h = (++launch_no) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    a = (h % total) | 1;
    auto gcd = [](uint64_t x, uint64_t y) { while (y) { const uint64_t t = x % y; x = y; y = t; } return x; };
    while (gcd(a, total) != 1) a += 2;
    a %= total; if (a == 0) a = 1;
    c = (h >> 17) % total;
  }
  for (uint64_t k = 0; k < total; k++) {
    const uint64_t lin = order_mode == 1 ? total - 1 - k : (order_mode == 2 ? (uint64_t)(((unsigned __int128)a * k + c) % total) : k);
    const unsigned bx = (unsigned)(lin % grid.x), by = (unsigned)((lin / grid.x) % grid.y), bz = (unsigned)(lin / ((uint64_t)grid.x * grid.y));
    blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
    for (int i = 0; i < n; i++) {
      Fiber& f = b.fibers[i];
      f.state = RUNNABLE;
      f.tid.x = i % block.x; f.tid.y = (i / block.x) % block.y; f.tid.z = i / (block.x * block.y);
      f.wave = i / 64; f.lane = i % 64;
      ctx_make(f.ctx, f.stack, kStack, fiber_entry);
    }
    run_block(b);
  }
}

void sync_threads()
{
  g_blk->fibers[g_blk->current].state = AT_BLOCK_BARRIER;
  yield_to_scheduler();
}
void wave_rendezvous()
{
  g_blk->fibers[g_blk->current].state = AT_WAVE_BARRIER;
  yield_to_scheduler();
}
unsigned long long* wave_slots() { return &g_blk->slots[g_blk->fibers[g_blk->current].wave * 64]; }
int lane_id() { return g_blk->fibers[g_blk->current].lane; }
unsigned long long live_mask()
{
  const int w = g_blk->fibers[g_blk->current].wave;
  unsigned long long m = 0;
  const int n = (int)g_blk->fibers.size();
  for (int i = w * 64; i < n && i < (w + 1) * 64; i++)
    if (g_blk->fibers[i].state != DONE) m |= 1ull << (i - w * 64);
  return m;
}

} // namespace simt

/* ---- host runtime: device memory is host memory ---- */
struct simt_stream { int dummy; };
struct simt_event { std::chrono::steady_clock::time_point t; };

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "simt_emu error"; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "simt_emu"); strcpy(p->gcnArchName, "simt_emu"); p->multiProcessorCount = 1; return hipSuccess; }
/* Device (and pinned) memory is NOT zero on the GPU, so it is not zero here: every allocation is filled with 0xA5 — a kernel or a
 * host path that relies on fresh memory being zero fails in the CPU tier.  SIMT_EMU_POISON=<byte> picks another fill (0 = zeros). */
static int poison_byte()
{
  static const int v = [] { const char* e = getenv("SIMT_EMU_POISON"); return e && *e ? (int)strtol(e, nullptr, 0) & 255 : 0xA5; }();
  return v;
}
hipError_t hipMalloc(void** p, size_t n)
{
  *p = malloc(n ? n : 1);
  if (*p) memset(*p, poison_byte(), n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }   /* pinned memory is not zero either */
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{
  for (size_t y = 0; y < h; y++) memmove((char*)d + y * dp, (const char*)s + y * sp, w);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { return hipMemcpy(d, s, n, hipMemcpyDeviceToDevice); }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new simt_stream; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new simt_event; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   /* launches are synchronous here */
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
