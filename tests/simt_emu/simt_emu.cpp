/* tests/simt_emu/simt_emu.cpp — fiber scheduler + trivial host runtime of the SIMT interpreter.
 * See hip/hip_runtime.h in this directory for what this is (and is not). */
#include "hip/hip_runtime.h"

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {

enum State { RUNNABLE, AT_BLOCK_BARRIER, AT_WAVE_BARRIER, DONE };

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  State state = RUNNABLE;
  uint3 tid;
  int wave = 0, lane = 0;
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<unsigned long long> slots;   /* 64 per wave */
  ucontext_t sched;
  int current = -1;
  const std::function<void()>* body = nullptr;
};

Block* g_blk = nullptr;
const size_t kStack = 256 * 1024;
hipError_t g_last = hipSuccess;

void fiber_entry()
{
  Block* b = g_blk;
  (*b->body)();
  b->fibers[b->current].state = DONE;
  swapcontext(&b->fibers[b->current].ctx, &b->sched);
}

void yield_to_scheduler()
{
  Block* b = g_blk;
  Fiber& f = b->fibers[b->current];
  swapcontext(&f.ctx, &b->sched);
  /* resumed: restore the thread's built-ins */
  threadIdx = f.tid;
}

void run_block(Block& b)
{
  g_blk = &b;
  const int n = (int)b.fibers.size();
  for (;;) {
    bool progressed = false;
    int done = 0;
    for (int i = 0; i < n; i++) {
      Fiber& f = b.fibers[i];
      if (f.state == DONE) { done++; continue; }
      if (f.state != RUNNABLE) continue;
      b.current = i;
      threadIdx = f.tid;
      swapcontext(&b.sched, &f.ctx);
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

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
  gridDim = grid; blockDim = block;
  const int n = (int)(block.x * block.y * block.z);
  Block b;
  b.body = &body;
  b.fibers.resize(n);
  b.slots.assign(((n + 63) / 64) * 64, 0);
  for (int i = 0; i < n; i++) b.fibers[i].stack = (char*)malloc(kStack);
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
        for (int i = 0; i < n; i++) {
          Fiber& f = b.fibers[i];
          f.state = RUNNABLE;
          f.tid.x = i % block.x; f.tid.y = (i / block.x) % block.y; f.tid.z = i / (block.x * block.y);
          f.wave = i / 64; f.lane = i % 64;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fiber_entry, 0);
        }
        run_block(b);
      }
  for (int i = 0; i < n; i++) free(b.fibers[i].stack);
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
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "simt_emu"); strcpy(p->gcnArchName, "simt_emu"); p->multiProcessorCount = 1; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
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
