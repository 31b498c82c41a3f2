/*
 * runtime_ipc.hip — tile sharding across the rank PROCESSES of one node without a collective library (the couplings of
 * deblock.cc:191-209 / sao.cc:158-163 as neighbour exchanges, the finished tiles as direct all-to-all peer copies, not a ring).
 *
 * The transport of m355_decode_sharded that m355_shard_ipc_init installs is the in-process group's algorithm (runtime_shard.hip
 * group_rank_decode) carried across process boundaries:
 *   buffers — every rank's exchange buffers X0..X3 of a picture handle are exported once (hipIpcGetMemHandle) through a table in a POSIX
 *             shared-memory segment and mapped by the ranks that read them (hipIpcOpenMemHandle: over xGMI on a node, the same HBM when the ranks
 *             share a GPU).  X0..X2: a rank fetches its neighbours' buffers as they packed them, and adds once its own has been read by all of
 *             them.  X3: a rank copies every other rank's finished-tile slot straight out of that rank's gather buffer — N - 1 concurrent peer
 *             reads per rank, one per link.
 *   order   — FLAG WORDS in device memory, one set per rank, exported like the buffers: "my buffer of exchange k is packed" / "I have read yours" /
 *             "I have read the gather buffers of handle h" hold the number (picture * 8 + exchange + 1) of the last picture for which that is
 *             true.  A rank raises a flag with a one-thread kernel on the picture's own stream (behind the kernels that packed / the copies that
 *             read) and waits for a peer's flag with a one-wave kernel in front of its copy; the numbers only grow, so a wait of picture n can
 *             never be satisfied by anything but picture n or a later one of the same stream order — no host thread waits for another rank in
 *             steady state, no meeting per picture, pictures stay in flight across the exchanges.  (HIP's own interprocess events were tried
 *             first: hipStreamWaitEvent on an imported event returns hipErrorInvalidValue once the ranks run one picture at a time — round 6,
 *             visit 7 — and its waits are host callbacks.)  Every wait kernel is bounded (M355_IPC_TIMEOUT seconds): a peer that never arrives
 *             costs an error word, not a hung GPU.
 *   M355_IPC_HOST_SYNC=1 (and the CPU tier's interpreter, whose launches are synchronous): no kernel waits — a rank drains its stream, then
 *             publishes the same number in the segment, and the reader's HOST thread spins on it.
 * Every rank must decode the same pictures in the same order with the same handle numbers (ShardedDecoder / bench.py do).  A rank that fails
 * raises the segment's abort word: the others' host-side waits end with an error (every wait is bounded as well).
 */
#include "runtime_internal.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>

#define IPC_MAXR 16     /* ranks of one node */
#define IPC_MAXH 32     /* picture handles with exchange buffers per rank */
#define IPC_MAGIC 0x4D333536u
/* a rank's flag words (device memory), nine per picture handle h: packed[k] (k = 0..3), copied[k], the other ranks' tiles of h read.  Per HANDLE, not per
   exchange: pictures in flight run on different streams, and a later picture's exchange k may well complete before an earlier picture's — two decodes of one
   handle never overlap (the second starts behind the first's completion mark), so a handle's words only ever grow */
#define IPC_F_PACK(h, k) ((h) * 9 + (k))
#define IPC_F_COPIED(h, k) ((h) * 9 + 4 + (k))
#define IPC_F_X3(h) ((h) * 9 + 8)
#define IPC_FLAGS (9 * IPC_MAXH)

struct IpcRank {
  std::atomic<unsigned> ready;                               /* 1: `flags` below is valid */
  hipIpcMemHandle_t flags;                                   /* the rank's flag words */
  std::atomic<unsigned long long> host_flag[IPC_FLAGS];      /* the same numbers for host-side waits (M355_IPC_HOST_SYNC, the interpreter) */
  struct Buf { std::atomic<unsigned long long> gen; hipIpcMemHandle_t mem[4]; } buf[IPC_MAXH];   /* the exchange buffers of handle h (gen 0: none) */
  std::atomic<unsigned long long> rel[IPC_MAXH];            /* how many generations of handle h's buffers this rank has let go of (ipc_before_free: a meeting of all ranks) */
};
struct IpcShm {
  std::atomic<unsigned> magic, abort_flag;
  IpcRank rank[IPC_MAXR];
};

struct Ipc {
  m355_ctx* c = nullptr;
  int rank = 0, n = 1;
  IpcShm* shm = nullptr;
  std::string name;
  bool host_sync = false, owner = false;
  unsigned long long* flags = nullptr;                       /* this rank's flag words (device) */
  unsigned* err = nullptr;                                   /* pinned host word a wait kernel raises when it gives up */
  /* what this rank opened of the others */
  struct Peer {
    unsigned long long* flags = nullptr;
    struct Map { unsigned long long gen = 0, closed = 0; void* p[4] = {nullptr, nullptr, nullptr, nullptr}; } map[IPC_MAXH];   /* closed: the last generation this rank has let go of */
  } peer[IPC_MAXR];
  unsigned long long rel_count[IPC_MAXH] = {};
  bool broken = false;                                       /* a wait for another rank has failed: no further meetings */
  unsigned long long my_gen[IPC_MAXH] = {};                  /* generation under which handle h's buffers were exported (0: not yet) */
  unsigned long long exported[IPC_MAXH] = {};                /* ... and which allocation that was (Resident::xb_epoch: a re-allocation is exported again) */
  unsigned long long x3_packed[IPC_MAXH] = {};               /* the number of this rank's last gather on handle h (0: none): what the readers' flags must reach before it is repacked */
  unsigned long long pic = 0;                                /* pictures decoded through this transport (the same number on every rank) */
  unsigned long long gen_counter = 0;
};

static double ipc_timeout_s() { const char* e = getenv("M355_IPC_TIMEOUT"); const double t = e ? atof(e) : 30.0; return t > 0 ? t : 30.0; }
static int ipc_abort(Ipc& I, int rc) { if (I.shm) I.shm->abort_flag.store(1); return rc; }

/* raise a flag: everything enqueued on the stream so far is visible to whoever sees the new number */
__global__ void k_ipc_signal(unsigned long long* flag, unsigned long long v)
{
  __threadfence_system();
  __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
/* wait until a peer's flag has reached v (one lane polls; bounded by `ticks` of the 100 MHz wall clock: then *err = 1 and the stream goes on) */
__global__ void k_ipc_wait(const unsigned long long* flag, unsigned long long v, unsigned* err, unsigned long long ticks)
{
  if (threadIdx.x) return;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
  }
}

/* host-side: spin until a >= want; an error when the job is aborted or the bound is hit */
static int ipc_await(Ipc& I, std::atomic<unsigned long long>& a, unsigned long long want, const char* what, int q)
{
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (a.load(std::memory_order_acquire) < want) {
    if (I.shm->abort_flag.load(std::memory_order_relaxed)) return fail(M355_ERR_HIP, "ipc transport: the job was aborted by another rank (waiting for rank %d: %s)", q, what);
    if ((++spins & 1023u) == 0) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) {
        I.shm->abort_flag.store(1);
        return fail(M355_ERR_TIMEOUT, "ipc transport: rank %d did not reach '%s' within %.0f s", q, what, ipc_timeout_s());
      }
      std::this_thread::yield();
    }
  }
  return M355_OK;
}

/* rank q's flag words, mapped (first use) */
static int ipc_peer_flags(Ipc& I, int q)
{
  Ipc::Peer& P = I.peer[q];
  if (P.flags || I.host_sync) return M355_OK;
  IpcRank& R = I.shm->rank[q];
  const auto t0 = std::chrono::steady_clock::now();
  while (!R.ready.load(std::memory_order_acquire)) {
    if (I.shm->abort_flag.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) return ipc_abort(I, fail(M355_ERR_TIMEOUT, "ipc transport: rank %d never joined", q));
    std::this_thread::yield();
  }
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, R.flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipIpcOpenMemHandle failed (rank %d's flag words)", q));
  P.flags = (unsigned long long*)p;
  return M355_OK;
}

/* rank q's exchange buffer k of picture handle h, mapped into this process (once per export; the first use waits for the export) */
static int ipc_peer_buf(Ipc& I, int q, int h, int k, void** out)
{
  IpcRank::Buf& B = I.shm->rank[q].buf[h];
  Ipc::Peer::Map& M = I.peer[q].map[h];
  unsigned long long gen = B.gen.load(std::memory_order_acquire);
  if (gen <= M.closed) {          /* (not yet exported — or still the generation this rank let go of at the handle's release: the owner exports the new buffers at its first exchange) */
    int rc = ipc_await(I, B.gen, M.closed + 1, "buffers exported", q);
    if (rc) return rc;
    gen = B.gen.load(std::memory_order_acquire);
  }
  if (M.gen != gen) {
    for (int j = 0; j < 4; j++) if (M.p[j]) { hipIpcCloseMemHandle(M.p[j]); M.p[j] = nullptr; }
    for (int j = 0; j < 4; j++)
      if (hipIpcOpenMemHandle(&M.p[j], B.mem[j], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipIpcOpenMemHandle failed (rank %d, handle %d, buffer %d)", q, h, j));
    M.gen = gen;
  }
  *out = M.p[k];
  return M355_OK;
}

/* this rank's buffers of handle h into the table (at the handle's first exchange, and again when they were re-allocated) */
static int ipc_export(Ipc& I, int h)
{
  Resident& r = I.c->resident[h];
  if (I.my_gen[h] && I.exported[h] == r.xb_epoch) return M355_OK;
  IpcRank::Buf& B = I.shm->rank[I.rank].buf[h];
  for (int k = 0; k < 4; k++)
    if (hipIpcGetMemHandle(&B.mem[k], r.xb[k]) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipIpcGetMemHandle failed (handle %d, buffer %d): is HSA_ENABLE_IPC_MODE_LEGACY=0 set?", h, k));
  I.my_gen[h] = ++I.gen_counter;                             /* (a generation of its own per export: the readers map again when it changes) */
  I.exported[h] = r.xb_epoch;
  B.gen.store(I.my_gen[h], std::memory_order_release);
  return M355_OK;
}

/* the stream has reached flag `f` of this picture: raise it (host_sync: drain the stream, then publish the number in the segment) */
static int ipc_raise(Ipc& I, int f, unsigned long long v, hipStream_t st)
{
  if (I.host_sync) {
    if (hipStreamSynchronize(st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipStreamSynchronize failed"));
    I.shm->rank[I.rank].host_flag[f].store(v, std::memory_order_release);
    return M355_OK;
  }
  hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(1), 0, st, I.flags + f, v);
  return M355_OK;
}
/* the stream continues when rank q's flag `f` has reached v */
static int ipc_wait(Ipc& I, int q, int f, unsigned long long v, const char* what, hipStream_t st)
{
  if (I.host_sync) return ipc_await(I, I.shm->rank[q].host_flag[f], v, what, q);
  int rc = ipc_peer_flags(I, q);
  if (rc) return rc;
  if (*(volatile unsigned*)I.err) return ipc_abort(I, fail(M355_ERR_TIMEOUT, "ipc transport: an earlier wait for another rank gave up (M355_IPC_TIMEOUT)"));
  hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, st, (const unsigned long long*)(I.peer[q].flags + f), v, I.err, (unsigned long long)(ipc_timeout_s() * 1e8));
  return M355_OK;
}

static int ipc_halo_sum(void* user, void* buf, size_t bytes, const int* peers, int n_peers, void* scratch, void* stream)
{
  Ipc& I = *(Ipc*)user;
  m355_ctx* c = I.c;
  hipStream_t st = (hipStream_t)stream;
  const int h = c->xchg_h, k = c->xchg_k;
  if (h < 0 || h >= IPC_MAXH || k < 0 || k > 2) return ipc_abort(I, fail(M355_ERR_INVALID, "ipc transport: handle %d / exchange %d out of range (at most %d handles)", h, k, IPC_MAXH));
  int rc = ipc_export(I, h);
  if (rc) return rc;
  const unsigned long long v = I.pic * 8 + (unsigned long long)k + 1;
  const size_t pitch = (bytes + 255) & ~(size_t)255;
  if ((rc = ipc_raise(I, IPC_F_PACK(h, k), v, st))) return rc;
  /* step 1: the neighbours' buffers as they packed them */
  for (int i = 0; i < n_peers; i++) {
    const int q = peers[i];
    void* pb = nullptr;
    if ((rc = ipc_peer_buf(I, q, h, k, &pb)) || (rc = ipc_wait(I, q, IPC_F_PACK(h, k), v, "packed", st))) return rc;
    if (hipMemcpyAsync((char*)scratch + pitch * (size_t)i, pb, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "peer copy failed (rank %d)", q));
  }
  if ((rc = ipc_raise(I, IPC_F_COPIED(h, k), v, st))) return rc;
  /* step 2: add, once this rank's own buffer has been read by all of them */
  for (int i = 0; i < n_peers; i++)
    if ((rc = ipc_wait(I, peers[i], IPC_F_COPIED(h, k), v, "copied", st))) return rc;
  m355_launch_halo_add((uint32_t*)buf, (const uint32_t*)scratch, (uint32_t)(pitch / 4), n_peers, (uint32_t)((bytes + 3) / 4), st);
  return 0;
}

static int ipc_all_gather(void* user, void* buf, size_t slot_bytes, int rank, int nranks, void* stream)
{
  Ipc& I = *(Ipc*)user;
  m355_ctx* c = I.c;
  hipStream_t st = (hipStream_t)stream;
  const int h = c->xchg_h;
  if (h < 0 || h >= IPC_MAXH || rank != I.rank || nranks != I.n) return ipc_abort(I, fail(M355_ERR_INVALID, "ipc transport: bad all-gather arguments"));
  int rc = ipc_export(I, h);
  if (rc) return rc;
  const unsigned long long v = I.pic * 8 + 3 + 1;
  if ((rc = ipc_raise(I, IPC_F_PACK(h, 3), v, st))) return rc;
  for (int q = 0; q < I.n; q++) {
    if (q == I.rank) continue;
    void* pb = nullptr;
    if ((rc = ipc_peer_buf(I, q, h, 3, &pb)) || (rc = ipc_wait(I, q, IPC_F_PACK(h, 3), v, "tiles packed", st))) return rc;
    if (hipMemcpyAsync((char*)buf + slot_bytes * (size_t)q, (const char*)pb + slot_bytes * (size_t)q, slot_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return ipc_abort(I, fail(M355_ERR_HIP, "peer copy of rank %d's tiles failed", q));
  }
  /* this rank has read the others' gather buffers of handle h: said per handle, for their next repack of it (ipc_before_repack) */
  if ((rc = ipc_raise(I, IPC_F_X3(h), v, st))) return rc;
  I.x3_packed[h] = v;
  return 0;
}

/* ---- hooks of m355_decode_sharded (runtime_shard.hip) ---- */
/* in front of phase 3 of a reference picture: the gather buffer of handle h is about to be repacked — every rank's read of it at the handle's
   previous gather has to be over */
int ipc_before_repack(m355_ctx* c, int h, hipStream_t st)
{
  Ipc& I = *(Ipc*)c->ipc;
  if (h < 0 || h >= IPC_MAXH || !I.x3_packed[h]) return M355_OK;
  for (int q = 0; q < I.n; q++) {
    if (q == I.rank) continue;
    int rc = ipc_wait(I, q, IPC_F_X3(h), I.x3_packed[h], "tiles read", st);
    if (rc) return rc;
  }
  return M355_OK;
}
/* The exchange buffers of handle h are about to be freed (m355_picture_release, a re-allocation, the rank leaving) — a MEETING of all ranks, which release their
   handles in the same order as they decode: (1) the other ranks' last reads of this rank's gather buffer are over (the halo buffers' readers were waited for by
   the exchange itself, the gather buffer's only by its next repack); (2) this rank lets go of its mappings of the others' buffers of h; (3) every rank has done
   the same — only then may anybody free: a reader that ran ahead into the handle's next life would otherwise copy out of the old mapping.  The next use maps the
   generation the owner exports then (ipc_peer_buf waits for one newer than the one closed here). */
int ipc_before_free(m355_ctx* c, int h)
{
  Ipc& I = *(Ipc*)c->ipc;
  if (h < 0 || h >= IPC_MAXH || !I.my_gen[h]) return M355_OK;
  int rc = M355_OK;
  if (!I.broken && !I.shm->abort_flag.load()) {
    hipStream_t st = (hipStream_t)m355_stream(c);
    if (I.x3_packed[h]) rc = ipc_before_repack(c, h, st);
    if (!rc && !I.host_sync && hipStreamSynchronize(st) != hipSuccess) rc = fail(M355_ERR_HIP, "hipStreamSynchronize failed");
  }
  for (int q = 0; q < I.n; q++) {
    Ipc::Peer::Map& M = I.peer[q].map[h];
    for (int j = 0; j < 4; j++) if (M.p[j]) { hipIpcCloseMemHandle(M.p[j]); M.p[j] = nullptr; }
    if (M.gen) M.closed = M.gen;
    M.gen = 0;
  }
  I.shm->rank[I.rank].buf[h].gen.store(0, std::memory_order_release);      /* (unpublished before the meeting ends: nobody can pick the old export up afterwards) */
  I.shm->rank[I.rank].rel[h].store(++I.rel_count[h], std::memory_order_release);
  for (int q = 0; q < I.n && !rc && !I.broken; q++)
    if (q != I.rank) rc = ipc_await(I, I.shm->rank[q].rel[h], I.rel_count[h], "handle released", q);
  if (rc) I.broken = true;
  I.x3_packed[h] = 0;
  I.my_gen[h] = 0;
  return rc;
}
/* behind the last call of a picture (or of one exchange timed on its own): the picture number moves on — on every rank alike */
int ipc_end_picture(m355_ctx* c, int rc_own)
{
  Ipc& I = *(Ipc*)c->ipc;
  if (rc_own) return ipc_abort(I, rc_own);
  I.pic++;
  return M355_OK;
}

extern "C" int m355_shard_ipc_close(m355_ctx* c)
{
  if (!c || !c->ipc) return M355_OK;
  Ipc* I = (Ipc*)c->ipc;
  hipSetDevice(c->device);
  sync_all(c);
  for (int h = 0; h < IPC_MAXH; h++) ipc_before_free(c, h);
  for (int q = 0; q < IPC_MAXR; q++) {
    Ipc::Peer& P = I->peer[q];
    for (int h = 0; h < IPC_MAXH; h++) for (int j = 0; j < 4; j++) if (P.map[h].p[j]) hipIpcCloseMemHandle(P.map[h].p[j]);
    if (P.flags) hipIpcCloseMemHandle(P.flags);
  }
  if (I->flags) hipFree(I->flags);
  if (I->err) hipHostFree(I->err);
  if (I->shm) munmap(I->shm, sizeof(IpcShm));
  if (I->owner) shm_unlink(I->name.c_str());
  delete I;
  c->ipc = nullptr;
  m355_shard_set_comm(c, nullptr);
  return M355_OK;
}

extern "C" int m355_shard_ipc_init(m355_ctx* c, const char* name, int rank, int nranks)
{
  if (!c || !name || !*name || nranks < 1 || nranks > IPC_MAXR || rank < 0 || rank >= nranks) return fail(M355_ERR_INVALID, "m355_shard_ipc_init: bad arguments (at most %d ranks)", IPC_MAXR);
  if (c->ipc) m355_shard_ipc_close(c);
  int rc = m355_shard_set(c, rank, nranks);
  if (rc) return rc;
  hipSetDevice(c->device);
  Ipc* I = new Ipc;
  I->c = c; I->rank = rank; I->n = nranks;
  I->name = std::string("/m355ipc_") + name;
  I->host_sync = getenv("M355_IPC_HOST_SYNC") != nullptr && atoi(getenv("M355_IPC_HOST_SYNC")) != 0;
#ifdef SIMT_EMU
  I->host_sync = true;           /* (launches are synchronous and take turns under the interpreter: a kernel that waits for another rank's kernel would wait for ever) */
#endif
  /* the segment: rank 0 makes it (a stale one of the same name is replaced), the others attach once its magic word is there */
  int fd = -1;
  if (rank == 0) {
    shm_unlink(I->name.c_str());
    fd = shm_open(I->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(IpcShm)) != 0) { if (fd >= 0) close(fd); delete I; return fail(M355_ERR_HIP, "shm_open(%s) failed", name); }
    I->owner = true;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      fd = shm_open(I->name.c_str(), O_RDWR, 0600);
      struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(IpcShm)) break;
      if (fd >= 0) { close(fd); fd = -1; }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) { delete I; return fail(M355_ERR_TIMEOUT, "m355_shard_ipc_init: rank 0 never created the segment %s", name); }
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  void* m = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { if (I->owner) shm_unlink(I->name.c_str()); delete I; return fail(M355_ERR_HIP, "mmap of the segment failed"); }
  I->shm = (IpcShm*)m;
  c->ipc = I;
  if (rank == 0) I->shm->magic.store(IPC_MAGIC, std::memory_order_release);
  else {
    const auto t0 = std::chrono::steady_clock::now();
    while (I->shm->magic.load(std::memory_order_acquire) != IPC_MAGIC) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) { m355_shard_ipc_close(c); return fail(M355_ERR_TIMEOUT, "m355_shard_ipc_init: segment never initialised"); }
      std::this_thread::yield();
    }
  }
  /* this rank's flag words (zero: no picture yet) and the word a wait kernel raises when it gives up */
  IpcRank& me = I->shm->rank[rank];
  bool ok = hipMalloc((void**)&I->flags, 65536) == hipSuccess && hipMemset(I->flags, 0, 65536) == hipSuccess &&
            hipHostMalloc((void**)&I->err, 64, hipHostMallocMapped) == hipSuccess;
  if (ok) { *I->err = 0; ok = I->host_sync || hipIpcGetMemHandle(&me.flags, I->flags) == hipSuccess; }
  if (!ok) { ipc_abort(*I, 0); m355_shard_ipc_close(c); return fail(M355_ERR_HIP, "m355_shard_ipc_init: the flag words could not be allocated / exported (HSA_ENABLE_IPC_MODE_LEGACY=0?)"); }
  me.ready.store(1, std::memory_order_release);
  m355_comm cm = {I, ipc_halo_sum, ipc_all_gather};
  return m355_shard_set_comm(c, &cm);
}
