/*
 * runtime_ipc.hip — tile sharding across the rank PROCESSES of one node without a collective library (the couplings of
 * deblock.cc:191-209 / sao.cc:158-163 as neighbour exchanges, the finished tiles as direct all-to-all peer copies, not a ring).
 *
 * The transport of m355_decode_sharded that m355_shard_ipc_init installs is the in-process group's algorithm (runtime_shard.hip
 * group_rank_decode) with its three ingredients carried across process boundaries:
 *   buffers  — every rank's exchange buffers X0..X3 of a picture handle are exported once (hipIpcGetMemHandle) through a table in a POSIX
 *              shared-memory segment and mapped by the ranks that read them (hipIpcOpenMemHandle: over xGMI on a node, the same HBM when
 *              the ranks share a GPU); X0..X2: a rank fetches its neighbours' buffers as they packed them and adds; X3: a rank copies every
 *              other rank's finished-tile slot straight out of that rank's gather buffer — N-1 concurrent peer reads per rank, one per link;
 *   events   — "my buffer is packed" / "I have read yours" are interprocess events (hipEventInterprocess) recorded on the picture's own
 *              stream and waited for on the reader's (M355_IPC_HOST_SYNC=1: the recorder drains its stream instead and nobody waits on the
 *              device — the fallback where interprocess events are not to be trusted);
 *   order    — an event may only be waited for once it has been RECORDED for this picture: per rank a sequence word in the segment
 *              (picture * 8 + exchange + 1), published behind the record, spun on by the waiter; and a rank re-records an event for picture
 *              n + 1 only when every rank has enqueued ALL of picture n (one more word, met at the end of m355_decode_sharded), so a wait of
 *              picture n never sees a record of n + 1 (the group meets the same way when m355_group_decode returns).
 * Every rank must decode the same pictures in the same order with the same handle numbers (ShardedDecoder / bench.py do).  A rank that fails
 * raises the segment's abort word: the others' spins end with an error instead of waiting for it (every spin is bounded as well).
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
#define IPC_MAGIC 0x4D333535u

struct IpcRank {
  std::atomic<unsigned long long> seq_pack, seq_copied, seq_done;
  std::atomic<unsigned> ready;                               /* 1: the event handles below are valid */
  hipIpcEventHandle_t ev_pack[4], ev_copied[4];              /* [exchange] */
  hipIpcEventHandle_t ev_x3[IPC_MAXH];                       /* this rank (the READER) has copied the other ranks' gather buffers of handle h */
  std::atomic<unsigned long long> x3_seq[IPC_MAXH];          /* ... recorded for picture x3_seq - 1 (0: never) */
  struct Buf { std::atomic<unsigned long long> gen; hipIpcMemHandle_t mem[4]; } buf[IPC_MAXH];   /* the exchange buffers of handle h (gen 0: none) */
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
  hipEvent_t ev_pack[4] = {}, ev_copied[4] = {}, ev_x3[IPC_MAXH] = {};
  /* what this rank opened of the others */
  struct Peer {
    bool events = false;
    hipEvent_t ev_pack[4] = {}, ev_copied[4] = {}, ev_x3[IPC_MAXH] = {};
    struct Map { unsigned long long gen = 0; void* p[4] = {nullptr, nullptr, nullptr, nullptr}; } map[IPC_MAXH];
  } peer[IPC_MAXR];
  unsigned long long my_gen[IPC_MAXH] = {};                  /* generation under which handle h's buffers were exported (0: not yet) */
  unsigned long long exported[IPC_MAXH] = {};                /* ... and which allocation that was (Resident::xb_epoch: a re-allocation is exported again) */
  unsigned long long pic = 0;                                /* pictures decoded through this transport (the same number on every rank) */
  unsigned long long gen_counter = 0;
};

static double ipc_timeout_s() { const char* e = getenv("M355_IPC_TIMEOUT"); const double t = e ? atof(e) : 30.0; return t > 0 ? t : 30.0; }

/* spin until a >= want; an error when the job is aborted or the bound is hit */
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
static int ipc_abort(Ipc& I, int rc) { if (I.shm) I.shm->abort_flag.store(1); return rc; }

static int ipc_peer_events(Ipc& I, int q)
{
  Ipc::Peer& P = I.peer[q];
  if (P.events) return M355_OK;
  IpcRank& R = I.shm->rank[q];
  {
    const auto t0 = std::chrono::steady_clock::now();
    while (!R.ready.load(std::memory_order_acquire)) {
      if (I.shm->abort_flag.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) return ipc_abort(I, fail(M355_ERR_TIMEOUT, "ipc transport: rank %d never joined", q));
      std::this_thread::yield();
    }
  }
  for (int k = 0; k < 4; k++)
    if (hipIpcOpenEventHandle(&P.ev_pack[k], R.ev_pack[k]) != hipSuccess || hipIpcOpenEventHandle(&P.ev_copied[k], R.ev_copied[k]) != hipSuccess)
      return ipc_abort(I, fail(M355_ERR_HIP, "hipIpcOpenEventHandle failed (rank %d)", q));
  for (int h = 0; h < IPC_MAXH; h++)
    if (hipIpcOpenEventHandle(&P.ev_x3[h], R.ev_x3[h]) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipIpcOpenEventHandle failed (rank %d)", q));
  P.events = true;
  return M355_OK;
}

/* rank q's exchange buffer k of picture handle h, mapped into this process (once per export) */
static int ipc_peer_buf(Ipc& I, int q, int h, int k, void** out)
{
  IpcRank::Buf& B = I.shm->rank[q].buf[h];
  const unsigned long long gen = B.gen.load(std::memory_order_acquire);
  if (!gen) return ipc_abort(I, fail(M355_ERR_INVALID, "ipc transport: rank %d has not exported the buffers of handle %d", q, h));
  Ipc::Peer::Map& M = I.peer[q].map[h];
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

/* the stream has reached "packed" / "copied" of exchange k: record + publish (host_sync: drain instead — nobody then waits on the device) */
static int ipc_mark(Ipc& I, hipEvent_t ev, std::atomic<unsigned long long>& seq, int k, hipStream_t st)
{
  if (I.host_sync) { if (hipStreamSynchronize(st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipStreamSynchronize failed")); }
  else if (hipEventRecord(ev, st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipEventRecord failed"));
  seq.store(I.pic * 8 + (unsigned long long)k + 1, std::memory_order_release);
  return M355_OK;
}
static void ipc_wait_dev(Ipc& I, hipStream_t st, hipEvent_t ev) { if (!I.host_sync) hipStreamWaitEvent(st, ev, 0); }

static int ipc_halo_sum(void* user, void* buf, size_t bytes, const int* peers, int n_peers, void* scratch, void* stream)
{
  Ipc& I = *(Ipc*)user;
  m355_ctx* c = I.c;
  hipStream_t st = (hipStream_t)stream;
  const int h = c->xchg_h, k = c->xchg_k;
  if (h < 0 || h >= IPC_MAXH || k < 0 || k > 2) return ipc_abort(I, fail(M355_ERR_INVALID, "ipc transport: handle %d / exchange %d out of range", h, k));
  int rc = ipc_export(I, h);
  if (rc) return rc;
  IpcRank& me = I.shm->rank[I.rank];
  const unsigned long long want = I.pic * 8 + (unsigned long long)k + 1;
  const size_t pitch = (bytes + 255) & ~(size_t)255;
  if ((rc = ipc_mark(I, I.ev_pack[k], me.seq_pack, k, st))) return rc;
  /* step 1: the neighbours' buffers as they packed them */
  for (int i = 0; i < n_peers; i++) {
    const int q = peers[i];
    if ((rc = ipc_peer_events(I, q)) || (rc = ipc_await(I, I.shm->rank[q].seq_pack, want, "packed", q))) return rc;
    void* pb = nullptr;
    if ((rc = ipc_peer_buf(I, q, h, k, &pb))) return rc;
    ipc_wait_dev(I, st, I.peer[q].ev_pack[k]);
    if (hipMemcpyAsync((char*)scratch + pitch * (size_t)i, pb, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "peer copy failed (rank %d)", q));
  }
  if ((rc = ipc_mark(I, I.ev_copied[k], me.seq_copied, k, st))) return rc;
  /* step 2: add, once this rank's own buffer has been read by all of them */
  for (int i = 0; i < n_peers; i++) {
    const int q = peers[i];
    if ((rc = ipc_await(I, I.shm->rank[q].seq_copied, want, "copied", q))) return rc;
    ipc_wait_dev(I, st, I.peer[q].ev_copied[k]);
  }
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
  IpcRank& me = I.shm->rank[I.rank];
  const unsigned long long want = I.pic * 8 + 3 + 1;
  if ((rc = ipc_mark(I, I.ev_pack[3], me.seq_pack, 3, st))) return rc;
  for (int q = 0; q < I.n; q++) {
    if (q == I.rank) continue;
    if ((rc = ipc_peer_events(I, q)) || (rc = ipc_await(I, I.shm->rank[q].seq_pack, want, "tiles packed", q))) return rc;
    void* pb = nullptr;
    if ((rc = ipc_peer_buf(I, q, h, 3, &pb))) return rc;
    ipc_wait_dev(I, st, I.peer[q].ev_pack[3]);
    if (hipMemcpyAsync((char*)buf + slot_bytes * (size_t)q, (const char*)pb + slot_bytes * (size_t)q, slot_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return ipc_abort(I, fail(M355_ERR_HIP, "peer copy of rank %d's tiles failed", q));
  }
  /* this rank has read the others' gather buffers of handle h: said per handle, for their next repack of it (ipc_before_repack) */
  if (I.host_sync) { if (hipStreamSynchronize(st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipStreamSynchronize failed")); }
  else if (hipEventRecord(I.ev_x3[h], st) != hipSuccess) return ipc_abort(I, fail(M355_ERR_HIP, "hipEventRecord failed"));
  me.x3_seq[h].store(I.pic + 1, std::memory_order_release);
  return 0;
}

/* ---- hooks of m355_decode_sharded (runtime_shard.hip) ---- */
/* in front of phase 3 of a reference picture: the gather buffer of handle h is about to be repacked — every rank's read of it at the handle's
   previous decode has to be over (those reads were enqueued before that decode's end-of-picture meeting: their records exist) */
int ipc_before_repack(m355_ctx* c, int h, hipStream_t st)
{
  Ipc& I = *(Ipc*)c->ipc;
  if (I.host_sync || h < 0 || h >= IPC_MAXH) return M355_OK;
  for (int q = 0; q < I.n; q++) {
    if (q == I.rank || !I.shm->rank[q].x3_seq[h].load(std::memory_order_acquire)) continue;
    int rc = ipc_peer_events(I, q);
    if (rc) return rc;
    hipStreamWaitEvent(st, I.peer[q].ev_x3[h], 0);
  }
  return M355_OK;
}
/* behind the last call of a picture: every rank has enqueued all of it (see the header) */
int ipc_end_picture(m355_ctx* c, int rc_own)
{
  Ipc& I = *(Ipc*)c->ipc;
  if (rc_own) return ipc_abort(I, rc_own);
  I.shm->rank[I.rank].seq_done.store(I.pic + 1, std::memory_order_release);
  for (int q = 0; q < I.n; q++) {
    if (q == I.rank) continue;
    int rc = ipc_await(I, I.shm->rank[q].seq_done, I.pic + 1, "picture enqueued", q);
    if (rc) return rc;
  }
  I.pic++;
  return M355_OK;
}

extern "C" int m355_shard_ipc_close(m355_ctx* c)
{
  if (!c || !c->ipc) return M355_OK;
  Ipc* I = (Ipc*)c->ipc;
  hipSetDevice(c->device);
  sync_all(c);
  for (int q = 0; q < IPC_MAXR; q++) {
    Ipc::Peer& P = I->peer[q];
    for (int h = 0; h < IPC_MAXH; h++) for (int j = 0; j < 4; j++) if (P.map[h].p[j]) hipIpcCloseMemHandle(P.map[h].p[j]);
    if (P.events) {
      for (int k = 0; k < 4; k++) { hipEventDestroy(P.ev_pack[k]); hipEventDestroy(P.ev_copied[k]); }
      for (int h = 0; h < IPC_MAXH; h++) hipEventDestroy(P.ev_x3[h]);
    }
  }
  for (int k = 0; k < 4; k++) { if (I->ev_pack[k]) hipEventDestroy(I->ev_pack[k]); if (I->ev_copied[k]) hipEventDestroy(I->ev_copied[k]); }
  for (int h = 0; h < IPC_MAXH; h++) if (I->ev_x3[h]) hipEventDestroy(I->ev_x3[h]);
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
  /* this rank's events */
  IpcRank& me = I->shm->rank[rank];
  auto mk = [&](hipEvent_t* e, hipIpcEventHandle_t* hdl) -> bool {
    return hipEventCreateWithFlags(e, hipEventDisableTiming | hipEventInterprocess) == hipSuccess && hipIpcGetEventHandle(hdl, *e) == hipSuccess;
  };
  bool ok = true;
  for (int k = 0; k < 4 && ok; k++) ok = mk(&I->ev_pack[k], &me.ev_pack[k]) && mk(&I->ev_copied[k], &me.ev_copied[k]);
  for (int h = 0; h < IPC_MAXH && ok; h++) ok = mk(&I->ev_x3[h], &me.ev_x3[h]);
  if (!ok) { ipc_abort(*I, 0); m355_shard_ipc_close(c); return fail(M355_ERR_HIP, "interprocess events are not available (hipEventInterprocess / hipIpcGetEventHandle failed)"); }
  me.ready.store(1, std::memory_order_release);
  m355_comm cm = {I, ipc_halo_sum, ipc_all_gather};
  return m355_shard_set_comm(c, &cm);
}
