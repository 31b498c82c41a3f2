/*
 * runtime_shard.hip — host side of the picture layer, part 4 of 4: tile sharding — the phases of a sharded picture, the halo / tile exchanges, the
 * in-process group (m355_group_*), the built-in RCCL transport.
 */
#include "runtime_internal.h"

Rccl g_rccl;

extern "C" {
/* ------------------------------------------------------------------ tile-sharded decode -------- */

int m355_shard_owner_of_tile(int tile, int n_tiles, int nranks)
{
  if (nranks <= 1 || n_tiles <= 0) return 0;
  return (int)(((long long)tile * nranks) / n_tiles);
}

int m355_shard_set(m355_ctx* c, int rank, int nranks)
{
  if (nranks == 0) { c->shard_rank = 0; c->shard_n = 0; return M355_OK; }
  if (nranks < 0 || rank < 0 || rank >= nranks) return fail(M355_ERR_INVALID, "bad shard rank %d of %d", rank, nranks);
  c->shard_rank = rank; c->shard_n = nranks;
  return M355_OK;
}

static size_t halo_sample_bytes(const Resident& r, int which)
{
  const HaloLayout& h = r.halo;
  const size_t n = ((which & 1) ? (size_t)h.col_ofs[3] : 0) + ((which & 2) ? (size_t)h.row_ofs[3] : 0);
  const size_t b = n * (r.hdr.pp.bit_depth_luma <= 8 ? 1 : 2);
  return (b + 3) & ~(size_t)3;
}

int64_t m355_shard_xbuf_bytes(m355_ctx* c, int h, int which)
{
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded) return -(int64_t)fail(M355_ERR_INVALID, "not a sharded picture handle");
  const Resident& r = c->resident[h];
  switch (which) {
    case 0: return (int64_t)((size_t)r.halo.n_units * 16 + halo_sample_bytes(r, 1));
    case 1: return (int64_t)halo_sample_bytes(r, 2);
    case 2: return (int64_t)halo_sample_bytes(r, 3);
    case 3: return (int64_t)(slot_bytes(r.hdr.pp, r.shard_n) * (size_t)r.shard_n);
    default: return -(int64_t)fail(M355_ERR_INVALID, "exchange buffer index %d", which);
  }
}

int m355_decode_phase(m355_ctx* c, int h, int phase, void* xbuf)
{
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded) return fail(M355_ERR_INVALID, "not a sharded picture handle");
  Resident& r = c->resident[h];
  if (phase < 0 || phase > 4 || (phase < 4 && !xbuf)) return fail(M355_ERR_INVALID, "bad phase / buffer");
  if (phase > 0 && !r.live_valid) return fail(M355_ERR_INVALID, "phase %d before phase 0", phase);
  hipSetDevice(c->device);
  /* pictures in flight: phase 0 of consecutive pictures goes round the lanes like decode(); the later phases of a picture run
     on the lane that holds its working planes.  m355_stream() is that lane's stream after every call, so the host orders its
     exchange of this picture against it while other pictures' phases run on the other lanes. */
  const bool piped = c->depth >= 2;
  if (phase == 0) { if (piped) select_lane(c, (c->active + 1) % c->depth); r.lane = c->active; }
  else if (r.lane != c->active) select_lane(c, r.lane);
  hipStream_t st = c->stream;
  const m355_pic_params& pp = r.hdr.pp;
  const bool hbd = pp.bit_depth_luma > 8;
  const size_t meta_bytes = (size_t)r.halo.n_units * 16;
  if (phase == 0) {
    int rc = prepare(c, r, r.live, r.live_sao);
    if (rc) return rc;
    r.live_valid = true;
  }
  const DevPic& d = r.live;
  const bool deblock = (c->stages & M355_STAGE_DEBLOCK) && (pp.flags & M355_PF_DEBLOCK_ENABLED);
  Frame* dstf = get_frame(c, r.hdr.dst_frame);
  auto dst_hazards = [&]() {     /* as in decode(): right before the first write of the destination frame */
    if (dstf->dl_pending) hipStreamWaitEvent(st, dstf->ev_dl, 0);
    if (!piped) return;
    ev_wait(c, st, dstf->wr);
    for (int k = 0; k < M355_MAX_LANES; k++) ev_wait(c, st, dstf->rd[k]);
  };
  auto dst_written = [&]() -> int {
    EvRef done;
    const int rcm = ev_mark(c, st, &done);                   /* one mark: the lists, the lane, the destination frame */
    if (rcm) return rcm;
    r.done = done; r.fresh = false;
    dstf->wr_stream = st;
    c->last = done; c->last_stream = st;
    dstf->wr = done;
    return M355_OK;
  };
  switch (phase) {
    case 0: {
      ev_wait(c, st, c->last);                               /* the lane's scratch and working planes (decode()) */
      /* the exchange buffers of m355_decode_sharded belong to the handle, not to a lane: a second decode of the same lists
         starts behind the last unpack of the one before */
      if (r.xb[0]) ev_wait(c, st, r.done);
      if (piped) {
        ev_wait(c, st, r.up);
      }
      if (r.device_validate) m355_launch_validate(d, st);
      if (!r.live_sao) dst_hazards();
      if (pp.flags & M355_PF_CLEAR_DST) clear_target(c, d, r.live_sao ? &c->work : dstf, r.device_validate && !r.live_sao, st);
      launch_prediction(c, r, d, hbd, nullptr);
      if (piped) {    /* the reference frames are not read after this phase */
        EvRef read;
        bool marked = false;
        for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
          Frame* f = r.hdr.ref_frames[i] >= 0 ? get_frame(c, r.hdr.ref_frames[i]) : nullptr;
          if (!f) continue;
          if (!marked) { const int rcm = ev_mark(c, st, &read); if (rcm) return rcm; marked = true; }
          f->rd[c->active] = read;
        }
      }
      m355_launch_halo_pack(d, r.halo, hbd, 1, (char*)xbuf + meta_bytes, (uint32_t*)xbuf, st);
      break;
    }
    case 1:
      m355_launch_halo_unpack(d, r.halo, hbd, 1, (const char*)r.xprev + meta_bytes, (const uint32_t*)r.xprev, st);
      if (deblock) m355_launch_deblock_pass(d, hbd, true, st);
      m355_launch_halo_pack(d, r.halo, hbd, 2, xbuf, nullptr, st);
      break;
    case 2:
      m355_launch_halo_unpack(d, r.halo, hbd, 2, r.xprev, nullptr, st);
      if (deblock) m355_launch_deblock_pass(d, hbd, false, st);
      m355_launch_halo_pack(d, r.halo, hbd, 3, xbuf, nullptr, st);
      break;
    case 3: {
      m355_launch_halo_unpack(d, r.halo, hbd, 3, r.xprev, nullptr, st);
      if (r.live_sao) { dst_hazards(); m355_launch_sao(d, hbd, st); }
      if (r.shard_n > 1) {     /* (a single rank owns every tile: nothing to hand to anybody) */
        int rc = copy_tiles(c, pp, dstf, r.shard_rank, r.shard_rank + 1, -1, r.shard_n, (char*)xbuf, slot_bytes(pp, r.shard_n), true);
        if (rc) return rc;
      }
      int rc = dst_written();    /* a non-reference picture ends here: its tiles stay where they were decoded */
      if (rc) return rc;
      break;
    }
    case 4: {
      int rc = copy_tiles(c, pp, dstf, 0, r.shard_n, r.shard_rank, r.shard_n, (char*)r.xprev, slot_bytes(pp, r.shard_n), false);
      if (rc) return rc;
      rc = dst_written();
      if (rc) return rc;
      r.live_valid = false;
      break;
    }
  }
  r.xprev = xbuf;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(M355_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return M355_OK;
}


/* ------------------------------------------------------------------ sharded picture in one call ---- */

int m355_shard_peers(const m355_pic_params* pp, int rank, int nranks, int* peers, int max_peers)
{
  const int ntc = pp->num_tile_cols, ntr = pp->num_tile_rows, n = ntc * ntr;
  bool is_peer[256] = {};
  if (nranks > 256) return -fail(M355_ERR_INVALID, "more than 256 ranks");
  for (int ty = 0; ty < ntr; ty++)
    for (int tx = 0; tx < ntc; tx++) {
      if (m355_shard_owner_of_tile(ty * ntc + tx, n, nranks) != rank) continue;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int x = tx + dx, y = ty + dy;
          if (x < 0 || y < 0 || x >= ntc || y >= ntr) continue;
          const int q = m355_shard_owner_of_tile(y * ntc + x, n, nranks);
          if (q != rank) is_peer[q] = true;
        }
    }
  int k = 0;
  for (int q = 0; q < nranks; q++) if (is_peer[q]) { if (k < max_peers) peers[k] = q; k++; }
  return k;
}

int m355_shard_set_comm(m355_ctx* c, const m355_comm* comm)
{
  if (comm) c->comm = *comm; else c->comm = m355_comm{nullptr, nullptr, nullptr};
  return M355_OK;
}

/* first sharded decode of these lists: the exchange buffers (zeroed once: a rank's pack kernels write only its own elements, the
   unpack kernels read what the exchange completed), the peers, and one scratch slot per peer */
static int shard_buffers(m355_ctx* c, int h)
{
  Resident& r = c->resident[h];
  if (r.xb[0]) return M355_OK;
  size_t mx = 0;
  for (int k = 0; k < 4; k++) {
    const int64_t b = m355_shard_xbuf_bytes(c, h, k);
    if (b < 0) return M355_ERR_INVALID;
    r.xb_bytes[k] = (size_t)b;
    HIPCHK(hipMalloc(&r.xb[k], (size_t)b + 256));
    HIPCHK(hipMemsetAsync(r.xb[k], 0, (size_t)b + 256, c->stream));
    if (k < 3) mx = std::max(mx, (size_t)b);
  }
  int peers[256];
  const int np = m355_shard_peers(&r.hdr.pp, r.shard_rank, r.shard_n, peers, 256);
  if (np < 0) return M355_ERR_INVALID;
  r.peers.assign(peers, peers + np);
  r.x3_read.assign((size_t)std::max(1, r.shard_n), nullptr);
  { static std::atomic<unsigned long long> epochs{0}; r.xb_epoch = ++epochs; }
  r.xscratch_pitch = (mx + 255) & ~(size_t)255;
  if (np) HIPCHK(hipMalloc(&r.xscratch, (r.xscratch_pitch + 256) * (size_t)np));
  HIPCHK(hipStreamSynchronize(c->stream));
  return M355_OK;
}

int m355_decode_sharded(m355_ctx* c, int h, int gather)
{
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded) return fail(M355_ERR_INVALID, "not a sharded picture handle");
  Resident& r = c->resident[h];
  const int N = r.shard_n;
  if (N > 1 && (!c->comm.halo_sum || !c->comm.all_gather)) return fail(M355_ERR_INVALID, "m355_decode_sharded: no exchange callbacks (m355_shard_set_comm / m355_shard_rccl_init)");
  hipSetDevice(c->device);
  {
    const int rc0 = shard_buffers(c, h);
    if (rc0) return rc0;
  }
  const int last = gather ? 4 : 3;
  const bool ipc = c->ipc && N > 1;                          /* (runtime_ipc.hip: the transport needs to know which buffers it is moving, and two hooks) */
  auto run = [&]() -> int {
    for (int k = 0; k <= last; k++) {
      int rc = 0;
      if (ipc && k == 3 && (rc = ipc_before_repack(c, h, (hipStream_t)m355_stream(c)))) return rc;      /* (phase 3 packs this rank's tiles whether or not they are gathered) */
      rc = m355_decode_phase(c, h, k, k < 4 ? r.xb[k] : nullptr);
      if (rc) return rc;
      if (N <= 1 || k >= last) continue;                       /* a single rank owns every tile: nothing to exchange */
      c->xchg_h = h; c->xchg_k = k;
      if (k < 3) {
        if (!r.peers.empty() && (rc = c->comm.halo_sum(c->comm.user, r.xb[k], r.xb_bytes[k], r.peers.data(), (int)r.peers.size(), r.xscratch, (void*)c->stream)))
          return ipc ? rc : fail(M355_ERR_HIP, "halo exchange %d failed (%d)", k, rc);      /* (the interprocess transport has said what failed) */
      } else if ((rc = c->comm.all_gather(c->comm.user, r.xb[3], r.xb_bytes[3] / (size_t)N, r.shard_rank, N, (void*)c->stream)))
        return ipc ? rc : fail(M355_ERR_HIP, "tile all-gather failed (%d)", rc);
    }
    return M355_OK;
  };
  const int rc = run();
  return ipc ? ipc_end_picture(c, rc) : rc;
}

/* device time of one exchange of a sharded picture's buffers, on its own (bench.py --gpus N: what X0..X3 cost over this transport);
   every rank must call it with the same arguments; the buffers must exist (one m355_decode_sharded of the picture before) */
int m355_shard_time_exchange(m355_ctx* c, int h, int which, int iters, float* ms_each)
{
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded || which < 0 || which > 3 || iters < 1 || !ms_each) return fail(M355_ERR_INVALID, "bad arguments");
  Resident& r = c->resident[h];
  if (!r.xb[which]) return fail(M355_ERR_INVALID, "no exchange buffers yet");
  *ms_each = 0.f;
  if (r.shard_n <= 1) return M355_OK;
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  auto once = [&]() -> int {
    c->xchg_h = h; c->xchg_k = which;
    int rc1 = 0;
    if (which < 3) rc1 = r.peers.empty() ? 0 : c->comm.halo_sum(c->comm.user, r.xb[which], r.xb_bytes[which], r.peers.data(), (int)r.peers.size(), r.xscratch, (void*)c->stream);
    else rc1 = c->comm.all_gather(c->comm.user, r.xb[3], r.xb_bytes[3] / (size_t)r.shard_n, r.shard_rank, r.shard_n, (void*)c->stream);
    return c->ipc ? ipc_end_picture(c, rc1) : rc1;         /* (the interprocess transport counts every exchange round as a picture of its own) */
  };
  int rc = 0;
  for (int i = 0; i < 2 && !rc; i++) rc = once();
  hipEventRecord(e0, c->stream);
  for (int i = 0; i < iters && !rc; i++) rc = once();
  hipEventRecord(e1, c->stream);
  hipError_t he = hipStreamSynchronize(c->stream);
  float ms = 0.f;
  if (he == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc || he != hipSuccess) return fail(M355_ERR_HIP, "exchange %d failed", which);
  *ms_each = ms / (float)iters;
  return M355_OK;
}

/* ---- tile sharding inside ONE process: a group of contexts (one per device, or several on one device) decodes one picture.
 * The exchanges between the phases are copies between the contexts' buffers — hipMemcpyPeerAsync, ordered by events on the
 * contexts' own streams — instead of a collective library: rank r reads what its neighbours packed (their X buffers are untouched
 * until everybody has read them), then adds.  X3: every rank copies the other ranks' finished-tile slots into its gather buffer. ---- */
struct m355_group {
  std::vector<m355_ctx*> ctx;
  /* [rank][exchange 0..3]: recorded behind the rank's pack of that phase / behind its fetch of the neighbours' buffers */
  std::vector<std::array<hipEvent_t, 4>> ev_pack, ev_copied;
  /* One host thread per rank enqueues that rank's phases and exchanges (a single thread issuing every rank's ≈60 calls per picture
     is what bounds a group of 4: 1.46 ms per 8K picture against 0.4 unsharded, profiles/r04_m_*).  The threads meet only where one
     needs an event another has to have RECORDED first: seq_* = (picture number * 8 + exchange + 1) once the event of that exchange is on
     its stream; a reader spins until its peer got there.  Every event is recorded once per picture, and m355_group_decode returns
     only when every rank has enqueued the whole picture, so the next picture's record never overtakes a wait of this one. */
  std::vector<std::thread> th;
  std::vector<std::atomic<unsigned long long>> seq_pack, seq_copied;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  unsigned long long job = 0;          /* picture number (1, 2, ...) the threads are to enqueue */
  int pending = 0;
  bool stop = false;
  const int* handles = nullptr;
  int gather = 0;
  std::vector<int> rc;
  std::vector<std::string> err;
};

/* X3 read-complete handshake, per gather buffer: reader `q` has copied `owner`'s xb[3] (the event is created on the reader's device — the
   current one — and recorded on the reader's stream); the owner's next repack of that buffer waits for every reader's event */
static int x3_mark_read(Resident& owner, int q, int N, hipStream_t st)
{
  if ((int)owner.x3_read.size() < N) return fail(M355_ERR_INVALID, "no exchange buffers yet");
  hipEvent_t& e = owner.x3_read[(size_t)q];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return fail(M355_ERR_HIP, "hipEventCreate failed"); }
  return hipEventRecord(e, st) == hipSuccess ? M355_OK : fail(M355_ERR_HIP, "hipEventRecord failed");
}
static void x3_wait_readers(Resident& me, int r, hipStream_t st)
{
  for (size_t q = 0; q < me.x3_read.size(); q++) if ((int)q != r && me.x3_read[q]) hipStreamWaitEvent(st, me.x3_read[q], 0);
}

/* rank r's share of one picture: phases 0..last with the exchanges between them */
static int group_rank_decode(m355_group* g, int r, unsigned long long n, const int* handles, int gather)
{
  const int N = (int)g->ctx.size();
  m355_ctx* c = g->ctx[(size_t)r];
  hipSetDevice(c->device);
  const int h = handles[r];
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded) return fail(M355_ERR_INVALID, "rank %d: not a sharded picture handle", r);
  if (c->resident[h].shard_n != N || c->resident[h].shard_rank != r) return fail(M355_ERR_INVALID, "rank %d: the picture was uploaded for another group layout", r);
  const int rc0 = shard_buffers(c, h);
  Resident& me = c->resident[h];
  auto published = [&](std::atomic<unsigned long long>& a, int k) { a.store(n * 8 + (unsigned long long)k + 1, std::memory_order_release); };
  auto await = [&](std::atomic<unsigned long long>& a, int k) {
    const unsigned long long want = n * 8 + (unsigned long long)k + 1;
    while (a.load(std::memory_order_acquire) < want) std::this_thread::yield();
  };
  const int last = gather ? 4 : 3;
  int rc = rc0;
  for (int k = 0; k <= last; k++) {
    /* (a rank that failed keeps publishing its steps: the others must not wait for it forever) */
    if (k == 3 && N > 1)
      /* the gather buffer of THIS handle is repacked (phase 3 packs this rank's tiles whether or not they are then gathered): every rank's X3 read of the handle's previous decode has to be over (the halo exchanges in
         between order it only three hops along a rank row).  The events belong to the buffer (Resident::x3_read[reader]); they were recorded
         before that m355_group_decode returned — a never-recorded one is no wait */
      x3_wait_readers(me, r, (hipStream_t)m355_stream(c));
    if (!rc) rc = m355_decode_phase(c, h, k, k < 4 ? me.xb[k] : nullptr);
    if (N <= 1 || k >= last) continue;
    hipStream_t st = (hipStream_t)m355_stream(c);
    if (!rc) hipEventRecord(g->ev_pack[(size_t)r][(size_t)k], st);
    published(g->seq_pack[(size_t)r], k);
    if (k < 3) {
      /* step 1: fetch the neighbours' buffers as they packed them into this rank's scratch slots */
      for (size_t i = 0; i < me.peers.size() && !rc; i++) {
        const int q = me.peers[i];
        await(g->seq_pack[(size_t)q], k);
        Resident& other = g->ctx[(size_t)q]->resident[handles[q]];
        if (!other.xb[k]) { rc = fail(M355_ERR_INVALID, "rank %d has no exchange buffers", q); break; }
        hipStreamWaitEvent(st, g->ev_pack[(size_t)q][(size_t)k], 0);
        if (hipMemcpyPeerAsync((char*)me.xscratch + me.xscratch_pitch * i, c->device, other.xb[k], g->ctx[(size_t)q]->device, me.xb_bytes[k], st) != hipSuccess) rc = fail(M355_ERR_HIP, "hipMemcpyPeerAsync failed");
      }
      if (!rc) hipEventRecord(g->ev_copied[(size_t)r][(size_t)k], st);
      published(g->seq_copied[(size_t)r], k);
      /* step 2: add them once this rank's own buffer has been read by all of them */
      if (!rc && !me.peers.empty()) {
        for (int q : me.peers) { await(g->seq_copied[(size_t)q], k); hipStreamWaitEvent(st, g->ev_copied[(size_t)q][(size_t)k], 0); }
        m355_launch_halo_add((uint32_t*)me.xb[k], (const uint32_t*)me.xscratch, (uint32_t)(me.xscratch_pitch / 4), (int)me.peers.size(), (uint32_t)((me.xb_bytes[k] + 3) / 4), st);
      }
    } else {
      /* X3: the other ranks' finished tiles, slot by slot, straight out of their gather buffers */
      const size_t slot = me.xb_bytes[3] / (size_t)N;
      for (int q = 0; q < N && !rc; q++) {
        if (q == r) continue;
        await(g->seq_pack[(size_t)q], 3);
        Resident& other = g->ctx[(size_t)q]->resident[handles[q]];
        if (!other.xb[3]) { rc = fail(M355_ERR_INVALID, "rank %d has no exchange buffers", q); break; }
        hipStreamWaitEvent(st, g->ev_pack[(size_t)q][3], 0);
        if (hipMemcpyPeerAsync((char*)me.xb[3] + slot * (size_t)q, c->device, (const char*)other.xb[3] + slot * (size_t)q, g->ctx[(size_t)q]->device, slot, st) != hipSuccess) rc = fail(M355_ERR_HIP, "hipMemcpyPeerAsync failed");
      }
      /* this rank has read the others' gather buffers: said per buffer (see phase 3 above) */
      for (int q = 0; q < N && !rc; q++) if (q != r) rc = x3_mark_read(g->ctx[(size_t)q]->resident[handles[q]], r, N, st);
    }
  }
  return rc;
}

static void group_thread(m355_group* g, int r)
{
  unsigned long long seen = 0;
  for (;;) {
    const int* handles; int gather; unsigned long long n;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_go.wait(lk, [&]() { return g->stop || g->job != seen; });
      if (g->stop) return;
      n = seen = g->job; handles = g->handles; gather = g->gather;
    }
    const int rc = group_rank_decode(g, r, n, handles, gather);
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->rc[(size_t)r] = rc;
      if (rc) g->err[(size_t)r] = g_err;
      if (--g->pending == 0) g->cv_done.notify_all();
    }
  }
}

int m355_group_create(m355_ctx* const* ctxs, int n, m355_group** out)
{
  if (!ctxs || n < 1 || n > 256 || !out) return fail(M355_ERR_INVALID, "bad group");
  m355_group* g = new m355_group;
  for (int r = 0; r < n; r++) {
    if (!ctxs[r]) { delete g; return fail(M355_ERR_INVALID, "null context in group"); }
    g->ctx.push_back(ctxs[r]);
    int rc = m355_shard_set(ctxs[r], r, n);
    if (rc) { delete g; return rc; }
    m355_shard_set_comm(ctxs[r], nullptr);
  }
  g->ev_pack.resize((size_t)n); g->ev_copied.resize((size_t)n);
  for (int r = 0; r < n; r++) for (int k = 0; k < 4; k++) { g->ev_pack[(size_t)r][(size_t)k] = nullptr; g->ev_copied[(size_t)r][(size_t)k] = nullptr; }
  g->seq_pack = std::vector<std::atomic<unsigned long long>>((size_t)n);
  g->seq_copied = std::vector<std::atomic<unsigned long long>>((size_t)n);
  for (int r = 0; r < n; r++) { g->seq_pack[(size_t)r].store(0); g->seq_copied[(size_t)r].store(0); }
  g->rc.assign((size_t)n, 0); g->err.assign((size_t)n, std::string());
  for (int r = 0; r < n; r++) {
    hipSetDevice(ctxs[r]->device);
    for (int k = 0; k < 4; k++)
      if (hipEventCreateWithFlags(&g->ev_pack[(size_t)r][(size_t)k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->ev_copied[(size_t)r][(size_t)k], hipEventDisableTiming) != hipSuccess) {
        m355_group_destroy(g);
        return fail(M355_ERR_HIP, "hipEventCreate failed");
      }
  }
  bool threads = n > 1;
#ifdef SIMT_EMU
  threads = false;
#endif
  if (threads) for (int r = 0; r < n; r++) g->th.emplace_back(group_thread, g, r);
  *out = g;
  return M355_OK;
}

void m355_group_destroy(m355_group* g)
{
  if (!g) return;
  { std::lock_guard<std::mutex> lk(g->mu); g->stop = true; }
  g->cv_go.notify_all();
  for (auto& t : g->th) t.join();
  for (size_t r = 0; r < g->ctx.size() && r < g->ev_pack.size(); r++) {
    hipSetDevice(g->ctx[r]->device);
    for (int k = 0; k < 4; k++) {
      if (g->ev_pack[r][(size_t)k]) hipEventDestroy(g->ev_pack[r][(size_t)k]);
      if (g->ev_copied[r][(size_t)k]) hipEventDestroy(g->ev_copied[r][(size_t)k]);
    }
  }
  delete g;
}

/* the same picture enqueued by ONE thread, rank after rank in lockstep (M355_GROUP_THREADS=0, and the SIMT interpreter of the CPU
   test tier, whose launches are not thread-safe) */
static int group_decode_lockstep(m355_group* g, const int* handles, int gather)
{
  const int N = (int)g->ctx.size();
  std::vector<Resident*> R((size_t)N);
  for (int r = 0; r < N; r++) {
    m355_ctx* c = g->ctx[(size_t)r];
    const int h = handles[r];
    if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded) return fail(M355_ERR_INVALID, "rank %d: not a sharded picture handle", r);
    if (c->resident[h].shard_n != N || c->resident[h].shard_rank != r) return fail(M355_ERR_INVALID, "rank %d: the picture was uploaded for another group layout", r);
    hipSetDevice(c->device);
    const int rc = shard_buffers(c, h);
    if (rc) return rc;
    R[(size_t)r] = &c->resident[h];
  }
  const int last = gather ? 4 : 3;
  for (int k = 0; k <= last; k++) {
    for (int r = 0; r < N; r++) {
      m355_ctx* c = g->ctx[(size_t)r];
      hipSetDevice(c->device);
      if (k == 3 && N > 1) x3_wait_readers(*R[(size_t)r], r, (hipStream_t)m355_stream(c));   /* (as group_rank_decode) */
      const int rc = m355_decode_phase(c, handles[r], k, k < 4 ? R[(size_t)r]->xb[k] : nullptr);
      if (rc) return rc;
      if (N > 1 && k < last) hipEventRecord(g->ev_pack[(size_t)r][(size_t)k], (hipStream_t)m355_stream(c));
    }
    if (N <= 1 || k >= last) continue;
    for (int step = 0; step < (k < 3 ? 2 : 1); step++)
      for (int r = 0; r < N; r++) {
        m355_ctx* c = g->ctx[(size_t)r];
        Resident& me = *R[(size_t)r];
        hipSetDevice(c->device);
        hipStream_t st = (hipStream_t)m355_stream(c);
        if (k < 3 && step == 0) {
          for (size_t i = 0; i < me.peers.size(); i++) {
            const int q = me.peers[i];
            hipStreamWaitEvent(st, g->ev_pack[(size_t)q][(size_t)k], 0);
            HIPCHK(hipMemcpyPeerAsync((char*)me.xscratch + me.xscratch_pitch * i, c->device, R[(size_t)q]->xb[k], g->ctx[(size_t)q]->device, me.xb_bytes[k], st));
          }
          hipEventRecord(g->ev_copied[(size_t)r][(size_t)k], st);
        } else if (k < 3) {
          if (me.peers.empty()) continue;
          for (int q : me.peers) hipStreamWaitEvent(st, g->ev_copied[(size_t)q][(size_t)k], 0);
          m355_launch_halo_add((uint32_t*)me.xb[k], (const uint32_t*)me.xscratch, (uint32_t)(me.xscratch_pitch / 4), (int)me.peers.size(), (uint32_t)((me.xb_bytes[k] + 3) / 4), st);
        } else {
          const size_t slot = me.xb_bytes[3] / (size_t)N;
          for (int q = 0; q < N; q++) {
            if (q == r) continue;
            hipStreamWaitEvent(st, g->ev_pack[(size_t)q][3], 0);
            HIPCHK(hipMemcpyPeerAsync((char*)me.xb[3] + slot * (size_t)q, c->device, (const char*)R[(size_t)q]->xb[3] + slot * (size_t)q, g->ctx[(size_t)q]->device, slot, st));
          }
          for (int q = 0; q < N; q++) if (q != r) { const int rcx = x3_mark_read(*R[(size_t)q], r, N, st); if (rcx) return rcx; }
        }
      }
  }
  return M355_OK;
}

int m355_group_decode(m355_group* g, const int* handles, int gather)
{
  if (!g || !handles) return fail(M355_ERR_INVALID, "bad arguments");
  const int N = (int)g->ctx.size();
  /* every rank's handle and layout is checked BEFORE any rank starts: a rank thread that left early would never publish its steps, and
     its neighbours would wait for them forever (the rank threads only tolerate failures behind this point: they keep publishing) */
  for (int r = 0; r < N; r++) {
    m355_ctx* c = g->ctx[(size_t)r];
    const int h = handles[r];
    if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used || !c->resident[h].sharded) return fail(M355_ERR_INVALID, "rank %d: not a sharded picture handle", r);
    if (c->resident[h].shard_n != N || c->resident[h].shard_rank != r) return fail(M355_ERR_INVALID, "rank %d: the picture was uploaded for another group layout", r);
  }
  if (g->th.empty()) return group_decode_lockstep(g, handles, gather);
  std::unique_lock<std::mutex> lk(g->mu);
  g->handles = handles; g->gather = gather; g->pending = N; g->job++;
  g->cv_go.notify_all();
  g->cv_done.wait(lk, [&]() { return g->pending == 0; });
  for (int r = 0; r < N; r++)
    if (g->rc[(size_t)r]) { g_err = g->err[(size_t)r]; return g->rc[(size_t)r]; }
  return M355_OK;
}

/* ---- built-in RCCL transport (struct Rccl above: librccl is loaded on demand, the library itself does not link against it) ---- */
static int rccl_load(Rccl& R)
{
  if (R.so) return M355_OK;
  /* resolved into a local copy and committed only when every symbol is there: a partial table must never look loaded */
  Rccl L;
  L.so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!L.so) L.so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!L.so) return fail(M355_ERR_HIP, "cannot load librccl.so: %s", dlerror());
#define RSYM(field, name) L.field = (decltype(L.field))dlsym(L.so, name); if (!L.field) { dlclose(L.so); return fail(M355_ERR_HIP, "librccl lacks %s", name); }
  RSYM(GetUniqueId, "ncclGetUniqueId") RSYM(CommInitRank, "ncclCommInitRank") RSYM(CommDestroy, "ncclCommDestroy") RSYM(GroupStart, "ncclGroupStart")
  RSYM(GroupEnd, "ncclGroupEnd") RSYM(Send, "ncclSend") RSYM(Recv, "ncclRecv") RSYM(AllGather, "ncclAllGather")
#undef RSYM
  R = L;
  return M355_OK;
}
static int rccl_halo_sum(void* user, void* buf, size_t bytes, const int* peers, int n_peers, void* scratch, void* stream)
{
  Rccl& R = g_rccl;
  void* comm = ((m355_ctx*)user)->rccl;
  const size_t pitch = (bytes + 255) & ~(size_t)255;
  int rc = R.GroupStart();
  for (int i = 0; i < n_peers && !rc; i++) {
    rc = R.Send(buf, bytes, /* ncclInt8 */ 0, peers[i], comm, stream);
    if (!rc) rc = R.Recv((char*)scratch + pitch * (size_t)i, bytes, 0, peers[i], comm, stream);
  }
  const int rc2 = R.GroupEnd();
  if (rc || rc2) return rc ? rc : rc2;
  m355_launch_halo_add((uint32_t*)buf, (const uint32_t*)scratch, (uint32_t)(pitch / 4), n_peers, (uint32_t)((bytes + 3) / 4), (hipStream_t)stream);
  return 0;
}
static int rccl_all_gather(void* user, void* buf, size_t slot_bytes, int rank, int nranks, void* stream)
{
  (void)nranks;
  return g_rccl.AllGather((const char*)buf + slot_bytes * (size_t)rank, buf, slot_bytes, 0, ((m355_ctx*)user)->rccl, stream);
}

int m355_rccl_unique_id(void* out128)
{
  int rc = rccl_load(g_rccl);
  if (rc) return rc;
  if (g_rccl.GetUniqueId(out128)) return fail(M355_ERR_HIP, "ncclGetUniqueId failed");
  return M355_OK;
}

int m355_shard_rccl_init(m355_ctx* c, const void* id128, int rank, int nranks)
{
  int rc = rccl_load(g_rccl);
  if (rc) return rc;
  if ((rc = m355_shard_set(c, rank, nranks))) return rc;
  hipSetDevice(c->device);
  Id128 id;
  memcpy(&id, id128, 128);
  if (c->rccl) { g_rccl.CommDestroy(c->rccl); c->rccl = nullptr; }
  if (g_rccl.CommInitRank(&c->rccl, nranks, id, rank)) return fail(M355_ERR_HIP, "ncclCommInitRank failed");
  m355_comm cm = {c, rccl_halo_sum, rccl_all_gather};
  return m355_shard_set_comm(c, &cm);
}

/* Moves real bytes through the built-in RCCL transport on this context's communicator and checks them on the host: the halo
 * exchange (ncclSend / ncclRecv grouped per peer + k_halo_add) with every OTHER rank as peer — or, in a communicator of one
 * rank, with itself (a grouped self-send) — and the in-place all-gather.  Every rank calls it alike.  What each rank sends is
 * a function of (rank, word index), so the sums and the gathered slots are known everywhere. */
int m355_shard_rccl_selftest(m355_ctx* c, size_t words)
{
  if (!c->rccl || c->shard_n < 1) return fail(M355_ERR_INVALID, "no RCCL communicator (m355_shard_rccl_init)");
  if (words < 1 || words > (1u << 24)) return fail(M355_ERR_INVALID, "bad size");
  hipSetDevice(c->device);
  const int N = c->shard_n, me = c->shard_rank;
  std::vector<int> peers;
  for (int q = 0; q < N; q++) if (q != me) peers.push_back(q);
  if (peers.empty()) peers.push_back(me);                   /* one rank: send to / receive from itself */
  const size_t bytes = words * 4, pitch = (bytes + 255) & ~(size_t)255;
  auto val = [](int rank, size_t i) { return (uint32_t)(rank + 1) * 0x01000193u + (uint32_t)i * 2654435761u; };
  uint32_t *buf = nullptr, *scratch = nullptr, *gat = nullptr;
  HIPCHK(hipMalloc(&buf, bytes + 256));
  HIPCHK(hipMalloc(&scratch, pitch * peers.size() + 256));
  HIPCHK(hipMalloc(&gat, bytes * (size_t)N + 256));
  std::vector<uint32_t> h(words), back(words * (size_t)N);
  for (size_t i = 0; i < words; i++) h[i] = val(me, i);
  int rc = M355_OK;
  do {
    if (hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(gat + words * (size_t)me, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(M355_ERR_HIP, "copy failed"); break; }
    int e = rccl_halo_sum(c, buf, bytes, peers.data(), (int)peers.size(), scratch, (void*)c->stream);
    if (e) { rc = fail(M355_ERR_HIP, "halo exchange over RCCL failed (%d)", e); break; }
    e = rccl_all_gather(c, gat, bytes, me, N, (void*)c->stream);
    if (e) { rc = fail(M355_ERR_HIP, "ncclAllGather failed (%d)", e); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(M355_ERR_HIP, "the exchange did not complete: %s", hipGetErrorString(hipGetLastError())); break; }
    if (hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(back.data(), gat, bytes * (size_t)N, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(M355_ERR_HIP, "copy failed"); break; }
    for (size_t i = 0; i < words && !rc; i++) {
      uint32_t want = val(me, i);
      for (int q : peers) want += val(q, i);
      if (h[i] != want) rc = fail(M355_ERR_HIP, "halo sum: word %zu is %08x, expected %08x", i, h[i], want);
    }
    for (int q = 0; q < N && !rc; q++)
      for (size_t i = 0; i < words && !rc; i++)
        if (back[(size_t)q * words + i] != val(q, i)) rc = fail(M355_ERR_HIP, "all-gather: slot %d word %zu is %08x, expected %08x", q, i, back[(size_t)q * words + i], val(q, i));
  } while (0);
  hipFree(buf); hipFree(scratch); hipFree(gat);
  return rc;
}

} /* extern "C" */
