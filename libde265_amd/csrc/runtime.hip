/*
 * runtime.hip — host side of the picture layer, part 1 of 4 (runtime_internal.h lists the parts): context, lanes, events,
 * device-resident frames (the DPB lives in HBM), arenas, submit / wait, timing.
 *
 * Per picture the executor enqueues, on the context's own HIP stream:
 *   H2D (one pinned arena copy)  ->  k_meta_*  ->  k_inter  ->  k_residual<2..5>  ->  k_intra
 *   ->  k_deblock<V>  ->  k_deblock<H>  ->  k_sao
 * which is the deferred form of decode_TU / decode_prediction_unit / run_postprocessing_filters_*
 * (slice.cc:3460, motion.cc:2190, decctx.cc:1783-1833).  Nothing here falls back to the CPU: if HIP
 * is unavailable every entry point fails with M355_ERR_NO_DEVICE.
 */
#include "runtime_internal.h"

thread_local std::string g_err;
int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}

extern "C" {
/* mark the point the stream has reached (one event packet); -> *out */
int ev_mark(m355_ctx* c, hipStream_t st, EvRef* out) {
  const unsigned long long t = ++c->ev_ticket;
  m355_ctx::EvSlot& e = c->evring[t % M355_EV_RING];
  if (!e.ev) { if (hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) return fail(M355_ERR_HIP, "hipEventCreate failed"); }
  else if (e.ticket) hipEventSynchronize(e.ev);            /* the slot's old mark, M355_EV_RING marks ago (passed long since: this is the ring's invariant, not a wait) */
  if (hipEventRecord(e.ev, st) != hipSuccess) return fail(M355_ERR_HIP, "hipEventRecord failed");
  e.ticket = t;
  out->ticket = t; out->stream = st;
  return M355_OK;
}
/* `st` continues behind the mark: nothing to enqueue when the mark has passed or lies on `st` itself (stream order) */
void ev_wait(m355_ctx* c, hipStream_t st, const EvRef& r) {
  if (!r.ticket || r.stream == st) return;
  const m355_ctx::EvSlot& e = c->evring[r.ticket % M355_EV_RING];
  if (e.ticket == r.ticket) hipStreamWaitEvent(st, e.ev, 0);
}
/* the host waits for the mark / asks whether it has passed */
hipError_t ev_sync(m355_ctx* c, const EvRef& r) {
  if (!r.ticket) return hipSuccess;
  const m355_ctx::EvSlot& e = c->evring[r.ticket % M355_EV_RING];
  return e.ticket == r.ticket ? hipEventSynchronize(e.ev) : hipSuccess;
}
hipError_t ev_query(m355_ctx* c, const EvRef& r) {
  if (!r.ticket) return hipSuccess;
  const m355_ctx::EvSlot& e = c->evring[r.ticket % M355_EV_RING];
  return e.ticket == r.ticket ? hipEventQuery(e.ev) : hipSuccess;
}

void select_lane(m355_ctx* c, int lane) {
  if (lane == c->active) return;
#define PARK_FIELD(f) c->lanes[c->active].f = c->f;
#define LOAD_FIELD(f) c->f = c->lanes[lane].f;
  LANE_FIELDS(PARK_FIELD)
  LANE_FIELDS(LOAD_FIELD)
#undef PARK_FIELD
#undef LOAD_FIELD
  c->active = lane;
}
/* The HIP runtime multiplexes its streams onto a few hardware queues PER STREAM PRIORITY (GPU_MAX_HW_QUEUES, default 4), and
 * kernels of different streams that share a hardware queue mostly run one after the other.  Three lanes (six streams) do well on
 * the default priority's queues.  Every further group of three lanes belongs to the next priority class, whose streams have
 * hardware queues of their own — used by INTRA PICTURES only (they keep to one stream, launch_prediction, and their k_intra is
 * what gains from more pictures in flight: 1080p, nine lanes 0.84 -> 0.334 ms per picture, profiles/r03_v_*, r03_x_c2_in_flight):
 * such a picture runs on its lane's stream_hi.  Inter pictures stay on the default-priority streams: every queue beyond the first
 * few slows their short kernels down (8K at depth 4: 0.436 -> 0.466 ms with all streams in classes).
 * M355_LANE_PRIORITIES=0: no classes at all; =1: ALL streams of lanes 3.. in their class (the measurement above). */
int lane_class_priority(int index) {
  static int lo = 0, hi = 0, probed = 0;
  if (!probed) { probed = 1; if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0; }   /* (least, greatest) */
  const int cls = (index / 3) % 3;
  return cls == 0 ? 0 : (cls == 1 ? hi : lo);
}
int lane_priorities_mode() { return 2; }              /* 0 off, 1 every stream, 2 intra pictures only (what the measurements of round 3 left: profiles/r03_v_*) */
static int lane_priority(int index) { return lane_priorities_mode() == 1 ? lane_class_priority(index) : 0; }
/* (the ORDER in which the streams are created is load-bearing: the runtime deals its streams round its four hardware queues in creation order, kernels of streams
   that share a queue run one after the other, and which lanes share decides how three pictures' stages interleave — main, side, main, side, main, side (lanes 0 and 2 on
   one pair of queues, lane 1 on the other) is the second best of the 187 ways of dealing six streams to four queues, 1 % behind the best and up to 37 % ahead of the
   others at C5: tools/qmap_search.py, profiles/r06_v49_lane_stream_queue_mapping.txt) */
static int lane_create(m355_ctx* c, Lane& l, int index)
{
  HIPCHK(hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, lane_priority(index)));
  HIPCHK(hipStreamCreateWithPriority(&l.stream2, hipStreamNonBlocking, lane_priority(index)));
  HIPCHK(hipEventCreateWithFlags(&l.ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&l.ev_fork2, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&l.ev_join, hipEventDisableTiming));
  HIPCHK(hipMalloc(&l.ticket, 64));
  HIPCHK(hipMalloc(&l.timeout, 128));
  HIPCHK(hipMemsetAsync(l.ticket, 0, 64, l.stream));
  HIPCHK(hipMemsetAsync(l.timeout, 0, 128, l.stream));
  HIPCHK(hipMemsetAsync(l.timeout + 2, 0xFF, 8, l.stream));     /* rejected record of the latest rejected decode: none */
  HIPCHK(hipStreamSynchronize(l.stream));
  return M355_OK;
}
static void lane_destroy(Lane& l)
{
  if (l.stream) hipStreamSynchronize(l.stream);
  if (l.stream2) hipStreamSynchronize(l.stream2);
  if (l.work.used) frame_free(l.work);
  void* bufs[] = {l.pb_of, l.edge, l.ticket, l.timeout, l.edge_tu, l.cuf, l.resbuf, l.jobs, l.sao_nb, l.iplan, l.job_base};
  for (void* b : bufs) if (b) hipFree(b);
  if (l.ev_fork) hipEventDestroy(l.ev_fork);
  if (l.ev_fork2) hipEventDestroy(l.ev_fork2);
  if (l.ev_join) hipEventDestroy(l.ev_join);
  if (l.stream_hi) { hipStreamSynchronize(l.stream_hi); hipStreamDestroy(l.stream_hi); }
  if (l.stream2) hipStreamDestroy(l.stream2);
  if (l.stream) hipStreamDestroy(l.stream);
  l = Lane();
}
/* all work of the context, on every lane */
hipError_t sync_all(m355_ctx* c) {
  hipError_t e = hipStreamSynchronize(c->stream);
  for (int k = 0; k < M355_MAX_LANES; k++)
    if (k != c->active && c->lanes[k].stream) { hipError_t e2 = hipStreamSynchronize(c->lanes[k].stream); if (e == hipSuccess) e = e2; }
  if (c->stream_hi) { hipError_t e2 = hipStreamSynchronize(c->stream_hi); if (e == hipSuccess) e = e2; }
  for (hipStream_t bs : c->batch_stream) if (bs) { hipError_t e2 = hipStreamSynchronize(bs); if (e == hipSuccess) e = e2; }
  for (int k = 0; k < M355_MAX_LANES; k++)
    if (k != c->active && c->lanes[k].stream_hi) { hipError_t e2 = hipStreamSynchronize(c->lanes[k].stream_hi); if (e == hipSuccess) e = e2; }
  return e;
}


static std::vector<TileRect> rank_tiles(const m355_pic_params& pp, int rank, int nranks)
{
  std::vector<TileRect> v;
  const int cs = 1 << pp.log2_ctb_size, n_tiles = pp.num_tile_cols * pp.num_tile_rows;
  for (int ty = 0, t = 0; ty < pp.num_tile_rows; ty++)
    for (int tx = 0; tx < pp.num_tile_cols; tx++, t++) {
      if (m355_shard_owner_of_tile(t, n_tiles, nranks) != rank) continue;
      TileRect r = {pp.col_bd[tx] * cs, pp.row_bd[ty] * cs, pp.col_bd[tx + 1] * cs, pp.row_bd[ty + 1] * cs};
      if (r.x1 > pp.width) r.x1 = pp.width;
      if (r.y1 > pp.height) r.y1 = pp.height;
      v.push_back(r);
    }
  return v;
}
static size_t tiles_bytes(const m355_pic_params& pp, const std::vector<TileRect>& v)
{
  const int cf = pp.chroma_format_idc;
  const int sw = (cf == 1 || cf == 2) ? 2 : 1, sh = cf == 1 ? 2 : 1;
  const size_t bl = pp.bit_depth_luma <= 8 ? 1 : 2, bc = pp.bit_depth_chroma <= 8 ? 1 : 2;
  size_t n = 0;
  for (const TileRect& r : v) {
    n += (size_t)(r.x1 - r.x0) * (r.y1 - r.y0) * bl;
    if (cf) n += 2 * (size_t)((r.x1 - r.x0) / sw) * ((r.y1 - r.y0) / sh) * bc;
  }
  return (n + 255) & ~(size_t)255;
}
size_t slot_bytes(const m355_pic_params& pp, int nranks) {
  size_t m = 0;
  for (int k = 0; k < nranks; k++) { const size_t b = tiles_bytes(pp, rank_tiles(pp, k, nranks)); if (b > m) m = b; }
  return m;
}
/* copy the tiles of ranks [k0, k1) except `skip` between the frame planes and their slots of the all-gather buffer (to_slot) or
   back: all rectangles in as few launches as the argument block allows */
int copy_tiles(m355_ctx* c, const m355_pic_params& pp, Frame* f, int k0, int k1, int skip, int nranks, char* xbuf, size_t slot, bool to_slot) {
  const int cf = pp.chroma_format_idc;
  const int sw = (cf == 1 || cf == 2) ? 2 : 1, sh = cf == 1 ? 2 : 1;
  TileCopyArgs a;
  for (int cc = 0; cc < 3; cc++) { a.plane[cc] = (char*)f->plane[cc]; a.pitch[cc] = (size_t)f->stride[cc] * f->bpp[cc]; }
  int n = 0;
  for (int k = k0; k < k1; k++) {
    if (k == skip) continue;
    size_t o = slot * (size_t)k;
    for (const TileRect& r : rank_tiles(pp, k, nranks))
      for (int cc = 0; cc < 3; cc++) {
        if (cc && !cf) continue;
        const int x = cc ? r.x0 / sw : r.x0, y = cc ? r.y0 / sh : r.y0;
        const int w = cc ? (r.x1 - r.x0) / sw : r.x1 - r.x0, h = cc ? (r.y1 - r.y0) / sh : r.y1 - r.y0;
        const size_t bpp = f->bpp[cc], wb = (size_t)w * bpp;
        if (w > 0 && h > 0) {
          if ((wb | (x * bpp)) & 3) return fail(M355_ERR_INVALID, "tile rectangle is not a whole number of 32-bit words");
          TileCopyRect& t = a.r[n++];
          t.plane = (uint32_t)cc; t.xb = (uint32_t)(x * bpp); t.y = (uint32_t)y; t.wb = (uint32_t)wb; t.h = (uint32_t)h; t.pad = 0; t.ofs = o;
          if (n == M355_TILE_COPY_RECTS) { m355_launch_tiles_copy(a, n, xbuf, to_slot, c->stream); n = 0; }
        }
        o += wb * h;
      }
  }
  m355_launch_tiles_copy(a, n, xbuf, to_slot, c->stream);
  return M355_OK;
}



const char* m355_last_error(void) { return g_err.c_str(); }
const char* m355_version(void) { return "libde265_mi355x 0.1 (gfx950)"; }
int m355_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int m355_create(int device, m355_ctx** out)
{
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(M355_ERR_NO_DEVICE, "no HIP device visible (the MI355X backend has no CPU fallback)");
  if (device < 0 || device >= n) return fail(M355_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
  HIPCHK(hipSetDevice(device));
  m355_ctx* c = new m355_ctx;
  c->device = device;
  {
    Lane l;
    int rc = lane_create(c, l, 0);
    if (rc) { lane_destroy(l); delete c; return rc; }
#define LOAD_FIELD(f) c->f = l.f;
    LANE_FIELDS(LOAD_FIELD)                       /* lane 0 is the active one: it lives in the context's own fields */
#undef LOAD_FIELD
  }
  *out = c;
  return M355_OK;
}

/* the exchange buffers of a sharded picture are read by the OTHER ranks' devices (m355_group_*: peer copies on their streams), which a hipFree on this
   device does not wait for: before they go, every device is drained (a rare path: teardown, or a handle re-used for another geometry) */
static void drain_all_devices(int keep_current)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return;
  for (int d = 0; d < n; d++) if (hipSetDevice(d) == hipSuccess) hipDeviceSynchronize();
  hipSetDevice(keep_current);
}

static void resident_free(Resident& r)
{
  if (r.xb[0]) { int cur = 0; hipGetDevice(&cur); drain_all_devices(cur); }
  for (void* b : r.xb) if (b) hipFree(b);
  if (r.xscratch) hipFree(r.xscratch);
  for (hipEvent_t e : r.x3_read) if (e) hipEventDestroy(e);
  if (r.dev) hipFree(r.dev);
  if (r.host) hipHostFree(r.host);
  if (r.refs_dev) hipFree(r.refs_dev);
  if (r.refs_host) hipHostFree(r.refs_host);
  r = Resident();
}

void m355_destroy(m355_ctx* c)
{
  if (!c) return;
  hipSetDevice(c->device);
  sync_all(c);
  if (c->ipc) m355_shard_ipc_close(c);          /* (first: it waits for the other ranks' last reads of this rank's exchange buffers) */
  for (auto& f : c->frames) if (f.used) frame_free(f);
  for (auto& r : c->resident) if (r.used) resident_free(r);
  for (auto& t : c->transient) resident_free(t);
  for (hipEvent_t e : c->evs) hipEventDestroy(e);
  if (c->hash_acc) hipFree(c->hash_acc);
  for (auto& e : c->inter_tabs) hipFree(e.second);
  for (auto& b : c->batch) { if (b.host) hipHostFree(b.host); if (b.dev) hipFree(b.dev); if (b.ev) hipEventDestroy(b.ev); }
  for (hipEvent_t e : c->batch_ev_pre) if (e) hipEventDestroy(e);
  for (hipStream_t bs : c->batch_stream) if (bs) hipStreamDestroy(bs);
  if (c->rccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->rccl);
  for (auto& e_ : c->evring) if (e_.ev) hipEventDestroy(e_.ev);
  if (c->status_words) hipHostFree(c->status_words);
  if (c->stage) hipHostFree(c->stage);
  for (hipEvent_t e : c->dl_evs) if (e) hipEventDestroy(e);
  {
    /* the active lane lives in the context's own fields: collect it into a Lane and destroy both */
    Lane a;
#define MOVE_FIELD(f) a.f = c->f;
    LANE_FIELDS(MOVE_FIELD)
#undef MOVE_FIELD
    lane_destroy(a);
    for (int k = 0; k < M355_MAX_LANES; k++) if (k != c->active) lane_destroy(c->lanes[k]);
  }
  delete c;
}

/* 1: pictures run one after the other on the context's stream (default).  2: consecutive decodes alternate between two
 * lanes (own streams, working planes and scratch) and overlap wherever the frames they touch allow it: a decode waits
 * for the last writer of every reference frame it reads, and — only right before its first write — for the last writer
 * and the readers of its destination frame. */
int m355_set_pipeline_depth(m355_ctx* c, int depth)
{
  if (depth < 1 || depth > M355_MAX_LANES) return fail(M355_ERR_INVALID, "pipeline depth must be 1..%d", M355_MAX_LANES);
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  select_lane(c, 0);
  for (int k = 1; k < depth; k++)
    if (!c->lanes[k].stream) {
      int rc = lane_create(c, c->lanes[k], k);
      if (rc) return rc;
    }
  c->depth = depth;
  return M355_OK;
}

/* the stream the active lane's last decode / phase ran on (an intra picture on lane 3.. runs on the lane's class stream, decode()) */
void* m355_stream(m355_ctx* c) { return (void*)(c->last_stream ? c->last_stream : c->stream); }

/* ------------------------------------------------------------------------------ frames -------- */

int m355_frame_create(m355_ctx* c, int width, int height, int cf, int bdl, int bdc)
{
  if (width <= 0 || height <= 0 || cf < 0 || cf > 3 || bdl < 8 || bdl > 16 || bdc < 8 || bdc > 16) return -fail(M355_ERR_INVALID, "bad frame geometry");
  if ((width & 7) || (height & 7)) return -fail(M355_ERR_INVALID, "frame size %dx%d: HEVC pictures are multiples of the minimum coding block size (>= 8)", width, height);
  if ((bdl <= 8) != (bdc <= 8) && cf != 0) return -fail(M355_ERR_INVALID, "luma/chroma must both be 8-bit or both be 9..16-bit");
  hipSetDevice(c->device);
  int idx = -1;
  for (size_t i = 0; i < c->frames.size(); i++) if (!c->frames[i].used) { idx = (int)i; break; }
  if (idx < 0) { c->frames.push_back(Frame()); idx = (int)c->frames.size() - 1; }
  Frame& f = c->frames[idx];
  f = Frame();
  frame_geometry(f, width, height, cf, bdl, bdc);
  int rc = frame_alloc(f, c->stream);
  if (rc) { frame_free(f); return -rc; }
  /* the zero fill is this frame's first write: whichever lane touches the frame next orders itself after it */
  ev_mark(c, c->stream, &f.wr);
  return idx;
}
Frame* get_frame(m355_ctx* c, int h) {
  if (h < 0 || h >= (int)c->frames.size() || !c->frames[h].used) return nullptr;
  return &c->frames[h];
}
int m355_frame_destroy(m355_ctx* c, int h)
{
  Frame* f = get_frame(c, h);
  if (!f) return fail(M355_ERR_INVALID, "bad frame handle %d", h);
  hipSetDevice(c->device);
  sync_all(c);
  frame_free(*f);
  return M355_OK;
}
/* The blocking transfers of a plane go through a pinned staging buffer of the context and a copy QUEUED ON A STREAM OF THE LIBRARY (the frame's last writer's for a
 * download) — not a blocking hipMemcpy2D between the device and pageable memory on the null stream.  Round 6's soak with 32 processes sharing the GPU and pictures of
 * 1-14 Mi samples (tools/soak_recheck.py, profiles/r06_v35_*): the blocking copies now and then delivered planes with stale / missing rows IN BOTH DIRECTIONS — 46 of
 * 1600 pictures: a reference uploaded wrong (every decode from it then repeats the same wrong samples), or a correct frame downloaded wrong with a different set of
 * samples on every call — while the copies of m355_frame_download_async (pinned planes, the writer's stream) were right every time; alone on the GPU neither fails. */
static int stage_reserve(m355_ctx* c, size_t bytes)
{
  if (bytes <= c->stage_bytes) return M355_OK;
  if (c->stage) { hipHostFree(c->stage); c->stage = nullptr; c->stage_bytes = 0; }
  const size_t want = (bytes + ((size_t)4 << 20) - 1) & ~(((size_t)4 << 20) - 1);
  if (hipHostMalloc(&c->stage, want, hipHostMallocDefault) != hipSuccess) { c->stage = nullptr; return fail(M355_ERR_NOMEM, "staging buffer of %zu bytes", want); }
  c->stage_bytes = want;
  return M355_OK;
}
/* rows of row_bytes bytes: device plane -> host (pitch dst_pitch bytes) */
static int frame_stage_down(m355_ctx* c, Frame* f, int cidx, void* dst, size_t dst_pitch)
{
  const size_t rb = (size_t)f->pw[cidx] * f->bpp[cidx];
  int rc = stage_reserve(c, rb * f->ph[cidx]);
  if (rc) return rc;
  hipStream_t st = f->wr_stream ? f->wr_stream : c->stream;
  HIPCHK(hipMemcpy2DAsync(c->stage, rb, f->plane[cidx], (size_t)f->stride[cidx] * f->bpp[cidx], rb, f->ph[cidx], hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (dst_pitch == rb) memcpy(dst, c->stage, rb * f->ph[cidx]);
  else for (int y = 0; y < f->ph[cidx]; y++) memcpy((uint8_t*)dst + (size_t)y * dst_pitch, (const uint8_t*)c->stage + (size_t)y * rb, rb);
  return M355_OK;
}
static int frame_stage_up(m355_ctx* c, Frame* f, int cidx, const void* src, size_t src_pitch)
{
  const size_t rb = (size_t)f->pw[cidx] * f->bpp[cidx];
  int rc = stage_reserve(c, rb * f->ph[cidx]);
  if (rc) return rc;
  if (src_pitch == rb) memcpy(c->stage, src, rb * f->ph[cidx]);
  else for (int y = 0; y < f->ph[cidx]; y++) memcpy((uint8_t*)c->stage + (size_t)y * rb, (const uint8_t*)src + (size_t)y * src_pitch, rb);
  HIPCHK(hipMemcpy2DAsync(f->plane[cidx], (size_t)f->stride[cidx] * f->bpp[cidx], c->stage, rb, rb, f->ph[cidx], hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return M355_OK;
}
int m355_frame_upload(m355_ctx* c, int h, int cidx, const void* src, ptrdiff_t stride)
{
  Frame* f = get_frame(c, h);
  if (!f || cidx < 0 || cidx > 2 || !f->pw[cidx]) return fail(M355_ERR_INVALID, "bad frame/plane");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  return frame_stage_up(c, f, cidx, src, (size_t)stride * f->bpp[cidx]);
}
int m355_frame_download(m355_ctx* c, int h, int cidx, void* dst, ptrdiff_t stride)
{
  Frame* f = get_frame(c, h);
  if (!f || cidx < 0 || cidx > 2 || !f->pw[cidx]) return fail(M355_ERR_INVALID, "bad frame/plane");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  return frame_stage_down(c, f, cidx, dst, (size_t)stride * f->bpp[cidx]);
}
/* The download of a whole frame, asynchronous: the copies run on the context's own copy stream, behind the frame's last writer and
 * beside the decodes of later pictures; the next picture written into the frame waits for them.  dst planes should be pinned
 * (m355_host_alloc), else the copies are staged by the runtime and block. */
int m355_frame_download_async(m355_ctx* c, int h, void* const dst[3], const ptrdiff_t stride[3])
{
  Frame* f = get_frame(c, h);
  if (!f || !dst || !stride) return fail(M355_ERR_INVALID, "bad frame / destination");
  hipSetDevice(c->device);
  /* The copies go on the stream of the lane that wrote the frame, right behind the decode: measured (tests/test_gpu_pipeline.py,
     profiles/r03_y_*) a copy on a stream of its own, ordered behind the writer by an event (even with the host waiting for that event
     first), now and then read a half-written picture — the writer's last stores were not yet visible to the copy engine; queued on
     the writer's own stream it never did.  They still run beside the host and beside the other lanes' decodes; the lane's next
     picture waits for them. */
  if (c->dl_evs.empty()) {
    std::vector<hipEvent_t> evs(128, nullptr);              /* swapped in only when every event exists */
    for (auto& e : evs)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        for (hipEvent_t x : evs) if (x) hipEventDestroy(x);
        return fail(M355_ERR_HIP, "hipEventCreate failed");
      }
    c->dl_evs.swap(evs);
  }
  hipEvent_t ev_done = c->dl_evs[c->dl_ev_next];            /* (a ring: never re-recorded while an earlier record may still be waited for) */
  c->dl_ev_next = (c->dl_ev_next + 1) % (int)c->dl_evs.size();
  hipStream_t cs = f->wr_stream ? f->wr_stream : c->stream;                     /* (no decode of this context wrote it: uploads and fills are synchronous) */
  if (!f->wr_stream) ev_wait(c, cs, f->wr);
  for (int cc = 0; cc < 3; cc++) {
    if (!f->pw[cc]) continue;
    if (!dst[cc]) return fail(M355_ERR_INVALID, "no destination for plane %d", cc);
    HIPCHK(hipMemcpy2DAsync(dst[cc], (size_t)stride[cc] * f->bpp[cc], f->plane[cc], (size_t)f->stride[cc] * f->bpp[cc],
                            (size_t)f->pw[cc] * f->bpp[cc], f->ph[cc], hipMemcpyDeviceToHost, cs));
  }
  HIPCHK(hipEventRecord(ev_done, cs));
  f->ev_dl = ev_done;                                                           /* (the ring's: not the frame's to destroy) */
  f->dl_pending = true;
  return M355_OK;
}
/* wait (this frame's download only) until the planes handed to m355_frame_download_async hold the picture */
int m355_frame_download_wait(m355_ctx* c, int h)
{
  Frame* f = get_frame(c, h);
  if (!f) return fail(M355_ERR_INVALID, "bad frame handle %d", h);
  if (!f->dl_pending) return M355_OK;
  hipSetDevice(c->device);
  HIPCHK(hipEventSynchronize(f->ev_dl));
  f->dl_pending = false;
  return M355_OK;
}
int m355_frame_fill(m355_ctx* c, int h, int vl, int vc)
{
  Frame* f = get_frame(c, h);
  if (!f) return fail(M355_ERR_INVALID, "bad frame handle %d", h);
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  for (int cc = 0; cc < 3; cc++) {
    if (!f->pw[cc]) continue;
    const size_t n = (size_t)f->stride[cc] * f->ph[cc];
    const int v = cc ? vc : vl;
    if (f->bpp[cc] == 1) { HIPCHK(hipMemsetAsync(f->plane[cc], v, n, c->stream)); HIPCHK(sync_all(c)); }
    else {
      int rc = stage_reserve(c, n * 2);
      if (rc) return rc;
      std::fill((uint16_t*)c->stage, (uint16_t*)c->stage + n, (uint16_t)v);
      HIPCHK(hipMemcpyAsync(f->plane[cc], c->stage, n * 2, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
  }
  return M355_OK;
}

/* float4 copy, grid-stride, streaming stores: the shape of the guide's "6.29 TB/s measured (float4 copy)" figure */
__global__ void __launch_bounds__(256) k_copy_rate(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    d_st_nt8(&dst[i].x, v.x, v.y); d_st_nt8(&dst[i].z, v.z, v.w);
  }
}
int m355_measure_copy_rate(m355_ctx* c, size_t bytes, int iters, double* gbps)
{
  if (!c || !gbps || bytes < (1u << 20) || iters < 1) return fail(M355_ERR_INVALID, "bad arguments");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { if (a) hipFree(a); return fail(M355_ERR_NOMEM, "hipMalloc(%zu) failed", bytes); }
  hipMemsetAsync(a, 0x5A, bytes, c->stream);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 32);
  std::vector<float> ms((size_t)iters, 0.f);
  for (int i = -2; i < iters; i++) {
    hipEventRecord(e0, c->stream);
    hipLaunchKernelGGL(k_copy_rate, dim3(grid), dim3(256), 0, c->stream, (const uint4*)a, (uint4*)b, n16);
    hipEventRecord(e1, c->stream);
    hipEventSynchronize(e1);
    if (i >= 0) hipEventElapsedTime(&ms[(size_t)i], e0, e1);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(a); hipFree(b);
  std::sort(ms.begin(), ms.end());
  const float med = ms[ms.size() / 2];
  *gbps = med > 0.f ? 2.0 * (double)(n16 * 16) / (med * 1e-3) / 1e9 : 0.0;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? M355_OK : fail(M355_ERR_HIP, "copy kernel failed: %s", hipGetErrorString(e));
}

void* m355_host_alloc(size_t bytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(M355_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
void m355_host_free(void* p) { if (p) hipHostFree(p); }

int m355_frame_hash(m355_ctx* c, int h, int type, m355_picture_hash* out)
{
  Frame* f = get_frame(c, h);
  if (!f || !out) return fail(M355_ERR_INVALID, "bad frame handle %d / null result", h);
  if (type != M355_HASH_MD5 && type != M355_HASH_CRC && type != M355_HASH_CHECKSUM) return fail(M355_ERR_INVALID, "bad hash type %d", type);
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  const int np = f->pw[1] ? 3 : 1;
  if (type == M355_HASH_MD5) {
    std::vector<uint8_t> host[3];
    std::vector<std::thread> th;
    for (int cc = 0; cc < np; cc++) {
      const size_t rb = (size_t)f->pw[cc] * f->bpp[cc];
      host[cc].resize(rb * f->ph[cc]);
      { int rc = frame_stage_down(c, f, cc, host[cc].data(), rb); if (rc) { for (auto& t : th) t.join(); return rc; } }
      th.emplace_back([&, cc, rb] { m355_md5_rows(host[cc].data(), rb, (int)rb, f->ph[cc], out->md5[cc]); });
    }
    for (auto& t : th) t.join();
    return M355_OK;
  }
  if (!c->hash_acc) HIPCHK(hipMalloc(&c->hash_acc, 4 * sizeof(uint32_t)));
  HashArgs a = {};
  a.out = c->hash_acc;
  {
    /* enough waves to fill 1024 SIMDs a few times over, each still covering >= 1 row */
    int rows = 0;
    for (int cc = 0; cc < np; cc++) rows += f->ph[cc];
    a.rows_per_wave = std::max(1, rows / 4096);
  }
  int nw = 0;
  for (int cc = 0; cc < 3; cc++) {
    a.first[cc] = nw;
    if (cc >= np) continue;
    a.pl[cc].base = (const uint8_t*)f->plane[cc];
    a.pl[cc].pitch = (size_t)f->stride[cc] * f->bpp[cc];
    a.pl[cc].row_bytes = f->pw[cc] * f->bpp[cc];
    a.pl[cc].h = f->ph[cc];
    a.pl[cc].bpp = f->bpp[cc];
    nw += (f->ph[cc] + a.rows_per_wave - 1) / a.rows_per_wave;
  }
  a.first[3] = nw;
  uint32_t acc[4] = {0, 0, 0, 0};
  HIPCHK(hipMemsetAsync(c->hash_acc, 0, 4 * sizeof(uint32_t), c->stream));
  m355_launch_frame_hash(a, type, c->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(acc, c->hash_acc, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int cc = 0; cc < np; cc++) {
    if (type == M355_HASH_CRC) out->crc[cc] = (uint16_t)(acc[cc] ^ m355_crc_init_term((uint64_t)a.pl[cc].row_bytes * a.pl[cc].h));
    else out->checksum[cc] = acc[cc];
  }
  return M355_OK;
}

/* room for `k` entries per list in the arenas of `r` (grown when it does not fit), list pointers into its pinned half */
static int arena_into(m355_ctx* c, Resident& r, m355_arena_caps* k, int halo_units, bool sharded, m355_picture* pic)
{
  Lay L;
  make_layout(*k, k->n_ctbs, halo_units, sharded, true, L);
  if (L.total > r.cap) {
    if (r.dev || r.host) HIPCHK(sync_all(c));
    if (r.dev) hipFree(r.dev);
    if (r.host) hipHostFree(r.host);
    r.dev = r.host = nullptr;
    r.cap = L.total + L.total / 4;
    HIPCHK(hipMalloc(&r.dev, r.cap));
    HIPCHK(hipHostMalloc(&r.host, r.cap, hipHostMallocDefault));
  } else if (r.done.ticket) {
    HIPCHK(ev_sync(c, r.done));                            /* the last decode of the lists that lived here */
    r.done = EvRef();
  }
  memset(pic, 0, sizeof(*pic));
  pic->slices = (const m355_slice*)(r.host + L.seg[L.i_sl].ofs);
  pic->ctbs = (const m355_ctb*)(r.host + L.seg[L.i_ct].ofs);
  pic->cus = (const m355_cu*)(r.host + L.seg[L.i_cu].ofs);
  pic->tus = (const m355_tu*)(r.host + L.seg[L.i_tu].ofs);
  pic->pbs = (const m355_pb*)(r.host + L.seg[L.i_pb].ofs);
  pic->wts = (const m355_wt*)(r.host + L.seg[L.i_wt].ofs);
  pic->rbs = (const m355_rb*)(r.host + L.seg[L.i_rb[0]].ofs);
  for (int b = 0; b < 4; b++) k->rb_bin[b] = (m355_rb*)(r.host + L.seg[L.i_rb[b]].ofs);
  r.arena = true; r.caps = *k; r.arena_halo_units = halo_units;
  pic->ibs = (const m355_ib*)(r.host + L.seg[L.i_ibin].ofs);
  pic->coeffs = (const uint32_t*)(r.host + L.seg[L.i_co].ofs);
  pic->pcm = (const uint16_t*)(r.host + L.seg[L.i_pc].ofs);
  pic->scaling_factors = k->scaling ? (const uint8_t*)(r.host + L.seg[L.i_sc].ofs) : nullptr;
  pic->dst_frame = -1;
  for (int i = 0; i < M355_MAX_REF_FRAMES; i++) pic->ref_frames[i] = -1;
  return M355_OK;
}
static bool caps_ok(const m355_arena_caps* k, const m355_picture* pic)
{
  return k && pic && k->n_ctbs > 0 && k->n_slices > 0 && k->n_cus >= 0 && k->n_tus >= 0 && k->n_pbs >= 0 && k->n_wts >= 0 && k->n_ibs >= 0 &&
         k->n_rbs[0] >= 0 && k->n_rbs[1] >= 0 && k->n_rbs[2] >= 0 && k->n_rbs[3] >= 0;
}

int m355_arena_begin(m355_ctx* c, m355_arena_caps* k, m355_picture* pic)
{
  if (!caps_ok(k, pic)) return fail(M355_ERR_INVALID, "bad arena capacities");
  if (c->shard_n >= 1) return fail(M355_ERR_INVALID, "m355_arena_begin is not available on a tile-sharded context (its pictures are decoded from handles: m355_picture_arena_begin)");
  hipSetDevice(c->device);
  return arena_into(c, c->transient[c->next_transient], k, 0, false, pic);   /* the arena the next m355_submit_picture uses */
}

/* the same for the arenas of a RESIDENT picture (tile-sharded contexts decode from handles): handle -1 makes one */
int m355_picture_arena_begin(m355_ctx* c, int h, m355_arena_caps* k, const m355_pic_params* pp, m355_picture* pic)
{
  if (!caps_ok(k, pic)) return -fail(M355_ERR_INVALID, "bad arena capacities");
  const bool sharded = c->shard_n >= 1;
  if (sharded && !pp) return -fail(M355_ERR_INVALID, "m355_picture_arena_begin: a tile-sharded context needs the picture parameters (room for the border units of other ranks)");
  int halo_units = 0;
  if (sharded) {
    if (pp->num_tile_cols < 1 || pp->num_tile_rows < 1 || pp->num_tile_cols > M355_MAX_TILE_COLS || pp->num_tile_rows > M355_MAX_TILE_ROWS || pp->width < 8 || pp->height < 8)
      return -fail(M355_ERR_INVALID, "m355_picture_arena_begin: bad picture parameters");
    HaloLayout halo;
    halo_layout(*pp, halo);
    halo_units = halo.n_units;
  }
  hipSetDevice(c->device);
  if (h < 0) {
    for (size_t i = 0; i < c->resident.size(); i++) if (!c->resident[i].used && !c->resident[i].reserved) { h = (int)i; break; }
    if (h < 0) { c->resident.push_back(Resident()); h = (int)c->resident.size() - 1; }
    c->resident[h].reserved = true;
  } else if (h >= (int)c->resident.size() || !(c->resident[h].used || c->resident[h].reserved)) return -fail(M355_ERR_INVALID, "bad picture handle");
  const int rc = arena_into(c, c->resident[h], k, halo_units, sharded, pic);
  return rc ? -rc : h;
}

int m355_submit_picture(m355_ctx* c, const m355_picture* pic)
{
  Resident& t = c->transient[c->next_transient];
  c->next_transient = (c->next_transient + 1) % c->transient_ring();
  /* the lists travel on the stream of the lane that decodes them: the copy of picture k runs beside the kernels of
     picture k-1 on the previous lane (uploading on the lane that is still active would queue it BEHIND those kernels) */
  if (c->depth >= 2) select_lane(c, (c->active + 1) % c->depth);
  static const bool prof = getenv("M355_PROFILE_UPLOAD") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = upload(c, t, pic);
  t.arena = false;                                          /* pointers handed out by m355_arena_begin are spent */
  if (rc) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  rc = decode(c, t, false);
  if (prof) fprintf(stderr, "m355 submit: upload (host phases + copy enqueue) %.3f ms, decode enqueue %.3f ms\n",
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return rc;
}

int m355_wait(m355_ctx* c)
{
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  for (auto& f : c->frames) { f.wr = EvRef(); f.dl_pending = false; for (int k = 0; k < M355_MAX_LANES; k++) f.rd[k] = EvRef(); }   /* everything is complete */
  uint32_t t = 0;
  HIPCHK(hipMemcpy(&t, c->timeout, 4, hipMemcpyDeviceToHost));
  for (int k = 0; k < M355_MAX_LANES; k++)
    if (k != c->active && c->lanes[k].timeout) { uint32_t t2 = 0; HIPCHK(hipMemcpy(&t2, c->lanes[k].timeout, 4, hipMemcpyDeviceToHost)); t |= t2; }
  /* lists checked on the device (recorded in place): the first rejected decode not reported yet (everything has finished) */
  {
    m355_ctx::Status* first = nullptr;
    for (m355_ctx::Status& st_ : c->status)
      if (st_.serial && st_.validated && !st_.reported && c->status_words[4 * (st_.serial % M355_STATUS_RING) + 1] == st_.epoch && (!first || st_.serial < first->serial)) first = &st_;
    if (c->lost_count) {
      const unsigned long long f = c->lost_first; const int n = c->lost_count;
      c->lost_first = 0; c->lost_count = 0;
      return fail(M355_ERR_INVALID, "picture %llu%s rejected by the device-side list validation (not decoded; its status had left the %d-entry ring: %d such picture%s)",
                  f, n > 1 ? " and later ones" : "", M355_STATUS_RING, n, n > 1 ? "s" : "");
    }
    if (first) return status_of(c, *first);
  }
  if (t) {
    hipMemsetAsync(c->timeout, 0, 4, c->stream);
    for (int k = 0; k < M355_MAX_LANES; k++)
      if (k != c->active && c->lanes[k].timeout) hipMemsetAsync(c->lanes[k].timeout, 0, 4, c->lanes[k].stream);
    sync_all(c);
    return fail(M355_ERR_TIMEOUT, "intra wavefront spin bound exceeded");
  }
  return M355_OK;
}

int m355_picture_upload(m355_ctx* c, const m355_picture* pic)
{
  int idx = -1;
  for (size_t i = 0; i < c->resident.size(); i++) if (!c->resident[i].used && !c->resident[i].reserved) { idx = (int)i; break; }
  if (idx < 0) { c->resident.push_back(Resident()); idx = (int)c->resident.size() - 1; }
  int rc = upload(c, c->resident[idx], pic);
  if (rc) { resident_free(c->resident[idx]); return -rc; }
  return idx;
}
/* new lists into the arenas of an uploaded picture (waits for the last decode of the old ones only — no allocation when they
   fit, no synchronisation of the context): how a caller cycles a few handles through a stream of pictures */
int m355_picture_replace(m355_ctx* c, int h, const m355_picture* pic)
{
  if (h < 0 || h >= (int)c->resident.size() || !(c->resident[h].used || c->resident[h].reserved) || !pic) return fail(M355_ERR_INVALID, "bad picture handle");
  Resident& r = c->resident[h];
  hipSetDevice(c->device);
  if (r.done.ticket) { HIPCHK(ev_sync(c, r.done)); r.done = EvRef(); }
  if (r.xb[0] && (memcmp(&r.hdr.pp, &pic->pp, sizeof(pic->pp)) != 0 || r.shard_n != c->shard_n || r.shard_rank != c->shard_rank)) {
    /* the exchange buffers of a sharded picture are sized by its geometry and tile structure */
    if (c->ipc) ipc_before_free(c, h);
    drain_all_devices(c->device);
    for (void*& b : r.xb) { if (b) hipFree(b); b = nullptr; }
    if (r.xscratch) { hipFree(r.xscratch); r.xscratch = nullptr; }
    for (hipEvent_t e : r.x3_read) if (e) hipEventDestroy(e);
    r.x3_read.clear();
    r.peers.clear();
  }
  const int rc = upload(c, r, pic);
  r.arena = false;                                          /* pointers handed out by m355_picture_arena_begin are spent */
  if (!rc) r.reserved = false;
  return rc;
}

int m355_picture_release(m355_ctx* c, int h)
{
  if (h < 0 || h >= (int)c->resident.size() || !(c->resident[h].used || c->resident[h].reserved)) return fail(M355_ERR_INVALID, "bad picture handle");
  hipSetDevice(c->device);
  sync_all(c);
  if (c->ipc && c->resident[h].xb[0]) ipc_before_free(c, h);     /* (the other rank processes' last reads of this handle's gather buffer) */
  resident_free(c->resident[h]);
  return M355_OK;
}
int m355_decode_resident(m355_ctx* c, int h)
{
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used) return fail(M355_ERR_INVALID, "bad picture handle");
  return decode(c, c->resident[h]);
}
int m355_set_stages(m355_ctx* c, int mask) { c->stages = mask & M355_STAGE_ALL; return M355_OK; }

int m355_timing_reset(m355_ctx* c) { c->ev_used = 0; c->timing_on = true; return M355_OK; }

/* averages over every decode enqueued since m355_timing_reset(); waits for them to finish */
int m355_timing_collect(m355_ctx* c, int* n_decodes, float* total_ms, float stage_ms[6])
{
  c->timing_on = false;
  if (c->ev_used == 0) return fail(M355_ERR_INVALID, "m355_timing_collect: no decode was timed — m355_timing_reset opens the timing window, this call (or an earlier one) closes it, and the pictures of m355_decode_batch are never stage-timed");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  double tot = 0, st[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < c->ev_used; k++) {
    hipEvent_t* ev = &c->evs[k * 7];
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[6])); tot += ms;
    /* stage order of the events: [meta, inter, residual, intra, deblock, sao] */
    for (int i = 0; i < 6; i++) { HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); st[i] += ms; }
  }
  if (n_decodes) *n_decodes = c->ev_used;
  if (total_ms) *total_ms = (float)(tot / c->ev_used);
  if (stage_ms) for (int i = 0; i < 6; i++) stage_ms[i] = (float)(st[i] / c->ev_used);
  return M355_OK;
}

} /* extern "C" */
