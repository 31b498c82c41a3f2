/*
 * runtime.hip — host side of the picture layer: context, device-resident frames (the DPB lives in
 * HBM), work-list validation/upload and the per-picture launch sequence.
 *
 * Per picture the executor enqueues, on the context's own HIP stream:
 *   H2D (one pinned arena copy)  ->  k_meta_*  ->  k_inter  ->  k_residual<2..5>  ->  k_intra
 *   ->  k_deblock<V>  ->  k_deblock<H>  ->  k_sao
 * which is the deferred form of decode_TU / decode_prediction_unit / run_postprocessing_filters_*
 * (slice.cc:3460, motion.cc:2190, decctx.cc:1783-1833).  Nothing here falls back to the CPU: if HIP
 * is unavailable every entry point fails with M355_ERR_NO_DEVICE.
 */
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include <pthread.h>
#include "k_common.h"
#include "k_hash.h"

static thread_local std::string g_err;
#ifdef M355_X_PROF
static unsigned long long* g_prof = nullptr;
extern "C" __attribute__((visibility("default"))) int m355_x_prof_read(unsigned long long* out, int n) { return g_prof ? (int)hipMemcpy(out, g_prof, 8 * (size_t)n, hipMemcpyDeviceToHost) : -1; }
#endif
static int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(M355_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

#define M355_STATUS_RING 64
#define M355_BATCH_RING 16 /* m355_decode_batch: picture-record arrays in flight (the host runs this many batches ahead) */
#define M355_MAX_LANES 32  /* pictures in flight per context (m355_set_pipeline_depth) */
#define M355_TRANSIENT_MAX 12 /* staging arenas of m355_submit_picture (m355_ctx::transient_ring) */

/* A MARK = "everything enqueued on `stream` up to here", one event of the context's ring (ev_mark / ev_wait / ev_sync below).  The
 * objects a decode touches — destination and reference frames, its lists, its lane, its status slot — all remember the SAME mark
 * behind its last kernel: one event packet per decode instead of one per object (each costs about 2 us of pipeline time on this
 * runtime, profiles/r04_aj_stage_events_ab.txt). */
struct EvRef { unsigned long long ticket = 0; hipStream_t stream = nullptr; };
#define M355_EV_RING 256

struct Frame {
  bool used = false;
  int w = 0, h = 0, cf = 0, bdl = 0, bdc = 0;
  int pw[3] = {0, 0, 0}, ph[3] = {0, 0, 0}, stride[3] = {0, 0, 0}, bpp[3] = {1, 1, 1};
  void* plane[3] = {nullptr, nullptr, nullptr};
  /* pictures in flight on different lanes (m355_set_pipeline_depth): last writer / last readers per lane */
  EvRef wr, rd[M355_MAX_LANES];
  /* a download in flight on the context's copy stream (m355_frame_download_async): the next writer of the frame waits for it */
  hipEvent_t ev_dl = nullptr;
  bool dl_pending = false;
  hipStream_t wr_stream = nullptr;         /* the stream that last wrote the frame (its downloads are queued on that stream) */
#ifdef M355_X_TILED
  void* tiled[3] = {nullptr, nullptr, nullptr};   /* EXPERIMENT: tiled copy read by k_inter_jobs (k_common.h DevRef) */
  int tiles_w[3] = {0, 0, 0};
  bool tiled_valid = false;
  hipEvent_t ev_tiled = nullptr;
#endif
};

static void frame_geometry(Frame& f, int w, int h, int cf, int bdl, int bdc)
{
  f.w = w; f.h = h; f.cf = cf; f.bdl = bdl; f.bdc = bdc;
  const int sw = (cf == 1 || cf == 2) ? 2 : 1, sh = (cf == 1) ? 2 : 1;
  for (int c = 0; c < 3; c++) {
    f.bpp[c] = ((c ? bdc : bdl) <= 8) ? 1 : 2;
    if (c && cf == 0) { f.pw[c] = f.ph[c] = f.stride[c] = 0; continue; }
    f.pw[c] = c ? w / sw : w;
    f.ph[c] = c ? h / sh : h;
    const int pitch_bytes = (f.pw[c] * f.bpp[c] + 127) & ~127;
    f.stride[c] = pitch_bytes / f.bpp[c];
  }
}
/* NOTE on memsets: hipMemset() on the null stream may return before the fill has run, and the context's
 * stream is non-blocking (it does not order against the null stream) — a fill issued that way can land AFTER
 * kernels launched later on the context's stream (seen with 8 contexts sharing one GPU).  Every fill is
 * therefore enqueued on the context's own stream. */
static int frame_alloc(Frame& f, hipStream_t st)
{
  for (int c = 0; c < 3; c++) {
    if (!f.pw[c]) continue;
    const size_t bytes = (size_t)f.stride[c] * f.ph[c] * f.bpp[c] + 256;
    HIPCHK(hipMalloc(&f.plane[c], bytes));
    HIPCHK(hipMemsetAsync(f.plane[c], 0, bytes, st)); /* planes are zero at allocation (image.cc:164) */
  }
  f.used = true;
  return M355_OK;
}
static void frame_free(Frame& f)
{
  for (int c = 0; c < 3; c++) { if (f.plane[c]) hipFree(f.plane[c]); f.plane[c] = nullptr; }
#ifdef M355_X_TILED
  for (int c = 0; c < 3; c++) { if (f.tiled[c]) hipFree(f.tiled[c]); f.tiled[c] = nullptr; }
  if (f.ev_tiled) hipEventDestroy(f.ev_tiled);
  f.ev_tiled = nullptr; f.tiled_valid = false;
#endif
  f.wr = EvRef();
  for (int k = 0; k < M355_MAX_LANES; k++) f.rd[k] = EvRef();
  f.ev_dl = nullptr; f.dl_pending = false; f.wr_stream = nullptr;
  f.used = false;
}

/* one picture's lists resident in HBM */
struct Resident {
  bool used = false;
  m355_picture hdr;            /* counts + params (pointers are NOT valid) */
  char* dev = nullptr;         /* device arena */
  char* host = nullptr;        /* pinned staging arena */
  size_t cap = 0, bytes = 0;
  DevPic dp;                   /* device pointers filled at upload; frame planes at decode */
  DevRef* refs_dev = nullptr;  /* reference-frame table (device) */
  DevRef* refs_host = nullptr; /* pinned staging + last uploaded contents */
  bool refs_valid = false;
  int n_intra_work = 0;
  uint32_t n_iplan = 0;        /* border-plan entries of the picture's intra blocks (k_intra_plan -> k_intra) */
  /* tile sharding (m355_decode_phase) */
  bool sharded = false;
  int shard_rank = 0, shard_n = 1;
  HaloLayout halo;
  DevPic live;                 /* the descriptor prepared by phase 0, reused by phases 1..4 */
  bool live_sao = false, live_valid = false;
  void* xprev = nullptr;       /* exchange buffer handed to the previous phase */
  int lane = 0;                /* the lane phase 0 ran on: the picture's working planes and scratch live there */
  void* xb[4] = {nullptr, nullptr, nullptr, nullptr};   /* m355_decode_sharded: the picture's exchange buffers X0..X3 + peer scratch (library-owned) */
  size_t xb_bytes[4] = {0, 0, 0, 0};
  void* xscratch = nullptr;
  std::vector<int> peers;      /* ranks this rank exchanges halos with */
  EvRef up;                    /* lists copied to the device (decodes on another lane continue behind it) */
  EvRef done;                  /* last decode of these lists: behind it the arenas may be overwritten */
  bool fresh = false;          /* uploaded and not decoded since: nothing in flight reads its reference table */
  bool arena = false;          /* m355_arena_begin handed out list pointers into `host`: the next upload of lists that sit there copies nothing */
  m355_arena_caps caps;        /* ... with room for this many entries */
  int arena_halo_units = 0;    /* ... and, on a tile-sharded context, for this many foreign border units behind cus[] / pbs[] */
  bool reserved = false;       /* m355_picture_arena_begin made this handle; no lists yet (m355_picture_replace brings them) */
  bool device_validate = false;    /* the record checks of these lists run on the device (k_validate) */
  size_t xscratch_pitch = 0;       /* m355_decode_sharded / m355_group_decode: bytes between the peers' slots of xscratch */
  std::vector<uint8_t> sched_u8;   /* upload(): per-CTB scratch of the intra schedule */
  std::vector<uint32_t> sched_u32;
  std::vector<uint32_t> sched_order, sched_cand, sched_bucket, sched_u32b;   /* ... and of the work list (order, counting-sort buckets, plan bases) */
};

/* Everything ONE picture in flight writes: streams, working planes, metadata / job / residual scratch.  The context's
 * own fields of the same names are the ACTIVE lane; select_lane() exchanges them with a parked copy, so all the code
 * below keeps addressing c->stream, c->work, c->resbuf ... (m355_set_pipeline_depth(ctx, n) decodes consecutive
 * pictures round-robin on n lanes: the dependency-bound tail of one picture's intra stage and its filters overlap
 * the next picture's prediction; frame hazards are ordered with per-frame events). */
struct Lane {
  hipStream_t stream = nullptr, stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join = nullptr;
  /* intra pictures on lanes 3.. run on a stream of the lane's priority class (own hardware queues, lane_class below); the lane's
     scratch is shared by both streams: a decode waits for the lane's previous one when that ran on the other stream */
  hipStream_t stream_hi = nullptr, last_stream = nullptr;
  EvRef last;                  /* behind the lane's last decode */
  Frame work;
  uint32_t *pb_of = nullptr, *ticket = nullptr, *timeout = nullptr;
  unsigned long long* edge = nullptr;   /* k_intra halo granules */
  uint8_t *edge_tu = nullptr, *cuf = nullptr;
  int16_t* resbuf = nullptr;
  uint32_t* jobs = nullptr;
  uint16_t* sao_nb = nullptr;
  uint16_t* iplan = nullptr;   /* border plans of the picture's intra blocks */
  uint32_t* res_map = nullptr; /* fused inter residuals: per component 4x4 unit -> tile piece (k_common.h) */
  uint32_t* job_base = nullptr; /* per 256-PB chunk the first job of each range + the three range ends (k_job_count / k_job_scan) */
  size_t cap_cb = 0, cap_u4 = 0, cap_edge = 0, cap_cuf = 0, cap_res = 0, cap_jobs = 0, cap_sao = 0, cap_iplan = 0, cap_resmap = 0, cap_jobbase = 0;
};

struct m355_ctx {
  int device = 0;
  Lane lanes[M355_MAX_LANES];  /* parked lanes; lanes[active] is stale: the active lane lives in the fields below */
  int depth = 1, active = 0;   /* pipeline depth, index of the active lane */
  std::vector<std::pair<uint32_t, uint32_t*>> inter_tabs;   /* k_inter_jobs' tap tables per (plane type, bit depths): m355_inter_tables */
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;           /* side stream: metadata planes are rasterised while k_inter / k_residual run */
  hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join = nullptr;
  hipStream_t stream_hi = nullptr, last_stream = nullptr;   /* (of the active lane, as in Lane) */
  EvRef last;
  /* the ring of marks (EvRef): a slot is taken over M355_EV_RING marks later, behind a host wait for its old mark — so "the slot
     carries another ticket" means "that mark has passed" */
  struct EvSlot { hipEvent_t ev = nullptr; unsigned long long ticket = 0; };
  EvSlot evring[M355_EV_RING];
  unsigned long long ev_ticket = 0;
  std::vector<hipEvent_t> dl_evs;          /* m355_frame_download_async: ring of completion events */
  int dl_ev_next = 0;
  std::vector<Frame> frames;
  std::vector<Resident> resident;
  /* m355_submit_picture: rotating staging arenas, so the host prepares picture k+1 while k decodes; a slot is free again when the
     decode of the lists it held has finished.  THREE slots: a longer ring was measured and buys nothing — the submitting thread's own work per picture (list checks, schedules,
     ~20 launches: 0.45 ms at 8K) is what bounds a submit-every-picture decoder, and with more slots it runs further ahead of the
     three lanes, which costs more than it hides (C5 submit_only 0.74-0.79 ms with 3 slots, 0.80-0.94 with 4, 0.81-0.92 with 6:
     profiles/r04_ai_submit_ring.txt). */
  Resident transient[M355_TRANSIENT_MAX];
  int next_transient = 0;
  int transient_ring() const { return 3; }
  Frame work;                  /* pre-SAO working planes */
  /* scratch */
  uint32_t *pb_of = nullptr, *ticket = nullptr, *timeout = nullptr;
  unsigned long long* edge = nullptr;   /* k_intra halo granules */
  uint8_t *edge_tu = nullptr, *cuf = nullptr;   /* edge_tu also holds edge_pb and cb_cu (one allocation) */
  int16_t* resbuf = nullptr;
  uint32_t* jobs = nullptr;
  uint16_t* sao_nb = nullptr;
  uint16_t* iplan = nullptr;
  uint32_t* res_map = nullptr;
  uint32_t* job_base = nullptr;
  size_t cap_cb = 0, cap_u4 = 0, cap_edge = 0, cap_cuf = 0, cap_res = 0, cap_jobs = 0, cap_sao = 0, cap_iplan = 0, cap_resmap = 0, cap_jobbase = 0;
  uint32_t epoch = 0;
  /* per-decode status (m355_decode_status): the last M355_STATUS_RING decodes; a device-validated decode copies its lane's gate
     words into `words` (pinned) behind its last kernel */
  struct Status { unsigned long long serial = 0; uint32_t epoch = 0; bool validated = false, reported = false; EvRef done; };
  Status status[M355_STATUS_RING];
  uint32_t* status_words = nullptr;   /* pinned: 4 words per ring slot = the lane's timeout[0..3] at the end of the decode */
  unsigned long long serial = 0;
  /* rejected decodes that left the status ring unreported (more than M355_STATUS_RING submits between two waits): latched when
     their slot is reused, reported by the next m355_wait */
  unsigned long long lost_first = 0; int lost_count = 0;
  int stages = M355_STAGE_ALL;
  int shard_rank = 0, shard_n = 0;   /* shard_n == 0: sharding off */
  m355_comm comm = {nullptr, nullptr, nullptr};   /* exchanges of m355_decode_sharded */
  void* rccl = nullptr;              /* built-in RCCL communicator (m355_shard_rccl_init) */
  std::vector<hipEvent_t> evs;  /* 7 events per timed decode (ring grows on demand) */
  std::vector<uint8_t> ev_fused;   /* per timed decode: the residual stage ran first (launch_prediction) */
  int ev_used = 0;             /* decodes recorded since the last m355_timing_reset */
  bool timed = false;
  bool timing_on = false;      /* between m355_timing_reset and m355_timing_collect: decodes record their seven stage events */
  uint32_t* hash_acc = nullptr; /* m355_frame_hash accumulators */
  /* m355_decode_batch: ring of picture-record arrays (pinned staging + device copy + the batch's ticket word); a slot's event is
     recorded behind the batch's k_intra — what the pictures' filter stages wait for, and what guards the slot's reuse */
  struct BatchSlot { DevPic* host = nullptr; DevPic* dev = nullptr; uint32_t* ticket = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
  BatchSlot batch[M355_BATCH_RING];
  int batch_next = 0;
  hipStream_t batch_stream[4] = {nullptr, nullptr, nullptr, nullptr};   /* consecutive batches' k_intra launches alternate between two streams of priority
                                                         classes of their own (own hardware queues): the tail of one batch's wavefronts
                                                         overlaps the head of the next batch's when they run on different lanes */
  unsigned batch_count = 0;
  hipEvent_t batch_ev_pre[M355_MAX_LANES] = {};   /* the front part of picture k of the current batch is enqueued */
  /* CtbAddrRStoTS / TStoRS / TileIdRS of the last tile structure seen (pps.cc:589-606), upload() */
  struct ScanCache { int ctbW = 0, ctbH = 0, ntc = 0, ntr = 0; decltype(m355_pic_params::col_bd) col_bd; decltype(m355_pic_params::row_bd) row_bd;
                     std::vector<uint32_t> ctb_ts, ts2rs; std::vector<uint16_t> tile_id; } scan;
};

#define LANE_FIELDS(X) X(stream) X(stream2) X(stream_hi) X(last_stream) X(last) X(ev_fork) X(ev_fork2) X(ev_join) X(work) X(pb_of) X(edge) X(ticket) X(timeout) X(edge_tu) X(cuf) \
  X(resbuf) X(jobs) X(sao_nb) X(iplan) X(res_map) X(cap_resmap) X(job_base) X(cap_jobbase) X(cap_iplan) X(cap_cb) X(cap_u4) X(cap_edge) X(cap_cuf) X(cap_res) X(cap_jobs) X(cap_sao)
/* mark the point the stream has reached (one event packet); -> *out */
static int ev_mark(m355_ctx* c, hipStream_t st, EvRef* out)
{
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
static void ev_wait(m355_ctx* c, hipStream_t st, const EvRef& r)
{
  if (!r.ticket || r.stream == st) return;
  const m355_ctx::EvSlot& e = c->evring[r.ticket % M355_EV_RING];
  if (e.ticket == r.ticket) hipStreamWaitEvent(st, e.ev, 0);
}
/* the host waits for the mark / asks whether it has passed */
static hipError_t ev_sync(m355_ctx* c, const EvRef& r)
{
  if (!r.ticket) return hipSuccess;
  const m355_ctx::EvSlot& e = c->evring[r.ticket % M355_EV_RING];
  return e.ticket == r.ticket ? hipEventSynchronize(e.ev) : hipSuccess;
}
static hipError_t ev_query(m355_ctx* c, const EvRef& r)
{
  if (!r.ticket) return hipSuccess;
  const m355_ctx::EvSlot& e = c->evring[r.ticket % M355_EV_RING];
  return e.ticket == r.ticket ? hipEventQuery(e.ev) : hipSuccess;
}

static void select_lane(m355_ctx* c, int lane)
{
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
static int lane_class_priority(int index)
{
  static int lo = 0, hi = 0, probed = 0;
  if (!probed) { probed = 1; if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0; }   /* (least, greatest) */
  const int cls = (index / 3) % 3;
  return cls == 0 ? 0 : (cls == 1 ? hi : lo);
}
static int lane_priorities_mode() { return 2; }              /* 0 off, 1 every stream, 2 intra pictures only (what the measurements of round 3 left: profiles/r03_v_*) */
static int lane_priority(int index) { return lane_priorities_mode() == 1 ? lane_class_priority(index) : 0; }
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
  void* bufs[] = {l.pb_of, l.edge, l.ticket, l.timeout, l.edge_tu, l.cuf, l.resbuf, l.jobs, l.sao_nb, l.iplan, l.res_map, l.job_base};
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
static hipError_t sync_all(m355_ctx* c)
{
  hipError_t e = hipStreamSynchronize(c->stream);
  for (int k = 0; k < M355_MAX_LANES; k++)
    if (k != c->active && c->lanes[k].stream) { hipError_t e2 = hipStreamSynchronize(c->lanes[k].stream); if (e == hipSuccess) e = e2; }
  if (c->stream_hi) { hipError_t e2 = hipStreamSynchronize(c->stream_hi); if (e == hipSuccess) e = e2; }
  for (hipStream_t bs : c->batch_stream) if (bs) { hipError_t e2 = hipStreamSynchronize(bs); if (e == hipSuccess) e = e2; }
  for (int k = 0; k < M355_MAX_LANES; k++)
    if (k != c->active && c->lanes[k].stream_hi) { hipError_t e2 = hipStreamSynchronize(c->lanes[k].stream_hi); if (e == hipSuccess) e = e2; }
  return e;
}

template <class T> static int grow(T** p, size_t* cap, size_t need, hipStream_t st, bool zero)
{
  if (need <= *cap) return M355_OK;
  HIPCHK(hipStreamSynchronize(st));
  if (*p) hipFree(*p);
  *p = nullptr;
  const size_t n = need + need / 4 + 64;
  HIPCHK(hipMalloc(p, n * sizeof(T)));
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(T), st));
  *cap = n;
  return M355_OK;
}

struct TileRect { int x0, y0, x1, y1; };   /* luma samples */
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
static size_t slot_bytes(const m355_pic_params& pp, int nranks)
{
  size_t m = 0;
  for (int k = 0; k < nranks; k++) { const size_t b = tiles_bytes(pp, rank_tiles(pp, k, nranks)); if (b > m) m = b; }
  return m;
}
/* copy the tiles of ranks [k0, k1) except `skip` between the frame planes and their slots of the all-gather buffer (to_slot) or
   back: all rectangles in as few launches as the argument block allows */
static int copy_tiles(m355_ctx* c, const m355_pic_params& pp, Frame* f, int k0, int k1, int skip, int nranks, char* xbuf, size_t slot, bool to_slot)
{
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

/* ---- built-in RCCL transport (librccl is loaded on demand: the library itself does not link against it) ---- */
struct Id128 { char b[128]; };
struct Rccl {
  void* so = nullptr;
  void* comm = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, void*) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
};
static Rccl g_rccl;

extern "C" {

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

static void resident_free(Resident& r)
{
  for (void* b : r.xb) if (b) hipFree(b);
  if (r.xscratch) hipFree(r.xscratch);
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
static Frame* get_frame(m355_ctx* c, int h)
{
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
int m355_frame_upload(m355_ctx* c, int h, int cidx, const void* src, ptrdiff_t stride)
{
  Frame* f = get_frame(c, h);
  if (!f || cidx < 0 || cidx > 2 || !f->pw[cidx]) return fail(M355_ERR_INVALID, "bad frame/plane");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
#ifdef M355_X_TILED
  f->tiled_valid = false;
#endif
  HIPCHK(hipMemcpy2D(f->plane[cidx], (size_t)f->stride[cidx] * f->bpp[cidx], src, (size_t)stride * f->bpp[cidx],
                     (size_t)f->pw[cidx] * f->bpp[cidx], f->ph[cidx], hipMemcpyHostToDevice));
  return M355_OK;
}
int m355_frame_download(m355_ctx* c, int h, int cidx, void* dst, ptrdiff_t stride)
{
  Frame* f = get_frame(c, h);
  if (!f || cidx < 0 || cidx > 2 || !f->pw[cidx]) return fail(M355_ERR_INVALID, "bad frame/plane");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  HIPCHK(hipMemcpy2D(dst, (size_t)stride * f->bpp[cidx], f->plane[cidx], (size_t)f->stride[cidx] * f->bpp[cidx],
                     (size_t)f->pw[cidx] * f->bpp[cidx], f->ph[cidx], hipMemcpyDeviceToHost));
  return M355_OK;
}
static hipError_t frame_event(hipEvent_t* e);
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
#ifdef M355_X_TILED
  f->tiled_valid = false;
#endif
  for (int cc = 0; cc < 3; cc++) {
    if (!f->pw[cc]) continue;
    const size_t n = (size_t)f->stride[cc] * f->ph[cc];
    const int v = cc ? vc : vl;
    if (f->bpp[cc] == 1) { HIPCHK(hipMemsetAsync(f->plane[cc], v, n, c->stream)); HIPCHK(sync_all(c)); }
    else {
      std::vector<uint16_t> tmp(n, (uint16_t)v);
      HIPCHK(hipMemcpy(f->plane[cc], tmp.data(), n * 2, hipMemcpyHostToDevice));
    }
  }
  return M355_OK;
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
      HIPCHK(hipMemcpy2D(host[cc].data(), rb, f->plane[cc], (size_t)f->stride[cc] * f->bpp[cc], rb, f->ph[cc], hipMemcpyDeviceToHost));
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

/* ----------------------------------------------------------------------- validation ----------- */

extern "C++" {
/* Host-side parallel helpers: an 8K picture's lists are > 1.5 million records and 30 MB — validating and copying them on
 * one thread costs several milliseconds per picture, ten times the device time. */
static int host_threads()
{
  static int n = 0;
  if (!n) {
    const unsigned hc = std::thread::hardware_concurrency();
    n = hc >= 128 ? 32 : (hc >= 32 ? 16 : (hc >= 16 ? 8 : (hc >= 4 ? 4 : 1)));
    if (const char* e = getenv("M355_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = v; }   /* validation / staging threads per submit */
  }
  return n;
}
/* persistent workers: a submit runs several short parallel phases (validation, copies, schedules); creating threads for each
   of them costs more than the phases themselves */
struct HostPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  const std::function<void(int)>* job = nullptr;
  int n_parts = 0, next = 0, pending = 0;
  unsigned long long gen = 0;
  /* a submit runs its phases back to back (validation's checks, the intra schedule, the work list): a worker that has just finished
     a part polls this copy of `gen` for a few tens of microseconds before it blocks — the next phase then starts without a futex
     wake-up per worker (about what a short phase itself takes); an idle process still sleeps */
  std::atomic<unsigned long long> gen_hint{0};
  std::atomic<int> pending_hint{0};
  bool stop = false;
  explicit HostPool(int workers)
  {
    for (int i = 0; i < workers; i++) th.emplace_back([this]() { work(); });
  }
  ~HostPool()
  {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv_go.notify_all();
    for (auto& t : th) t.join();
  }
  static void cpu_relax()
  {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  void work()
  {
    unsigned long long seen = 0;
    bool warm = false;                                   /* finished a part a moment ago */
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (warm && !stop && !(gen != seen && next < n_parts)) {
        lk.unlock();
        for (int spin = 0; spin < 4000 && gen_hint.load(std::memory_order_acquire) == seen; spin++) cpu_relax();
        lk.lock();
      }
      warm = false;
      cv_go.wait(lk, [&]() { return stop || (gen != seen && next < n_parts); });
      if (stop) return;
      while (next < n_parts) {
        const int part = next++;
        const std::function<void(int)>* f = job;
        lk.unlock();
        (*f)(part);
        lk.lock();
        warm = true;
        pending_hint.store(--pending, std::memory_order_release);
        if (pending == 0) cv_done.notify_all();
      }
      seen = gen;
    }
  }
  void run(int parts, const std::function<void(int)>& f)      /* f(0 .. parts-1); the caller works too */
  {
    std::unique_lock<std::mutex> lk(mu);
    job = &f; n_parts = parts; next = 0; pending = parts; gen++;
    pending_hint.store(parts, std::memory_order_relaxed);
    gen_hint.store(gen, std::memory_order_release);
    cv_go.notify_all();
    while (next < n_parts) {
      const int part = next++;
      lk.unlock();
      f(part);
      lk.lock();
      pending_hint.store(--pending, std::memory_order_release);
    }
    if (pending) {                                       /* the last parts are about to finish on the workers: poll before sleeping */
      lk.unlock();
      for (int spin = 0; spin < 4000 && pending_hint.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
      lk.lock();
    }
    cv_done.wait(lk, [&]() { return pending == 0; });
    job = nullptr; n_parts = 0;
  }
};
static HostPool* g_pool = nullptr;
static HostPool& host_pool()
{
  /* lives until process exit (worker threads must not outlive it); a fork()ed child has no worker threads: it forgets the parent's
     pool (its threads do not exist there) and makes its own at the first parallel phase */
  static std::once_flag once;
  std::call_once(once, []() { pthread_atfork(nullptr, nullptr, []() { g_pool = nullptr; }); });
  if (!g_pool) g_pool = new HostPool(host_threads() - 1);
  return *g_pool;
}
static std::mutex g_pool_mu;                                 /* one parallel phase at a time (contexts on several threads share the pool) */
template <class F> static void parallel_ranges(size_t n, size_t min_per_thread, F f)   /* f(begin, end) */
{
  int T = host_threads();
  if (n < 2 * min_per_thread) T = 1;
  else if (n / min_per_thread < (size_t)T) T = (int)(n / min_per_thread);
  if (T <= 1) { f((size_t)0, n); return; }
  const std::function<void(int)> part = [&](int t) { f(n * (size_t)t / T, n * ((size_t)t + 1) / T); };
  /* one parallel phase at a time on the shared pool; a context that finds it busy (several decoders in one process, each on its own
     thread) does its phase itself instead of queueing behind the others */
  std::unique_lock<std::mutex> g(g_pool_mu, std::try_to_lock);
  if (!g.owns_lock()) { f((size_t)0, n); return; }
  host_pool().run(T, part);
}
/* check(i) -> nullptr or a message; the LOWEST failing index is reported as "<what> <i>: <message>" */
template <class F> static int check_all(const char* what, size_t n, F check)
{
  std::atomic<size_t> first(n);
  std::atomic<const char*> msg(nullptr);
  std::mutex mu;
  parallel_ranges(n, 32768, [&](size_t b, size_t e) {
    for (size_t i = b; i < e && i < first.load(std::memory_order_relaxed); i++) {
      const char* m = check(i);
      if (m) { std::lock_guard<std::mutex> g(mu); if (i < first.load()) { first.store(i); msg.store(m); } return; }
    }
  });
  if (first.load() < n) return fail(M355_ERR_INVALID, "%s %zu: %s", what, first.load(), msg.load());
  return M355_OK;
}
static void parallel_memcpy(void* dst, const void* src, size_t bytes)
{
  parallel_ranges(bytes, (size_t)1 << 20, [=](size_t b, size_t e) { memcpy((char*)dst + b, (const char*)src + b, e - b); });
}
} /* extern "C++" */

static int validate(const m355_picture* pic, const m355_rb* const* rb_bin_in, bool records_on_device, int* ctbW_out, int* ctbH_out)
{
  const m355_pic_params& pp = pic->pp;
  if (pp.width <= 0 || pp.height <= 0 || pp.chroma_format_idc > 3) return fail(M355_ERR_INVALID, "bad picture size / chroma format");
  if (pp.log2_ctb_size < 4 || pp.log2_ctb_size > 6 || pp.log2_min_tb_size < 2 || pp.log2_min_tb_size > pp.log2_ctb_size ||
      pp.log2_min_cb_size < 3 || pp.log2_min_cb_size > pp.log2_ctb_size)
    return fail(M355_ERR_INVALID, "bad block-size parameters");
  if (pp.bit_depth_luma < 8 || pp.bit_depth_luma > 16 || pp.bit_depth_chroma < 8 || pp.bit_depth_chroma > 16) return fail(M355_ERR_INVALID, "bad bit depth");
  /* pic_width/height_in_luma_samples are multiples of MinCbSizeY (>= 8) in every conforming SPS (sps.cc:428-437 rejects others);
     the filter kernels rely on it: they work in whole 4x4 units of every plane (k_deblock, k_sao) */
  if ((pp.width & ((1 << pp.log2_min_cb_size) - 1)) || (pp.height & ((1 << pp.log2_min_cb_size) - 1)))
    return fail(M355_ERR_INVALID, "picture size %dx%d is not a multiple of the minimum coding block size %d", pp.width, pp.height, 1 << pp.log2_min_cb_size);
  if (pp.width > 65535 - 64 || pp.height > 65535 - 64) return fail(M355_ERR_INVALID, "picture larger than the 16-bit block coordinates allow");
  if (pic->n_pbs >= (1 << 25)) return fail(M355_ERR_INVALID, "too many prediction blocks (job words hold 25 index bits)");
  const int cs = 1 << pp.log2_ctb_size;
  const int ctbW = (pp.width + cs - 1) / cs, ctbH = (pp.height + cs - 1) / cs;
  if (pic->n_ctbs != ctbW * ctbH) return fail(M355_ERR_INVALID, "n_ctbs %d != %dx%d", pic->n_ctbs, ctbW, ctbH);
  if (pp.num_tile_cols < 1 || pp.num_tile_cols > M355_MAX_TILE_COLS || pp.num_tile_rows < 1 || pp.num_tile_rows > M355_MAX_TILE_ROWS)
    return fail(M355_ERR_INVALID, "bad tile counts");
  if (pp.col_bd[0] != 0 || pp.row_bd[0] != 0 || pp.col_bd[pp.num_tile_cols] != ctbW || pp.row_bd[pp.num_tile_rows] != ctbH)
    return fail(M355_ERR_INVALID, "tile boundaries do not cover the picture");
  for (int i = 0; i < pp.num_tile_cols; i++) if (pp.col_bd[i + 1] <= pp.col_bd[i]) return fail(M355_ERR_INVALID, "tile columns not increasing");
  for (int i = 0; i < pp.num_tile_rows; i++) if (pp.row_bd[i + 1] <= pp.row_bd[i]) return fail(M355_ERR_INVALID, "tile rows not increasing");
  if (pic->n_slices < 1) return fail(M355_ERR_INVALID, "no slices");
  if ((pp.flags & M355_PF_SCALING_LIST) && !pic->scaling_factors) return fail(M355_ERR_INVALID, "scaling list enabled but no factors");
  if (pic->n_cus < 0 || pic->n_tus < 0 || pic->n_pbs < 0 || pic->n_wts < 0 || pic->n_ibs < 0) return fail(M355_ERR_INVALID, "negative list length");
  const int sw = (pp.chroma_format_idc == 1 || pp.chroma_format_idc == 2) ? 2 : 1, sh = pp.chroma_format_idc == 1 ? 2 : 1;
  int rc;
  uint32_t ibsum = 0;
  for (int i = 0; i < pic->n_ctbs; i++) {
    const m355_ctb& c = pic->ctbs[i];
    if (c.slice_idx >= pic->n_slices) return fail(M355_ERR_INVALID, "ctb %d: slice index out of range", i);
    if ((uint64_t)c.ib_start + c.ib_count > (uint64_t)pic->n_ibs) return fail(M355_ERR_INVALID, "ctb %d: intra block range out of bounds", i);
    ibsum += c.ib_count;
  }
  if ((int)ibsum != pic->n_ibs) return fail(M355_ERR_INVALID, "intra blocks not all owned by a CTB");
  size_t nrb = 0, bin_end[4];
  for (int s = 0; s < 4; s++) { if (pic->rb_count[s] < 0) return fail(M355_ERR_INVALID, "negative rb_count"); nrb += (size_t)pic->rb_count[s]; bin_end[s] = nrb; }
  /* every record of every list, as ONE parallel sweep over their concatenation (one thread start-up per picture) */
  auto chk_ctb = [&](size_t i) -> const char* {
    const m355_ctb& c = pic->ctbs[i];
    const int cx = (int)i % ctbW, cy = (int)i / ctbW;
    for (uint32_t k = 0; k < c.ib_count; k++) {
      const m355_ib& ib = pic->ibs[c.ib_start + k];
      if (ib.cidx > 2 || ib.log2_size < 2 || ib.log2_size > 5) return "an intra block is malformed";
      const int csw = ib.cidx ? (sw == 2) : 0, csh = ib.cidx ? (sh == 2) : 0, n = 1 << ib.log2_size;
      const int x0 = (cx << pp.log2_ctb_size) >> csw, y0 = (cy << pp.log2_ctb_size) >> csh;
      if (ib.x < x0 || ib.y < y0 || ib.x + n > x0 + (cs >> csw) || ib.y + n > y0 + (cs >> csh)) return "an intra block lies outside the CTB";
    }
    return nullptr;
  };
  auto chk_cu = [&](size_t i) -> const char* {
    const m355_cu& cu = pic->cus[i];
    return (cu.log2_size < pp.log2_min_cb_size || cu.log2_size > pp.log2_ctb_size || cu.x >= pp.width || cu.y >= pp.height || cu.pred_mode > 2 || cu.part_mode > 7) ? "malformed" : nullptr;
  };
  auto chk_tu = [&](size_t i) -> const char* {
    const m355_tu& tu = pic->tus[i];
    return (tu.log2_size < 2 || tu.log2_size > 6 || tu.x >= pp.width || tu.y >= pp.height) ? "malformed" : nullptr;
  };
  auto chk_pb = [&](size_t i) -> const char* {
    const m355_pb& pb = pic->pbs[i];
    if (pb.w < 4 || pb.h < 4 || pb.w > 64 || pb.h > 64 || (pb.w & 3) || (pb.h & 3) || pb.x + pb.w > pp.width || pb.y + pb.h > pp.height) return "geometry";
    if (!(pb.flags & (M355_PBF_MC_L0 | M355_PBF_MC_L1))) return "no list selected";
    for (int l = 0; l < 2 && !records_on_device; l++) {
      if (!(pb.flags & (M355_PBF_MC_L0 << l))) continue;
      if (!(pb.flags & (M355_PBF_FILL_L0 << l)) && (pb.ref_slot[l] < 0 || pb.ref_slot[l] >= M355_MAX_REF_FRAMES || pic->ref_frames[pb.ref_slot[l]] < 0)) return "reference slot invalid";
      if ((pb.flags & M355_PBF_WEIGHTED) && pb.wt_idx[l] >= pic->n_wts) return "weight index";
    }
    return nullptr;
  };
  auto chk_wt = [&](size_t i) -> const char* {
    return (pic->wts[i].log2wd_luma < 1 || pic->wts[i].log2wd_luma > 31 || (pp.chroma_format_idc && (pic->wts[i].log2wd_chroma < 1 || pic->wts[i].log2wd_chroma > 31))) ? "log2WD out of range" : nullptr;
  };
  /* the four size bins: consecutive in rbs[], or (lists recorded in place) in the regions m355_arena_begin handed out */
  const m355_rb* rb_bin[4];
  for (int b = 0; b < 4; b++) rb_bin[b] = rb_bin_in ? rb_bin_in[b] : pic->rbs + (b ? bin_end[b - 1] : 0);
  auto chk_rb = [&](size_t k) -> const char* {
    const int s = k < bin_end[0] ? 0 : (k < bin_end[1] ? 1 : (k < bin_end[2] ? 2 : 3));
    const m355_rb& rb = rb_bin[s][k - (s ? bin_end[s - 1] : 0)];
    const int n = 1 << (s + 2);
    const int W = rb.cidx ? pp.width / sw : pp.width, H = rb.cidx ? pp.height / sh : pp.height;
    if (rb.log2_size != s + 2 || rb.cidx > 2 || rb.kind > 3 || rb.x + n > W || rb.y + n > H) return "malformed";
    if ((uint64_t)rb.coeff_ofs + rb.ncoeff > pic->n_coeffs) return "coefficient range";
    if ((rb.flags & M355_RBF_DEFERRED) && (uint64_t)rb.res_ofs + n * n > pic->res_len) return "residual range";
    if ((pp.flags & M355_PF_SCALING_LIST) && (rb.matrix_id & 7) > 5) return "matrix id";
    if (rb.kind == M355_RK_DST && s != 0) return "DST only exists for 4x4";
    return nullptr;
  };
  auto chk_ib = [&](size_t i) -> const char* {
    const m355_ib& ib = pic->ibs[i];
    const int n = 1 << ib.log2_size;
    const int W = ib.cidx ? pp.width / sw : pp.width, H = ib.cidx ? pp.height / sh : pp.height;
    if (ib.log2_size < 2 || ib.log2_size > 5 || ib.cidx > 2 || ib.mode > 34 || ib.x + n > W || ib.y + n > H) return "malformed";
    if ((ib.flags & M355_IBF_HAS_RESIDUAL) && (uint64_t)ib.res_ofs + n * n > pic->res_len) return "residual range";
    if ((ib.flags & M355_IBF_PCM) && (uint64_t)ib.res_ofs + n * n > pic->n_pcm) return "pcm range";
    return nullptr;
  };
  const char* const names[7] = {"ctb", "cu", "tu", "pb", "weight", "rb", "ib"};
  /* records_on_device: only what the host's own schedules index by is checked here — the CTB table with each CTB's intra blocks;
     every record check runs in k_validate before any kernel acts on the lists, and the inter job counts are made on the device
     (k_job_count / k_job_scan): the host does not read the PB list at all */
  const size_t cnts[7] = {(size_t)pic->n_ctbs, records_on_device ? 0 : (size_t)pic->n_cus, records_on_device ? 0 : (size_t)pic->n_tus, records_on_device ? 0 : (size_t)pic->n_pbs,
                          records_on_device ? 0 : (size_t)pic->n_wts, records_on_device ? 0 : nrb, records_on_device ? 0 : (size_t)pic->n_ibs};
  size_t ofs[8];
  ofs[0] = 0;
  for (int q = 0; q < 7; q++) ofs[q + 1] = ofs[q] + cnts[q];
  std::atomic<size_t> first(ofs[7]);
  std::atomic<const char*> first_msg(nullptr);
  std::mutex mu;
  /* (records_on_device: the CTB table alone — 8 160 entries at 8K, each with a walk over its intra blocks: smaller shares, or the
     whole check runs on the calling thread) */
  parallel_ranges(ofs[7], records_on_device ? 512 : 8192, [&](size_t b, size_t e) {
    /* the range cut by list: one tight loop per list (the compiler sees ONE check function per loop) */
    for (int q = 0; q < 7; q++) {
      const size_t lo = std::max(b, ofs[q]), hi = std::min(e, ofs[q + 1]);
      if (lo >= hi || lo >= first.load(std::memory_order_relaxed)) continue;
      size_t bad = hi;
      const char* m = nullptr;
      const size_t base = ofs[q];
#define SWEEP(chk) for (size_t g = lo; g < hi; g++) if ((m = chk(g - base)) != nullptr) { bad = g; break; }
      switch (q) {
        case 0: SWEEP(chk_ctb) break;
        case 1: SWEEP(chk_cu) break;
        case 2: SWEEP(chk_tu) break;
        case 3: SWEEP(chk_pb) break;
        case 4: SWEEP(chk_wt) break;
        case 5: SWEEP(chk_rb) break;
        default: SWEEP(chk_ib) break;
      }
#undef SWEEP
      if (bad < hi) { std::lock_guard<std::mutex> gd(mu); if (bad < first.load()) { first.store(bad); first_msg.store(m); } return; }
    }
  });
  if (first.load() < ofs[7]) {
    int q = 0;
    while (first.load() >= ofs[q + 1]) q++;
    return fail(M355_ERR_INVALID, "%s %zu: %s", names[q], first.load() - ofs[q], first_msg.load());
  }
  (void)rc;
  *ctbW_out = ctbW; *ctbH_out = ctbH;
  return M355_OK;
}

/* ----------------------------------------------------------------------- upload --------------- */

/* Which neighbour CTBs must the intra wavefront wait for?  (k_intra.hip reads this mask.)
 * touch bits per CTB: an intra block reaches its right column (1), bottom row (2), both (4);
 * need bits: an intra block reads across the left (L), top (T), top-left (TL), top-right (TR) border. */
/* The dependency levels of the intra blocks come from the side their MODE reads (intra_schedule below) — fewer levels per CTB, i.e.
   fewer barrier steps of k_intra's chain (C2: 33.5 -> 19.3 levels per CTB, 1.638 -> 1.416 ms per picture on hardware,
   profiles/r05_a_switches_one_sided.txt).  M355_INTRA_LEVEL_STATS=1 prints the level count of every scheduled picture. */
static void intra_dependencies(int ctbW, int ctbH, const uint16_t* tile_id, const uint8_t* touch, const uint8_t* need, uint8_t* dep)
{
  const int nCtb = ctbW * ctbH;
  memset(dep, 0, (size_t)nCtb);
  /* serial and short: one pass over the CTBs (the "somebody reads ours" bit lands on a neighbour) */
  for (int c = 0; c < nCtb; c++) {
    if (!need[c]) continue;
    const int cx = c % ctbW, cy = c / ctbW;
    const int dx[4] = {-1, -1, 0, 1}, dy[4] = {0, -1, -1, -1};
    const uint8_t tbit[4] = {1, 4, 2, 2};                   /* what the neighbour must touch on its side */
    for (int n = 0; n < 4; n++) {
      const int nx = cx + dx[n], ny = cy + dy[n];
      if (nx < 0 || ny < 0 || nx >= ctbW) continue;
      const int nb = ny * ctbW + nx;
      if (tile_id[nb] != tile_id[c]) continue;             /* never read across tiles (intrapred.h:499-508) */
      if (((need[c] >> n) & 1) && (touch[nb] & tbit[n])) {
        dep[c] |= (uint8_t)(1 << n);
        dep[nb] |= 16;                                     /* somebody reads nb's output */
      }
    }
  }
}

/* Dependency levels of the intra blocks inside each CTB (k_intra.hip): a block reads the column left of it and the row
 * above it over 2*nT + 1 samples each, so it depends on every EARLIER block of its component and CTB that covers one
 * of those samples (a superset of what the availability rules of intrapred.h:534-633 let it read); level = 1 + the
 * highest level among them.  Blocks of one level are independent: k_intra runs them concurrently on several waves with
 * a workgroup barrier between levels, instead of walking the CTB's blocks one by one.  `out` receives each CTB's blocks
 * sorted by (level, component), decode order kept inside; `aux` per sorted block: its 4-word exec record (M355_IBX_*: geometry,
 * mode parameters, smoothing / publish flags, offset of its border plan inside the CTB's plans — k_intra_plan: 4nT + 1 entries per
 * predicted block, none for a raw block —, level); plan_count[ctb] =
 * the CTB's plan entries; log2_waves[ctb] = how wide the CTB's widest level is in luma blocks (0: 1, 1: 2, 2: 3-4,
 * 3: more) -> how many waves k_intra runs on it; *dense = intra picture (8 or more intra blocks per CTB of the
 * picture on average: an I picture of 64x64 CUs has 12, the inter pictures of the bench 3.4).
 * Returns the first CTB whose intra blocks overlap (they never do in a picture the reference decodes: one
 * decode_intra_prediction per transform block; the LDS budgets of k_intra rest on it), or -1. */
static int intra_schedule(const m355_picture* pic, int ctbW, int ctbH, m355_ib* out, uint32_t* aux, uint32_t* plan_count, uint8_t* log2_waves, uint8_t* touch, uint8_t* need, int* dense)
{
  const m355_pic_params& pp = pic->pp;
  const int sw = (pp.chroma_format_idc == 1 || pp.chroma_format_idc == 2) ? 2 : 1, sh = pp.chroma_format_idc == 1 ? 2 : 1;
  std::atomic<long long> n_blocks(0), n_intra_ctbs(0), n_levels(0);
  std::atomic<int> overlap(-1);
  const bool one_sided = !(pp.flags & M355_PF_CONSTRAINED_INTRA_PRED);
  /* (an intra picture has hundreds of blocks per CTB: smaller shares, so that a 1080p picture's 510 CTBs still use the whole pool) */
  parallel_ranges((size_t)pic->n_ctbs, (size_t)pic->n_ibs >= 8 * (size_t)pic->n_ctbs ? 16 : 256, [&](size_t cb, size_t ce) {
    std::vector<std::pair<uint32_t, uint32_t>> key, sorted;      /* (level << 2 | cidx, index) */
    uint32_t hist[4 * 128 + 1];                          /* stable counting sort of a CTB's keys (no allocation per CTB) */
    long long my_blocks = 0, my_ctbs = 0, my_levels = 0;
    for (size_t c = cb; c < ce; c++) {
      const m355_ctb& ctb = pic->ctbs[c];
      log2_waves[c] = 0; plan_count[c] = 0; touch[c] = 0; need[c] = 0;
      if (!ctb.ib_count) continue;
      my_ctbs++; my_blocks += ctb.ib_count;
      const int cx = (int)c % ctbW, cy = (int)c / ctbW;
      int8_t grid[3][16][16];                            /* level of the block covering each 4x4 unit (a chain in a CTB is < 64 long) */
      memset(grid, 0xFF, sizeof(grid));                  /* -1: no intra block of this CTB there (yet) */
      key.clear();
      bool clash = false;
      for (uint32_t k = 0; k < ctb.ib_count; k++) {
        const m355_ib& ib = pic->ibs[ctb.ib_start + k];
        const int csw = ib.cidx ? (sw == 2) : 0, csh = ib.cidx ? (sh == 2) : 0;
        const int ux = (ib.x - ((cx << pp.log2_ctb_size) >> csw)) >> 2, uy = (ib.y - ((cy << pp.log2_ctb_size) >> csh)) >> 2;
        const int n4 = (1 << ib.log2_size) >> 2;
        int level = 0;
        {
          /* what the CTB-to-CTB dependencies are made of (intra_dependencies): does a block reach the CTB's right column (1) /
             bottom row (2) / both (4), and does a predicted block read across the left (1), top-left (2), top (4), top-right (8) border */
          const int cw = (1 << pp.log2_ctb_size) >> csw, ch = (1 << pp.log2_ctb_size) >> csh, n = 1 << ib.log2_size;
          const int lx = ib.x - ((cx << pp.log2_ctb_size) >> csw), ly = ib.y - ((cy << pp.log2_ctb_size) >> csh);
          uint8_t t = 0, n_ = 0;
          if (lx + n == cw) t |= 1;
          if (ly + n == ch) t |= 2;
          if (lx + n == cw && ly + n == ch) t |= 4;
          if (!(ib.flags & M355_IBF_PCM)) {                /* raw blocks read no neighbours */
            if (lx == 0) n_ |= 1;
            if (lx == 0 && ly == 0) n_ |= 2;
            if (ly == 0) n_ |= 4;
            if (ly == 0 && lx + 2 * n > cw) n_ |= 8;
          }
          touch[c] |= t; need[c] |= n_;
        }
        if (ux < 0 || uy < 0 || ux >= 16 || uy >= 16) { key.push_back(std::make_pair((uint32_t)ib.cidx, k)); continue; }   /* rejected by validate() */
        if (!(ib.flags & M355_IBF_PCM)) {                /* raw blocks read nothing */
          /* M355_INTRA_ONE_SIDED: wait only for the blocks whose samples the MODE can read, instead of the whole 2nT + 1 border on
             both sides.  Per side, the border entries a mode uses (intrapred.h:261-433; entry 0 = corner, i > 0 the row above, i < 0
             the column on the left; + 1 entry where the [1 2 1] smoothing of intrapred.h:185-258 applies):
               planar +-(nT + 1) | DC +-nT | 11..25 (negative angle) +-nT | 10 / 26: nT on their own side, nT on the other one only
               with the boundary filter | 27..34: the row above up to nT + ((nT * angle) >> 5) + 2, nothing on the left | 2..9: the
               column on the left, nothing above.
             What makes that sound is where SUBSTITUTED entries get their value from (intrapred.h:637-665: the scan runs from the
             bottom-left entry up to the corner and on to the top-right one, an unavailable entry repeats the one before it):
             * above, a block with neighbours of its own CTB there (uy > 0: earlier in z-order, hence available) has entries 1 .. nT
               available, so an unavailable entry further right repeats one inside the used range; a block in the CTB's first row
               has no block of this CTB above it anyway;
             * on the left an unavailable entry repeats the one BELOW it, i.e. possibly one outside the used range: the range is
               cut to nT only where entries -1 .. -nT are all that is used and no smoothing reaches below them (they are available
               when ux > 0, and no block of this CTB is there when ux == 0) — else it stays 2nT;
             * dropping a side altogether needs the used side's first nT entries available (uy > 0 resp. ux > 0), or the scan
               would carry the other side's samples across the corner;
             * not with constrained intra prediction (an inter neighbour is unavailable: none of the above holds), and 32x32 luma
               blocks under strong smoothing read both ends of both sides for the bi-linear decision (intrapred.h:196-215).
             The corner unit always stays.  The entries outside the used range are still fetched by k_intra — possibly while their
             block is being written — and never used. */
          const int nT = 1 << ib.log2_size;
          int top_e = 2 * nT, left_e = 2 * nT;               /* used entries per side (0: the corner unit only) */
          if (one_sided && !(ib.cidx == 0 && ib.log2_size == 5 && (pp.flags & M355_PF_STRONG_INTRA_SMOOTHING))) {
            const int m = ib.mode;
            bool filt = false;                                 /* as e0's M355_IBX_FILT below */
            if (!(pp.flags & M355_PF_INTRA_SMOOTHING_DISABLED) && (ib.cidx == 0 || pp.chroma_format_idc == 3) && m != 1 && ib.log2_size != 2) {
              const int minDist = std::min(abs(m - 26), abs(m - 10));
              filt = ib.log2_size == 3 ? minDist > 7 : (ib.log2_size == 4 ? minDist > 1 : (ib.log2_size == 5 ? minDist > 0 : false));
            }
            const bool bf = ib.cidx == 0 && ib.log2_size < 5 && (m == 1 || !(ib.flags & M355_IBF_DISABLE_BOUNDARY_FILTER));
            static const int8_t mag[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
            int te = 2 * nT, le = 2 * nT;
            if (m == 0) { te = nT + 1; le = 2 * nT; }
            else if (m == 1) { te = nT; le = nT; }
            else if (m > 10 && m < 26) { te = nT; le = nT; }
            else if (m == 26) { te = nT; le = (bf || uy == 0) ? nT : 0; }
            else if (m == 10) { le = nT; te = (bf || ux == 0) ? nT : 0; }
            else if (m > 26) { te = std::min(2 * nT, nT + ((nT * mag[m - 26]) >> 5) + 2); le = uy > 0 ? 0 : 2 * nT; }
            else /* 2..9 */ { le = 2 * nT; te = ux > 0 ? 0 : 2 * nT; }
            if (filt) { if (te) te = std::min(2 * nT, te + 1); if (le) le = 2 * nT; }
            top_e = te; left_e = le;
          }
          const int top_u = (top_e + 3) >> 2, left_u = (left_e + 3) >> 2;   /* units beside the corner */
          for (int t = -1; t < 2 * n4; t++) {
            if (t < left_u && ux - 1 >= 0 && uy + t >= 0 && uy + t < 16) level = std::max(level, grid[ib.cidx][uy + t][ux - 1] + 1);
            if (t < top_u && uy - 1 >= 0 && ux + t >= 0 && ux + t < 16) level = std::max(level, grid[ib.cidx][uy - 1][ux + t] + 1);
          }
        }
        level = std::min(level, 126);                    /* (only overlapping blocks — rejected below — could get there) */
        for (int y = uy; y < uy + n4 && y < 16; y++)
          for (int x = ux; x < ux + n4 && x < 16; x++) { if (grid[ib.cidx][y][x] >= 0) clash = true; grid[ib.cidx][y][x] = (int8_t)level; }
        key.push_back(std::make_pair(((uint32_t)level << 2) | ib.cidx, k));
      }
      if (clash) { int e = -1; overlap.compare_exchange_strong(e, (int)c); }
      {
        uint32_t kmax = 0;
        for (const auto& e : key) kmax = std::max(kmax, e.first);
        for (uint32_t i = 0; i <= kmax + 1; i++) hist[i] = 0;
        for (const auto& e : key) hist[e.first + 1]++;
        for (uint32_t i = 1; i <= kmax; i++) hist[i] += hist[i - 1];
        sorted.resize(key.size());
        for (const auto& e : key) sorted[hist[e.first]++] = e;
        key.swap(sorted);
      }
      uint32_t widest = 1, run = 0, rel = 0;
      for (uint32_t k = 0; k < ctb.ib_count; k++) {
        const m355_ib& ib = pic->ibs[ctb.ib_start + key[k].second];
        out[ctb.ib_start + k] = ib;
        /* the block's EXEC RECORD for k_intra's chain (k_common.h M355_IBX_*): everything about the block that is not a sample
           value, precomputed here so that no instruction between two dependent blocks has to derive it */
        const int csw_ = ib.cidx ? (sw == 2) : 0, csh_ = ib.cidx ? (sh == 2) : 0;
        const int cwc = (1 << pp.log2_ctb_size) >> csw_, chc = (1 << pp.log2_ctb_size) >> csh_, nT_ = 1 << ib.log2_size;
        const int lx_ = ib.x - ((cx << pp.log2_ctb_size) >> csw_), ly_ = ib.y - ((cy << pp.log2_ctb_size) >> csh_);
        uint32_t e0 = (uint32_t)(lx_ & 127) | ((uint32_t)(ly_ & 127) << 7) | ((uint32_t)(ib.log2_size & 7) << 14) | ((uint32_t)(ib.cidx & 3) << 17) | ((uint32_t)(ib.mode & 63) << 19);
        if (ib.flags & M355_IBF_HAS_RESIDUAL) e0 |= M355_IBX_HAS_RES;
        if (ib.flags & M355_IBF_PCM) e0 |= M355_IBX_PCM;
        /* boundary smoothing of luma blocks < 32x32: DC always (intrapred.h:305), pure horizontal / vertical unless disabled (intrapred.h:378, 416, intrapred.cc:306-308) */
        if (ib.cidx == 0 && ib.log2_size < 5 && (ib.mode == 1 || !(ib.flags & M355_IBF_DISABLE_BOUNDARY_FILTER))) e0 |= M355_IBX_BFILT;
        if (lx_ + nT_ == cwc && cx + 1 < ctbW) e0 |= M355_IBX_PUB_COL;
        if (ly_ + nT_ == chc && cy + 1 < ctbH) e0 |= M355_IBX_PUB_ROW;
        /* which smoothing intra_prediction_sample_filtering (intrapred.h:185-258) will apply */
        if (!(ib.flags & M355_IBF_PCM) && !(pp.flags & M355_PF_INTRA_SMOOTHING_DISABLED) && (ib.cidx == 0 || pp.chroma_format_idc == 3) && ib.mode != 1 && ib.log2_size != 2) {
          const int minDist = std::min(abs((int)ib.mode - 26), abs((int)ib.mode - 10));
          const bool filt = ib.log2_size == 3 ? minDist > 7 : (ib.log2_size == 4 ? minDist > 1 : (ib.log2_size == 5 ? minDist > 0 : false));
          if (filt) e0 |= M355_IBX_FILT | (((pp.flags & M355_PF_STRONG_INTRA_SMOOTHING) && ib.cidx == 0 && ib.log2_size == 5) ? M355_IBX_STRONG : 0u);
        }
        /* intraPredAngle / invAngle of the mode (intrapred.h:313-326, intrapred.cc:268-274) */
        int angle = 0, inv = 0;
        if (ib.mode >= 2 && ib.mode <= 34) {
          static const int8_t mag[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
          static const int16_t invm[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
          const int d = ib.mode >= 18 ? abs((int)ib.mode - 26) : abs((int)ib.mode - 10);
          const bool neg = ib.mode >= 18 ? ib.mode < 26 : ib.mode > 10;
          angle = neg ? -mag[d] : mag[d];
          inv = angle < 0 ? -invm[d] : 0;
        }
        uint32_t* ex = aux + 4 * (size_t)(ctb.ib_start + k);
        const uint32_t cls = ib.mode == 0 ? 0u : (ib.mode == 1 ? 1u : (angle == 0 ? 2u : (angle > 0 ? 3u : 4u)));   /* planar, DC, pure H/V, angular +/- */
        ex[0] = e0; ex[1] = ib.res_ofs; ex[2] = ((uint32_t)(uint16_t)(int16_t)inv << 16) | (cls << 8) | (uint32_t)(uint8_t)(int8_t)angle;
        ex[3] = (rel & 0xFFFFu) | (((key[k].first >> 2) & 0x3FFFu) << 16);
        if (!(ib.flags & M355_IBF_PCM) && ib.log2_size >= 2 && ib.log2_size <= 5) rel += (4u << ib.log2_size) + 1u;
        run = (k && key[k].first == key[k - 1].first) ? run + 1 : 1;
        if ((key[k].first & 3) == 0) widest = std::max(widest, run);      /* luma blocks of one level */
      }
      plan_count[c] = rel;
      log2_waves[c] = widest >= 5 ? 3 : (widest >= 3 ? 2 : (widest == 2 ? 1 : 0));
      if (ctb.ib_count) my_levels += (key[ctb.ib_count - 1].first >> 2) + 1;
    }
    n_blocks += my_blocks; n_intra_ctbs += my_ctbs; n_levels += my_levels;
  });
  {
    static const bool stats = getenv("M355_INTRA_LEVEL_STATS") != nullptr;
    if (stats) fprintf(stderr, "intra_schedule: %lld blocks in %lld CTBs, %lld levels (one-sided %d)\n", n_blocks.load(), n_intra_ctbs.load(), n_levels.load(), (int)one_sided);
  }
  *dense = (n_intra_ctbs.load() && n_blocks.load() >= 8 * (long)ctbW * ctbH) ? 1 : 0;
  return overlap.load();
}

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
/* canonical exchange-buffer layout of a picture (k_common.h HaloLayout); depends on the picture parameters only */
static void halo_layout(const m355_pic_params& pp, HaloLayout& h)
{
  memset(&h, 0, sizeof(h));
  const int cf = pp.chroma_format_idc;
  const int sw = (cf == 1 || cf == 2) ? 2 : 1, sh = cf == 1 ? 2 : 1;
  const int cs = 1 << pp.log2_ctb_size;
  h.n_vb = pp.num_tile_cols - 1; h.n_hb = pp.num_tile_rows - 1;
  int col = 0, row = 0;
  for (int c = 0; c < 3; c++) {
    h.col_ofs[c] = col; h.row_ofs[c] = row;
    if (c && cf == 0) continue;
    const int pw = c ? pp.width / sw : pp.width, ph = c ? pp.height / sh : pp.height;
    h.hw[c] = c ? 4 / sw : 4; h.hh[c] = c ? 4 / sh : 4;
    for (int b = 0; b < h.n_vb; b++) h.xb[c][b] = (pp.col_bd[b + 1] * cs) / (c ? sw : 1);
    for (int b = 0; b < h.n_hb; b++) h.yb[c][b] = (pp.row_bd[b + 1] * cs) / (c ? sh : 1);
    col += h.n_vb * ph * 2 * h.hw[c];
    row += h.n_hb * 2 * h.hh[c] * pw;
  }
  h.col_ofs[3] = col; h.row_ofs[3] = row;
  const int w4 = (pp.width + 3) / 4, h4 = (pp.height + 3) / 4;
  h.n_units = 2 * h.n_vb * h4 + 2 * h.n_hb * w4;
}

struct Seg { const void* src; size_t bytes; size_t ofs; };
struct Lay {
  Seg seg[32];
  int ns;
  size_t total;
  int i_sl, i_ct, i_cu, i_tu, i_pb, i_wt, i_rb[4], i_ibin, i_ib, i_il, i_co, i_pc, i_sc, i_ts, i_rs, i_ti, i_iw, i_dp, i_ow;
};
static void caps_of(const m355_picture* pic, m355_arena_caps& k)
{
  memset(&k, 0, sizeof(k));
  k.n_slices = pic->n_slices; k.n_ctbs = pic->n_ctbs; k.n_cus = pic->n_cus; k.n_tus = pic->n_tus; k.n_pbs = pic->n_pbs; k.n_wts = pic->n_wts;
  for (int b = 0; b < 4; b++) k.n_rbs[b] = pic->rb_count[b];
  k.n_ibs = pic->n_ibs; k.n_coeffs = pic->n_coeffs; k.n_pcm = pic->n_pcm; k.scaling = pic->scaling_factors != nullptr;
}
/* where everything of one picture sits in the (pinned host / device) arena, for given list capacities */
static void make_layout(const m355_arena_caps& k, int nCtb, int halo_units, bool sharded, bool with_ib_input, Lay& L)
{
  L.ns = 0; L.total = 0;
  auto add = [&](size_t bytes) { L.seg[L.ns].src = nullptr; L.seg[L.ns].bytes = 0; L.seg[L.ns].ofs = L.total; L.total += al(bytes ? bytes : 1); return L.ns++; };
  L.i_sl = add(sizeof(m355_slice) * (size_t)k.n_slices);
  L.i_ct = add(sizeof(m355_ctb) * (size_t)k.n_ctbs);
  L.i_cu = add(sizeof(m355_cu) * ((size_t)k.n_cus + (size_t)halo_units));
  L.i_tu = add(sizeof(m355_tu) * (size_t)k.n_tus);
  L.i_pb = add(sizeof(m355_pb) * ((size_t)k.n_pbs + (size_t)halo_units));
  L.i_wt = add(sizeof(m355_wt) * (size_t)k.n_wts);
  for (int b = 0; b < 4; b++) L.i_rb[b] = add(sizeof(m355_rb) * (size_t)k.n_rbs[b]);
  L.i_ibin = add(with_ib_input ? sizeof(m355_ib) * (size_t)k.n_ibs : 0);   /* in place: the caller's blocks in decode order (host only) */
  L.i_ib = add(sizeof(m355_ib) * (size_t)k.n_ibs);      /* each CTB's blocks sorted by dependency level */
  L.i_il = add(16 * (size_t)k.n_ibs);                   /* ib_aux: one exec record (4 words) per block */
  L.i_co = add(4 * (size_t)k.n_coeffs);
  L.i_pc = add(2 * (size_t)k.n_pcm);
  L.i_sc = add(k.scaling ? 6 * (16 + 64 + 256 + 1024) : 0);
  L.i_ts = add(4 * (size_t)nCtb);   /* ctb_ts   */
  L.i_rs = add(4 * (size_t)nCtb);   /* ts2rs    */
  L.i_ti = add(2 * (size_t)nCtb);   /* tile_id  */
  L.i_iw = add(sizeof(DevIntraWork) * (size_t)nCtb);   /* intra_work */
  L.i_dp = add((size_t)nCtb);       /* ctb_dep */
  L.i_ow = add(sharded ? (size_t)nCtb : 0);                          /* ctb_owner */
}

static int upload(m355_ctx* c, Resident& r, const m355_picture* pic)
{
  static const bool prof = getenv("M355_PROFILE_UPLOAD") != nullptr;     /* phase times of this function on stderr */
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_start = now();
  const m355_pic_params& pp = pic->pp;
  const bool in_place = r.arena && r.host && pic->n_ctbs > 0 && (const char*)pic->ctbs >= r.host && (const char*)pic->ctbs < r.host + r.cap;
  int ctbW, ctbH;
  r.device_validate = in_place && c->shard_n < 1;   /* (a sharded picture's phases have no status slot: its lists are checked here) */
  int rc = validate(pic, in_place ? (const m355_rb* const*)r.caps.rb_bin : nullptr, r.device_validate, &ctbW, &ctbH);
  if (rc) return rc;
  const auto t_valid = now();
  const int nCtb = ctbW * ctbH;
  /* tile sharding: foreign border units are appended to cus[] / pbs[] by k_halo_unpack_meta */
  const bool sharded = c->shard_n >= 1;
  HaloLayout halo;
  memset(&halo, 0, sizeof(halo));
  if (sharded) halo_layout(pp, halo);
  /* in place: the lists were written into this arena through m355_arena_begin (same capacities -> same layout) */
  m355_arena_caps cp;
  caps_of(pic, cp);
  if (in_place) {
    const m355_arena_caps& k = r.caps;
    if (halo.n_units != r.arena_halo_units) return fail(M355_ERR_INVALID, "in-place submit: the arena was laid out for another tile structure (m355_picture_arena_begin's pp)");
    if (cp.n_slices > k.n_slices || cp.n_ctbs > k.n_ctbs || cp.n_cus > k.n_cus || cp.n_tus > k.n_tus || cp.n_pbs > k.n_pbs || cp.n_wts > k.n_wts ||
        cp.n_rbs[0] > k.n_rbs[0] || cp.n_rbs[1] > k.n_rbs[1] || cp.n_rbs[2] > k.n_rbs[2] || cp.n_rbs[3] > k.n_rbs[3] || cp.n_ibs > k.n_ibs ||
        cp.n_coeffs > k.n_coeffs || cp.n_pcm > k.n_pcm || (cp.scaling && !k.scaling))
      return fail(M355_ERR_INVALID, "lists exceed the capacities given to m355_arena_begin");
    cp = k;
  }
  Lay L;
  make_layout(cp, nCtb, halo.n_units, sharded, in_place, L);
  Seg* seg = L.seg;
  const int ns = L.ns;
  const size_t total = L.total;
  const int i_sl = L.i_sl, i_ct = L.i_ct, i_cu = L.i_cu, i_tu = L.i_tu, i_pb = L.i_pb, i_wt = L.i_wt, i_ib = L.i_ib, i_il = L.i_il, i_co = L.i_co, i_pc = L.i_pc,
            i_sc = L.i_sc, i_ts = L.i_ts, i_rs = L.i_rs, i_ti = L.i_ti, i_iw = L.i_iw, i_dp = L.i_dp, i_ow = L.i_ow;
  /* used bytes (what travels to the device) and, when copying, where they come from */
  {
    size_t rb_o = 0;
    const void* srcs[32]; size_t used[32];
    for (int i = 0; i < ns; i++) { srcs[i] = nullptr; used[i] = 0; }
    srcs[i_sl] = pic->slices; used[i_sl] = sizeof(m355_slice) * (size_t)pic->n_slices;
    srcs[i_ct] = pic->ctbs; used[i_ct] = sizeof(m355_ctb) * (size_t)pic->n_ctbs;
    srcs[i_cu] = pic->cus; used[i_cu] = sizeof(m355_cu) * (size_t)pic->n_cus;
    srcs[i_tu] = pic->tus; used[i_tu] = sizeof(m355_tu) * (size_t)pic->n_tus;
    srcs[i_pb] = pic->pbs; used[i_pb] = sizeof(m355_pb) * (size_t)pic->n_pbs;
    srcs[i_wt] = pic->wts; used[i_wt] = sizeof(m355_wt) * (size_t)pic->n_wts;
    for (int b = 0; b < 4; b++) { srcs[L.i_rb[b]] = pic->rbs + rb_o; used[L.i_rb[b]] = sizeof(m355_rb) * (size_t)pic->rb_count[b]; rb_o += (size_t)pic->rb_count[b]; }
    used[i_ib] = sizeof(m355_ib) * (size_t)pic->n_ibs; used[i_il] = 16 * (size_t)pic->n_ibs;
    srcs[i_co] = pic->coeffs; used[i_co] = 4 * (size_t)pic->n_coeffs;
    srcs[i_pc] = pic->pcm; used[i_pc] = 2 * (size_t)pic->n_pcm;
    srcs[i_sc] = pic->scaling_factors; used[i_sc] = pic->scaling_factors ? 6 * (16 + 64 + 256 + 1024) : 0;
    used[i_ts] = used[i_rs] = 4 * (size_t)nCtb; used[i_iw] = sizeof(DevIntraWork) * (size_t)nCtb;   /* (cut down to the items in use below) */ used[i_ti] = 2 * (size_t)nCtb; used[i_dp] = (size_t)nCtb;
    used[i_ow] = sharded ? (size_t)nCtb : 0;
    for (int i = 0; i < ns; i++) { seg[i].src = srcs[i]; seg[i].bytes = used[i]; }
    if (in_place) {
      /* every list must sit where the arena put it (the four size bins of rbs[] in their own regions: m355_arena_begin
         returns them through m355_arena_caps.rb_bin) */
      const void* want[] = {pic->slices, pic->ctbs, pic->cus, pic->tus, pic->pbs, pic->wts, pic->coeffs, pic->pcm};
      const int idx[] = {i_sl, i_ct, i_cu, i_tu, i_pb, i_wt, i_co, i_pc};
      for (int k = 0; k < 8; k++)
        if (seg[idx[k]].bytes && want[k] != (const void*)(r.host + seg[idx[k]].ofs)) return fail(M355_ERR_INVALID, "in-place submit: a list is not where m355_arena_begin put it");
      if (pic->n_ibs && pic->ibs != (const m355_ib*)(r.host + seg[L.i_ibin].ofs)) return fail(M355_ERR_INVALID, "in-place submit: ibs[] is not where m355_arena_begin put it");
      if (pic->rbs != (const m355_rb*)(r.host + seg[L.i_rb[0]].ofs)) return fail(M355_ERR_INVALID, "in-place submit: rbs must point at the first size bin's region");
      if (pic->scaling_factors && pic->scaling_factors != (const uint8_t*)(r.host + seg[i_sc].ofs)) return fail(M355_ERR_INVALID, "in-place submit: scaling_factors is not where m355_arena_begin put it");
      for (int i = 0; i < ns; i++) seg[i].src = nullptr;        /* nothing to copy */
    }
  }

  hipSetDevice(c->device);
  if (total > r.cap) {
    if (r.dev || r.host) HIPCHK(sync_all(c));
    if (r.dev) hipFree(r.dev);
    if (r.host) hipHostFree(r.host);
    r.dev = r.host = nullptr;
    r.cap = total + total / 4;
    HIPCHK(hipMalloc(&r.dev, r.cap));
    HIPCHK(hipHostMalloc(&r.host, r.cap, hipHostMallocDefault));
  } else if (r.done.ticket) {
    /* the arenas may still be in use by the last decode of these lists */
    HIPCHK(ev_sync(c, r.done));
    r.done = EvRef();
  }
  const auto t_wait = now();
  for (int i = 0; i < ns; i++)
    if (seg[i].src && seg[i].bytes) parallel_memcpy(r.host + seg[i].ofs, seg[i].src, seg[i].bytes);
  const auto t_copy = now();
  /* (per-CTB scratch of the schedules: kept in the Resident, no allocation per picture) */
  r.sched_u8.resize((size_t)nCtb * 3); r.sched_u32.resize((size_t)nCtb);
  uint8_t* const log2_waves = r.sched_u8.data(); uint8_t* const ctb_touch = log2_waves + nCtb; uint8_t* const ctb_need = ctb_touch + nCtb;
  uint32_t* const plan_count = r.sched_u32.data();
  int intra_dense = 0;
  {
    const int bad = intra_schedule(pic, ctbW, ctbH, (m355_ib*)(r.host + seg[i_ib].ofs), (uint32_t*)(r.host + seg[i_il].ofs), plan_count, log2_waves, ctb_touch, ctb_need, &intra_dense);
    if (bad >= 0) return fail(M355_ERR_INVALID, "ctb %d: intra blocks overlap", bad);
    const uint32_t cap = (uint32_t)M355_INTRA_PLAN_CAP(pp.chroma_format_idc);
    for (int i = 0; i < nCtb; i++) if (plan_count[(size_t)i] > cap) return fail(M355_ERR_INVALID, "ctb %d: more intra blocks than a CTB holds", i);
  }
  const auto t_sched = now();
  /* derived scan tables (pps.cc:589-606) */
  uint32_t* ctb_ts = (uint32_t*)(r.host + seg[i_ts].ofs);
  uint32_t* ts2rs = (uint32_t*)(r.host + seg[i_rs].ofs);
  uint16_t* tile_id = (uint16_t*)(r.host + seg[i_ti].ofs);
  DevIntraWork* iw = (DevIntraWork*)(r.host + seg[i_iw].ofs);
  {
    /* the tables depend on the tile structure only: kept from picture to picture (a stream changes it with its PPS) */
    m355_ctx::ScanCache& sc = c->scan;
    const bool same = sc.ctbW == ctbW && sc.ctbH == ctbH && sc.ntc == pp.num_tile_cols && sc.ntr == pp.num_tile_rows &&
                      !memcmp(sc.col_bd, pp.col_bd, sizeof(pp.col_bd)) && !memcmp(sc.row_bd, pp.row_bd, sizeof(pp.row_bd));
    if (!same) {
      sc.ctbW = ctbW; sc.ctbH = ctbH; sc.ntc = pp.num_tile_cols; sc.ntr = pp.num_tile_rows;
      memcpy(sc.col_bd, pp.col_bd, sizeof(pp.col_bd)); memcpy(sc.row_bd, pp.row_bd, sizeof(pp.row_bd));
      sc.ctb_ts.assign((size_t)nCtb, 0); sc.ts2rs.assign((size_t)nCtb, 0); sc.tile_id.assign((size_t)nCtb, 0);
      uint32_t ts = 0; int tidx = 0;
      for (int ty = 0; ty < pp.num_tile_rows; ty++)
        for (int tx = 0; tx < pp.num_tile_cols; tx++) {
          for (int y = pp.row_bd[ty]; y < pp.row_bd[ty + 1]; y++)
            for (int x = pp.col_bd[tx]; x < pp.col_bd[tx + 1]; x++) {
              sc.ctb_ts[y * ctbW + x] = ts; sc.ts2rs[ts] = (uint32_t)(y * ctbW + x); sc.tile_id[y * ctbW + x] = (uint16_t)tidx; ts++;
            }
          tidx++;
        }
    }
    memcpy(ctb_ts, sc.ctb_ts.data(), 4 * (size_t)nCtb); memcpy(ts2rs, sc.ts2rs.data(), 4 * (size_t)nCtb); memcpy(tile_id, sc.tile_id.data(), 2 * (size_t)nCtb);
  }
  /* intra work list (claimed in this order through k_intra's ticket): first the CTBs that wait for no neighbour,
     LONGEST FIRST (a CTB's blocks are a serial chain, so the CTB with the most blocks is the stage's critical path:
     it must start at once, not at a random point of the launch), then the dependent ones in decode order.  A
     workgroup still only ever waits on lower tickets: free CTBs never wait, dependent ones wait on free ones (all
     earlier) or on dependent ones earlier in decode order. */
  intra_dependencies(ctbW, ctbH, tile_id, ctb_touch, ctb_need, (uint8_t*)(r.host + seg[i_dp].ofs));
  for (int i = 0; i < nCtb; i++) ((uint8_t*)(r.host + seg[i_dp].ofs))[i] |= (uint8_t)(log2_waves[i] << 5);
  const auto t_deps = now();
  int nw = 0, n_free = 0;
  uint32_t n_iplan = 0;                                     /* border-plan entries of the picture (k_intra_plan) */
  {
    const uint8_t* dep = (const uint8_t*)(r.host + seg[i_dp].ofs);
    /* the order first — two stable COUNTING sorts over the CTBs in decode order (keys are small: blocks per CTB, x + 2y inside a
       tile), a comparison sort of the 8K picture's 2800 intra CTBs cost more than everything else here —, then the items, in parallel */
    std::vector<uint32_t>& order = r.sched_order;          /* raster address of work item k */
    std::vector<uint32_t>& cand = r.sched_cand;            /* the CTBs with intra blocks in decode order: raster address, blocks | free << 31 */
    cand.clear();
    uint32_t max_cnt = 0;
    size_t n_dep = 0;
    for (int t = 0; t < nCtb; t++) {
      const uint32_t rs = ts2rs[t], cnt = pic->ctbs[rs].ib_count;
      if (!cnt) continue;
      const bool free_ctb = !(dep[rs] & 15);
      cand.push_back(rs); cand.push_back(cnt | (free_ctb ? 0x80000000u : 0u));
      if (free_ctb) max_cnt = std::max(max_cnt, cnt); else n_dep++;
    }
    const size_t n_cand = cand.size() / 2;
    n_free = (int)(n_cand - n_dep);
    order.resize(n_cand);
    std::vector<uint32_t>& bucket = r.sched_bucket;
    {
      /* free CTBs, LONGEST first (bucket = max - count), decode order inside a bucket */
      bucket.assign((size_t)max_cnt + 2, 0);
      for (size_t i = 0; i < n_cand; i++) if (cand[2 * i + 1] >> 31) bucket[(size_t)(max_cnt - (cand[2 * i + 1] & 0x7FFFFFFFu)) + 1]++;
      for (size_t i = 1; i < bucket.size(); i++) bucket[i] += bucket[i - 1];
      for (size_t i = 0; i < n_cand; i++) if (cand[2 * i + 1] >> 31) order[bucket[max_cnt - (cand[2 * i + 1] & 0x7FFFFFFFu)]++] = cand[2 * i];
    }
    if (n_dep) {
      /* the dependent CTBs in WAVEFRONT order of their tile (x + 2y, the time at which the CTB's neighbours L / TL / T / TR — all
         of smaller x + 2y — can have delivered): workgroups are dispatched in this order, so with more CTBs than the GPU holds
         at once (large pictures, several pictures in flight) the resident ones are those that can run, not the rest of a CTB row
         whose turn comes much later; any order in which a CTB follows its four neighbours keeps the ticket protocol deadlock-free */
      std::vector<int> tx0((size_t)pp.num_tile_cols * pp.num_tile_rows), ty0(tx0.size());
      for (int ty = 0, t = 0; ty < pp.num_tile_rows; ty++)
        for (int tx = 0; tx < pp.num_tile_cols; tx++, t++) { tx0[t] = pp.col_bd[tx]; ty0[t] = pp.row_bd[ty]; }
      auto wkey = [&](uint32_t rs) { const int cx = (int)rs % ctbW, cy = (int)rs / ctbW, ti = tile_id[rs]; return (uint32_t)((cx - tx0[ti]) + 2 * (cy - ty0[ti])); };
      bucket.assign((size_t)ctbW + 2 * (size_t)ctbH + 2, 0);
      for (size_t i = 0; i < n_cand; i++) if (!(cand[2 * i + 1] >> 31)) bucket[(size_t)wkey(cand[2 * i]) + 1]++;
      for (size_t i = 1; i < bucket.size(); i++) bucket[i] += bucket[i - 1];
      for (size_t i = 0; i < n_cand; i++) if (!(cand[2 * i + 1] >> 31)) order[(size_t)n_free + bucket[wkey(cand[2 * i])]++] = cand[2 * i];
    }
    nw = (int)order.size();
    /* where each item's border plans start (a running sum in work order) */
    std::vector<uint32_t>& pbase = r.sched_u32b;
    pbase.resize((size_t)nw + 1);
    for (int k = 0; k < nw; k++) { pbase[(size_t)k] = n_iplan; n_iplan += (plan_count[order[(size_t)k]] + 7u) & ~7u; }
    /* a work item = the CTB's descriptor: block range, wave count code, and the 3x3 neighbourhood facts every availability
       test of intrapred.h:486-508 / :534-633 needs (picture, slice, tile, decode order across CTBs) */
    parallel_ranges((size_t)nw, 512, [&](size_t kb, size_t ke) {
      for (size_t k = kb; k < ke; k++) {
        const uint32_t rs = order[k];
        DevIntraWork w;
        memset(&w, 0, sizeof(w));
        w.ctb = rs; w.ib_start = pic->ctbs[rs].ib_start; w.ib_count = pic->ctbs[rs].ib_count;
        w.waves_code = (uint8_t)(log2_waves[rs] & 3);
        w.plan_base = pbase[k]; w.plan_count = plan_count[rs];
        const int cx = (int)rs % ctbW, cy = (int)rs / ctbW;
        const uint32_t my_sa = pic->slices[pic->ctbs[rs].slice_idx].slice_addr_rs;
        for (int q = 0; q < 9; q++) {
          const int nx = cx + q % 3 - 1, ny = cy + q / 3 - 1;
          if (nx < 0 || ny < 0 || nx >= ctbW || ny >= ctbH) continue;
          const int n = ny * ctbW + nx;
          if (pic->slices[pic->ctbs[n].slice_idx].slice_addr_rs == my_sa && tile_id[n] == tile_id[rs]) w.nb_same |= (uint16_t)(1u << q);
          if (ctb_ts[n] < ctb_ts[rs]) w.nb_earlier |= (uint16_t)(1u << q);
        }
        iw[k] = w;
      }
    });
  }
  seg[i_iw].bytes = sizeof(DevIntraWork) * (size_t)(nw ? nw : 1);
  r.n_intra_work = nw; r.n_iplan = n_iplan;
  if (sharded) {
    uint8_t* ow = (uint8_t*)(r.host + seg[i_ow].ofs);
    const int n_tiles = pp.num_tile_cols * pp.num_tile_rows;
    for (int i = 0; i < nCtb; i++) ow[i] = m355_shard_owner_of_tile(tile_id[i], n_tiles, c->shard_n) == c->shard_rank;
    /* a sharded picture must hold only this rank's blocks (the lists drive the reconstruction kernels) */
    for (int i = 0; i < pic->n_cus; i++)
      if (!ow[(pic->cus[i].y >> pp.log2_ctb_size) * ctbW + (pic->cus[i].x >> pp.log2_ctb_size)])
        return fail(M355_ERR_INVALID, "sharded picture: cu %d lies in a tile of another rank", i);
    for (int i = 0; i < nCtb; i++)
      if (!ow[i] && pic->ctbs[i].ib_count) return fail(M355_ERR_INVALID, "sharded picture: ctb %d of another rank has intra blocks", i);
  }
  r.sharded = sharded; r.shard_rank = c->shard_rank; r.shard_n = c->shard_n; r.halo = halo; r.live_valid = false; r.xprev = nullptr;
  r.bytes = total; r.fresh = true; r.refs_valid = false;
  if (prof) fprintf(stderr, "m355 upload%s: validate %.3f ms, wait/alloc %.3f, copy %.3f (%.1f MB), intra schedule %.3f, tables + dependencies %.3f, work list + jobs %.3f\n",
                    in_place ? " (in place)" : "", ms(t_start, t_valid), ms(t_valid, t_wait), ms(t_wait, t_copy), total / 1e6, ms(t_copy, t_sched), ms(t_sched, t_deps), ms(t_deps, now()));
  {
    /* host -> device: what is used of every segment (capacities handed out by m355_arena_begin may be far larger); adjacent
       segments travel together */
    size_t run_b = 0, run_e = 0;
    for (int i = 0; i <= ns; i++) {
      const bool used = i < ns && seg[i].bytes && i != L.i_ibin;
      const size_t b = used ? seg[i].ofs : 0, e = used ? seg[i].ofs + seg[i].bytes : 0;
      if (used && run_e > run_b && b - run_e <= 4096) { run_e = e; continue; }      /* small gap: one copy */
      if (run_e > run_b) HIPCHK(hipMemcpyAsync(r.dev + run_b, r.host + run_b, run_e - run_b, hipMemcpyHostToDevice, c->stream));
      run_b = b; run_e = e;
    }
  }
  {
    const int rcm = ev_mark(c, c->stream, &r.up);  /* a decode on another lane continues behind the copy of the lists */
    if (rcm) return rcm;
  }

  r.hdr = *pic;
  DevPic& d = r.dp;
  memset(&d, 0, sizeof(d));
  d.pp = dev_pic_params(pp);
  d.sw = (pp.chroma_format_idc == 1 || pp.chroma_format_idc == 2) ? 2 : 1;
  d.sh = pp.chroma_format_idc == 1 ? 2 : 1;
  d.ctbW = ctbW; d.ctbH = ctbH; d.nCtb = nCtb;
  d.w4 = (pp.width + 3) / 4; d.h4 = (pp.height + 3) / 4;
  d.wcb = (pp.width + (1 << pp.log2_min_cb_size) - 1) >> pp.log2_min_cb_size;
  d.hcb = (pp.height + (1 << pp.log2_min_cb_size) - 1) >> pp.log2_min_cb_size;
  d.slices = (const m355_slice*)(r.dev + seg[i_sl].ofs);
  d.ctbs = (const m355_ctb*)(r.dev + seg[i_ct].ofs);
  d.cus = (const m355_cu*)(r.dev + seg[i_cu].ofs);
  d.tus = (const m355_tu*)(r.dev + seg[i_tu].ofs);
  d.pbs = (const m355_pb*)(r.dev + seg[i_pb].ofs);
  d.wts = (const m355_wt*)(r.dev + seg[i_wt].ofs);
  for (int b = 0; b < 4; b++) d.rb_bin[b] = (const m355_rb*)(r.dev + seg[L.i_rb[b]].ofs);
  d.ibs = (const m355_ib*)(r.dev + seg[i_ib].ofs);
  d.ib_aux = (const uint32_t*)(r.dev + seg[i_il].ofs);
  d.intra_dense = intra_dense;
  d.coeffs = (const uint32_t*)(r.dev + seg[i_co].ofs);
  d.pcm = (const uint16_t*)(r.dev + seg[i_pc].ofs);
  d.scaling = pic->scaling_factors ? (const uint8_t*)(r.dev + seg[i_sc].ofs) : nullptr;
  d.n_cus = pic->n_cus; d.n_tus = pic->n_tus; d.n_pbs = pic->n_pbs; d.n_ibs = pic->n_ibs;
  for (int s = 0; s < 4; s++) d.rb_count[s] = pic->rb_count[s];
  d.ctb_ts = (const uint32_t*)(r.dev + seg[i_ts].ofs);
  d.ts2rs = (const uint32_t*)(r.dev + seg[i_rs].ofs);
  d.tile_id = (const uint16_t*)(r.dev + seg[i_ti].ofs);
  d.intra_work = (const DevIntraWork*)(r.dev + seg[i_iw].ofs);
  d.n_intra_work = nw; d.n_intra_free = n_free;
  d.ctb_dep = (const uint8_t*)(r.dev + seg[i_dp].ofs);
  d.ctb_owner = sharded ? (const uint8_t*)(r.dev + seg[i_ow].ofs) : nullptr;
  d.halo_cu_base = pic->n_cus; d.halo_pb_base = pic->n_pbs;
  d.n_pb_records = pic->n_pbs + halo.n_units;
  d.device_validate = r.device_validate ? 1 : 0;
  d.n_wts = pic->n_wts; d.n_coeffs = pic->n_coeffs; d.n_pcm = pic->n_pcm; d.res_len = pic->res_len;
  r.used = true;
  return M355_OK;
}

/* ----------------------------------------------------------------------- decode --------------- */

/* frames, scratch and the device descriptor of one decode of `r` */
static int prepare(m355_ctx* c, Resident& r, DevPic& d_out, bool& want_sao_out)
{
  hipSetDevice(c->device);
  const m355_picture& pic = r.hdr;
  const m355_pic_params& pp = pic.pp;
  Frame* dst = get_frame(c, pic.dst_frame);
  if (!dst) return fail(M355_ERR_INVALID, "dst_frame %d is not a live frame", pic.dst_frame);
  if (dst->w != pp.width || dst->h != pp.height || dst->cf != pp.chroma_format_idc || dst->bdl != pp.bit_depth_luma || dst->bdc != pp.bit_depth_chroma)
    return fail(M355_ERR_INVALID, "dst frame geometry does not match the picture parameters");
  DevPic d = r.dp;
  d.ref_valid = 0;
  DevRef refs[M355_MAX_REF_FRAMES];
  memset(refs, 0, sizeof(refs));
  for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
    if (pic.ref_frames[i] < 0) continue;
    Frame* f = get_frame(c, pic.ref_frames[i]);
    if (!f) return fail(M355_ERR_INVALID, "ref_frames[%d]=%d is not a live frame", i, pic.ref_frames[i]);
    if (f->w != dst->w || f->h != dst->h || f->cf != dst->cf || f->bdl != dst->bdl || f->bdc != dst->bdc)
      return fail(M355_ERR_INVALID, "reference frame %d geometry differs (motion.cc:377-398 would conceal; record FILL instead)", i);
    if (f == dst) return fail(M355_ERR_INVALID, "a picture cannot reference itself");
    for (int cc = 0; cc < 3; cc++) { refs[i].plane[cc] = f->plane[cc]; refs[i].stride[cc] = f->stride[cc]; }
#ifdef M355_X_TILED
    for (int cc = 0; cc < 3; cc++) {
      if (!f->pw[cc]) continue;
      const int row_len = cc ? M355_TILE_ROW_C : M355_TILE_ROW_L;
      if (!f->tiled[cc]) {
        f->tiles_w[cc] = (f->pw[cc] + M355_TILE_W - 1) / M355_TILE_W;
        const size_t bytes = (size_t)((f->ph[cc] + M355_TILE_H - 1) / M355_TILE_H) * f->tiles_w[cc] * M355_TILE_H * row_len * f->bpp[cc] + 256;
        HIPCHK(hipMalloc(&f->tiled[cc], bytes));
        f->tiled_valid = false;
      }
      refs[i].tiled[cc] = f->tiled[cc]; refs[i].trs[cc] = f->tiles_w[cc] * M355_TILE_H * row_len;
    }
#endif
    refs[i].valid = 1;
    d.ref_valid |= 1u << i;
  }
  if (!r.refs_dev) {
    HIPCHK(hipMalloc(&r.refs_dev, sizeof(refs)));
    HIPCHK(hipHostMalloc(&r.refs_host, sizeof(refs), hipHostMallocDefault));
    r.refs_valid = false;
  }
  if (!r.refs_valid || memcmp(r.refs_host, refs, sizeof(refs)) != 0) {
    if (!r.fresh) HIPCHK(sync_all(c));       /* a decode in flight may still read the table / the staging copy */
    memcpy(r.refs_host, refs, sizeof(refs));
    HIPCHK(hipMemcpyAsync(r.refs_dev, r.refs_host, sizeof(refs), hipMemcpyHostToDevice, c->stream));
    r.refs_valid = true;
  }
  d.refs = r.refs_dev;
  {
    /* k_inter_jobs' tap tables: one small constant buffer per (plane type, bit depths) this context has decoded */
    const uint32_t key = (uint32_t)(dst->bpp[0] == 1) | ((uint32_t)pp.bit_depth_luma << 8) | ((uint32_t)pp.bit_depth_chroma << 16);
    const uint32_t* tab = nullptr;
    for (auto& e : c->inter_tabs) if (e.first == key) tab = e.second;
    if (!tab) {
      uint32_t host[M355_INTER_TAB_WORDS];
      m355_inter_tables(dst->bpp[0] == 1, std::min((int)pp.bit_depth_luma, 16), std::min((int)pp.bit_depth_chroma, 16), host);
      uint32_t* dev = nullptr;
      HIPCHK(hipMalloc(&dev, sizeof(host)));
      HIPCHK(hipMemcpy(dev, host, sizeof(host), hipMemcpyHostToDevice));
      c->inter_tabs.emplace_back(key, dev);
      tab = dev;
    }
    d.inter_tabs = tab;
  }
  /* scratch */
  int rc;
  {
    /* edge_tu | edge_pb | cb_cu in one allocation (one memset per picture, k_meta.hip); pb_of separate */
    const size_t u4 = (size_t)d.w4 * d.h4, ncb = (size_t)d.wcb * d.hcb;
    const size_t need = ((2 * u4 + 63) & ~(size_t)63) + ncb * 4 + 64;
    if ((rc = grow(&c->edge_tu, &c->cap_u4, need, c->stream, false))) return rc;
    if ((rc = grow(&c->pb_of, &c->cap_cb, u4, c->stream, true))) return rc;
  }
  {
    /* k_intra's halo granules: per component ctbW right columns of ph / 2 granules and ctbH bottom rows of pw / 2; zero at
       allocation, never cleared: a granule is valid when it carries the epoch of the decode that reads it */
    size_t n = 0;
    for (int cc = 0; cc < 3; cc++) {
      d.edge_col_ofs[cc] = (uint32_t)n; n += (size_t)d.ctbW * (size_t)(dst->ph[cc] >> 1);
      d.edge_row_ofs[cc] = (uint32_t)n; n += (size_t)d.ctbH * (size_t)(dst->pw[cc] >> 1);
    }
    if ((rc = grow(&c->edge, &c->cap_edge, n + 1, c->stream, true))) return rc;
  }
  if ((rc = grow(&c->cuf, &c->cap_cuf, (size_t)pic.n_cus + (size_t)r.halo.n_units + 1, c->stream, false))) return rc;
  /* Fused inter residuals (k_common.h res_map): whenever k_inter_jobs runs, k_residual hands the blocks of inter CUs over as int16
     tiles behind the deferred (intra) ones instead of read-modify-writing the picture.  Not for 16-bit samples (a residual of
     transform_idct_add, fallback-dct.cc:550-691, needs 18 bits there), not for the generic kernel's chroma formats, not when a
     stage is isolated. */
  /* When: with ONE picture in flight (the residual stage then runs beside the job list instead of behind k_inter_jobs: 0.505 vs
     0.52 ms per C5 picture).  With pictures in flight the read-modify-write order is the faster one although it moves 80 MB more
     per picture: k_inter_jobs is the stage everything else queues behind, the 20 us the residual rows add to it cost more than the
     35 us k_residual saves beside the other pictures' kernels (0.379 vs 0.395 ms, profiles/r04_g_*).  M355_RES_FUSED=0 / 1 forces it. */
  static const int fused_env = getenv("M355_RES_FUSED") ? atoi(getenv("M355_RES_FUSED")) : -1;
  const bool fused_on = fused_env >= 0 ? fused_env != 0 : c->depth == 1;
  const bool fused = fused_on && pic.n_pbs > 0 && pp.chroma_format_idc <= 1 && pp.bit_depth_luma < 16 && pp.bit_depth_chroma < 16 &&
                     (c->stages & M355_STAGE_INTER) && (c->stages & M355_STAGE_RESIDUAL) &&
                     (pic.rb_count[0] | pic.rb_count[1] | pic.rb_count[2] | pic.rb_count[3]);
  size_t res_need = (size_t)pic.res_len + 1;
  d.res_map = nullptr;
  if (fused) {
    size_t base = ((size_t)pic.res_len + 15) & ~(size_t)15;
    for (int s = 0; s < 4; s++) { d.res_fused_base[s] = (uint32_t)base; base += (size_t)pic.rb_count[s] << (2 * (s + 2)); }
    if (base >= ((size_t)1 << 30)) return fail(M355_ERR_INVALID, "residual blocks exceed the fused residual buffer");
    res_need = base + 1;
    size_t n = 0;
    for (int cc = 0; cc < (pp.chroma_format_idc ? 3 : 1); cc++) {
      d.res_map_ofs[cc] = (uint32_t)n; d.res_map_w[cc] = (dst->pw[cc] + 3) >> 2;
      n += (size_t)d.res_map_w[cc] * ((dst->ph[cc] + 3) >> 2);
    }
    if ((rc = grow(&c->res_map, &c->cap_resmap, n + 1, c->stream, false))) return rc;
    d.res_map = c->res_map;
    d.res_map_words = (uint32_t)n;
  }
  if ((rc = grow(&c->resbuf, &c->cap_res, res_need, c->stream, false))) return rc;
  if ((rc = grow(&c->sao_nb, &c->cap_sao, (size_t)d.nCtb * 3, c->stream, false))) return rc;
  {
    /* inter jobs of 4 x 8 luma samples: a list of disjoint prediction blocks makes at most one per 16 luma samples (8x4 blocks),
       and at most one per 32 plus eight per block; the counts themselves are made on the device (k_job_count / k_job_scan) */
    const size_t area = (size_t)pp.width * pp.height;
    const size_t cap = pic.n_pbs > 0 ? std::min(area / 16, area / 32 + 8 * (size_t)pic.n_pbs) + 256 : 1;
    const size_t n_chunks = ((size_t)(pic.n_pbs > 0 ? pic.n_pbs : 0) + 255) / 256;
    if ((rc = grow(&c->jobs, &c->cap_jobs, cap, c->stream, false))) return rc;
    if ((rc = grow(&c->job_base, &c->cap_jobbase, n_chunks * 4 + 8, c->stream, true))) return rc;
    d.jobs_cap = (uint32_t)cap; d.job_base = c->job_base; d.job_tot = c->job_base + n_chunks * 4;
  }
  if ((rc = grow(&c->iplan, &c->cap_iplan, (size_t)r.n_iplan + 8, c->stream, false))) return rc;

  const bool want_sao = (c->stages & M355_STAGE_SAO) && (pp.flags & M355_PF_SAO_ENABLED);
  Frame* target = dst;
  if (want_sao) {
    if (!c->work.used || c->work.w != dst->w || c->work.h != dst->h || c->work.cf != dst->cf || c->work.bdl != dst->bdl || c->work.bdc != dst->bdc) {
      HIPCHK(sync_all(c));
      if (c->work.used) frame_free(c->work);
      frame_geometry(c->work, dst->w, dst->h, dst->cf, dst->bdl, dst->bdc);
      if ((rc = frame_alloc(c->work, c->stream))) return rc;
    }
    target = &c->work;
  }
  for (int cc = 0; cc < 3; cc++) {
    d.pw[cc] = dst->pw[cc]; d.ph[cc] = dst->ph[cc];
    d.plane[cc] = target->plane[cc]; d.stride[cc] = target->stride[cc];
    d.out_plane[cc] = dst->plane[cc]; d.out_stride[cc] = dst->stride[cc];
  }
  d.edge_tu = c->edge_tu; d.edge_pb = c->edge_tu + (size_t)d.w4 * d.h4;
  d.cb_cu = (uint32_t*)(c->edge_tu + (((size_t)2 * d.w4 * d.h4 + 63) & ~(size_t)63));
  d.cuf = c->cuf; d.pb_of = c->pb_of;
  d.fill_pb_of_in_meta = ((c->stages & M355_STAGE_INTER) && pp.chroma_format_idc <= 1) ? 0 : 1;   /* else k_inter_jobs writes it */
  d.jobs = c->jobs; d.sao_nb = c->sao_nb; d.iplan = c->iplan;
#ifdef M355_X_PROF
  {
    static unsigned long long* prof = nullptr;
    if (!prof) { hipMalloc(&prof, 8 * 65536); }
    hipMemsetAsync(prof, 0, 8 * 65536, c->stream);
    d.prof = prof;
    g_prof = prof;
  }
#endif
  d.resbuf = c->resbuf; d.edge = c->edge; d.ticket = c->ticket; d.timeout = c->timeout;
  {
    /* intra pictures: k_intra's workgroups are persistent (k_intra.hip); with several pictures in flight every picture gets a
       share of the GPU's workgroup slots (2 per CU for this kernel) — enough for its active wavefront, not a slot per CTB */
    static const int grid_env = getenv("M355_INTRA_GRID") ? atoi(getenv("M355_INTRA_GRID")) : 0;
    static int slots = 0;
    if (!slots) { hipDeviceProp_t prop; slots = (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ? 2 * prop.multiProcessorCount : 512; }
    d.intra_grid = grid_env > 0 ? grid_env : std::max(64, slots / std::max(1, c->depth));
  }
  d.epoch = ++c->epoch;
  if (d.epoch == 0) d.epoch = ++c->epoch;
  d_out = d; want_sao_out = want_sao;
  return M355_OK;
}

static hipError_t frame_event(hipEvent_t* e)
{
  if (*e) return hipSuccess;
  return hipEventCreateWithFlags(e, hipEventDisableTiming);
}

/* The prediction half of a decode on the active lane: the metadata planes (read first by k_intra) are rasterised on the side
 * stream while the main stream runs job list -> inter prediction, which do not read them; the residual stage then runs in two
 * launches side by side — 32x32 + 16x16 blocks on the main stream, 8x8 + 4x4 on the side stream — and k_intra follows the join.
 * ev: the decode's timing events [1..4] (after meta jobs / inter / residual / intra) or nullptr. */
/* M355_PF_CLEAR_DST: a new picture starts from zero in the reference (image.cc:164); the planes being reconstructed are this
 * lane's working planes (SAO rewrites every sample of the destination) or the destination itself — then, for lists checked on
 * the device, by a kernel behind the decode's gate: a rejected picture must leave its destination frame untouched
 * (de265_mi355x.h, m355_decode_status). */
static void clear_target(m355_ctx* c, const DevPic& d, Frame* tgt, bool gated, hipStream_t st)
{
  for (int cc = 0; cc < 3; cc++) {
    if (!tgt->pw[cc]) continue;
    const size_t bytes = (size_t)tgt->stride[cc] * tgt->ph[cc] * tgt->bpp[cc];
    if (gated) m355_launch_clear_gated(d, tgt->plane[cc], bytes, st);
    else hipMemsetAsync(tgt->plane[cc], 0, bytes, st);
  }
}

static void launch_prediction(m355_ctx* c, const Resident& r, const DevPic& d, bool hbd, hipEvent_t* ev, bool with_intra = true)
{
  hipStream_t st = c->stream;
  /* an intra picture keeps to its lane's main stream: its side work (metadata planes, border plans: 0.07 ms) is nothing beside k_intra,
     and half as many streams compete for the runtime's hardware queues when many such pictures are in flight (C2 0.340 ms per picture
     = 1.50 M CTB64/s at depth 9, profiles/r03_v_*; forked: 0.59 at depth 8) — and so does a picture of up to 4K: the fork / join of the
     side stream is six packets (three event records, three waits) at about 2 us of pipeline time each, and what they buy — the metadata
     scatters and the second residual launch beside the main stream — is worth less than that once the kernels are short (three in
     flight, profiles/r04_al_*: C3 / C4 0.110 -> 0.098 / 0.100 ms on one stream, C5 0.347 -> 0.351) */
  const bool single = d.intra_dense || (long long)d.pp.width * d.pp.height <= 16ll << 20;
  const bool fused = d.res_map != nullptr;   /* prepare(): the residuals of inter CUs are added in k_inter_jobs' write-back */
  hipStream_t s2 = single ? st : c->stream2;
  /* the zero fill of the metadata planes rides in the picture's first main-stream launch (k_job_count), in FRONT of the fork: the
     side stream's scatters then start behind it — one launch less per inter picture (not with fused residuals: there the side
     stream starts with the residual stage, and the job count comes later) */
  /* (the fill is shared out over the launch's workgroups, one per 256 PBs: with a handful of them a fill of its own is faster;
     M355_CLEAR_IN_COUNT_MIN=<PBs> moves the threshold: tests/test_meta_merged_emu.py sends the CPU tier's small pictures down this path —
     and through the merged planes + job-list launch behind it — with 1) */
  static const int clear_min = getenv("M355_CLEAR_IN_COUNT_MIN") ? atoi(getenv("M355_CLEAR_IN_COUNT_MIN")) : 64 * 256;
  const bool clear_in_count = !fused && d.n_pbs >= std::max(1, clear_min);
  if (clear_in_count) m355_launch_job_count(d, true, st);
  if (fused) hipMemsetAsync(d.res_map, 0, (size_t)d.res_map_words * 4, st);
  if (!single) { hipEventRecord(c->ev_fork, st); hipStreamWaitEvent(s2, c->ev_fork, 0); }
  if (fused) {
    /* the residual stage reads nothing but the lists: it runs FIRST, side by side on the lane's two streams, beside the tail of
       the previous picture; its event order is [residual, meta, inter] (m355_timing_collect) */
    m355_launch_residual(d, hbd, false, s2);
    if (!single) hipEventRecord(c->ev_fork2, s2);
    m355_launch_residual(d, hbd, true, st);
    if (ev) hipEventRecord(ev[1], st);
  }
  /* transform edges and border plans in ONE launch (a packet less per picture: C3 0.098 -> 0.093 ms, profiles/r05_a_switches_merge.txt) */
  if (single && clear_in_count && (c->stages & M355_STAGE_INTRA)) {
    /* one stream: the planes' scatters and the job list are independent roles of ONE launch (k_meta_planes_jobs) */
    m355_launch_meta_planes_jobs(d, st);
    m355_launch_tu_plan(d, st);
  } else {
    if (c->stages & M355_STAGE_INTRA) {
      m355_launch_meta_planes(d, s2, clear_in_count, false);
      m355_launch_tu_plan(d, s2);
    } else m355_launch_meta_planes(d, s2, clear_in_count);
    if (clear_in_count) m355_launch_job_list(d, st); else m355_launch_meta_jobs(d, st);
  }
  if (ev) hipEventRecord(ev[fused ? 2 : 1], st);
  /* read-after-write on the reference frames: their last writers are waited for HERE, in front of the first kernel that reads a
     reference — the list copy, validation, metadata planes, job list (and fused residuals) of a picture run beside the tail
     (filters) of the picture it references */
  if (c->depth >= 2)
    for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
      Frame* f = r.hdr.ref_frames[i] >= 0 ? get_frame(c, r.hdr.ref_frames[i]) : nullptr;
      if (f) ev_wait(c, st, f->wr);
    }
  if (fused && !single) hipStreamWaitEvent(st, c->ev_fork2, 0);   /* the 8x8 + 4x4 tiles */
#ifdef M355_X_TILED
  if ((c->stages & M355_STAGE_INTER) && d.n_pbs)
    for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
      Frame* f = r.hdr.ref_frames[i] >= 0 ? get_frame(c, r.hdr.ref_frames[i]) : nullptr;
      if (!f || !f->tiled[0]) continue;
      if (!f->tiled_valid) {                 /* the conversion pass (its time is the experiment's cost side: k_tile_convert in the kernel trace) */
        for (int cc = 0; cc < 3; cc++) if (f->pw[cc]) m355_launch_tile_convert(f->plane[cc], f->stride[cc], f->pw[cc], f->ph[cc], f->bpp[cc], cc != 0, f->tiled[cc], f->tiles_w[cc], st);
        if (!f->ev_tiled) hipEventCreateWithFlags(&f->ev_tiled, hipEventDisableTiming);
        hipEventRecord(f->ev_tiled, st);
        f->tiled_valid = true;
      } else if (f->ev_tiled) hipStreamWaitEvent(st, f->ev_tiled, 0);
    }
#endif
  if (c->stages & M355_STAGE_INTER) m355_launch_inter(d, hbd, st);
  if (ev) hipEventRecord(ev[fused ? 3 : 2], st);
  if (!fused && (c->stages & M355_STAGE_RESIDUAL)) {
    /* inter residuals are added to the prediction samples: behind k_inter_jobs; the two launches side by side on the lane's two
       streams (one after the other on the main stream, without the second fork, was measured 1 % slower at C5 with three
       pictures in flight: 0.3573-0.3605 against 0.3538-0.3580 ms, profiles/r04_am_residual_streams_ab.txt) */
    hipStream_t sr = !single ? s2 : st;
    if (sr != st) { hipEventRecord(c->ev_fork2, st); hipStreamWaitEvent(s2, c->ev_fork2, 0); }
    m355_launch_residual(d, hbd, false, sr);
    m355_launch_residual(d, hbd, true, st);
  }
  if (!single) { hipEventRecord(c->ev_join, s2); hipStreamWaitEvent(st, c->ev_join, 0); }     /* join */
  if (!fused && ev) hipEventRecord(ev[3], st);
  if (with_intra && (c->stages & M355_STAGE_INTRA)) m355_launch_intra(d, hbd, st, clear_in_count);   /* (m355_decode_batch launches several pictures' intra stage as one kernel) */
  if (ev) hipEventRecord(ev[4], st);
}

/* write-after-write / write-after-read on the destination: waited for right before the first kernel that writes it — the SAO
   stage when SAO runs (everything before writes this lane's working planes), else the first stage */
static void dst_hazards(m355_ctx* c, Frame* dstf, bool piped)
{
  if (dstf->dl_pending) hipStreamWaitEvent(c->stream, dstf->ev_dl, 0);     /* (stays pending for the HOST until m355_frame_download_wait / m355_wait) */
  if (!piped) return;
  ev_wait(c, c->stream, dstf->wr);
  for (int k = 0; k < M355_MAX_LANES; k++) ev_wait(c, c->stream, dstf->rd[k]);
}

/* One decode = decode_pre (lane, hazards, validation, every stage in front of the intra stage [and, with_intra, that stage]) +
 * decode_post (in-loop filters, events, status slot).  m355_decode_batch runs the pre part of several intra pictures on their lanes,
 * ONE k_intra launch for all of them, then their post parts. */
struct DecodeState { DevPic d; bool want_sao = false; hipEvent_t* ev = nullptr; hipStream_t saved_stream = nullptr; bool swapped = false; };

/* front: PRE_ALL = everything up to and including the intra stage; PRE_NO_INTRA = without k_intra; PRE_HAZARDS = lane, hazards, validation and
   clearing only (m355_decode_batch launches the stages itself, one launch per stage for all its pictures) */
enum { PRE_ALL = 0, PRE_NO_INTRA = 1, PRE_HAZARDS = 2 };
static int decode_pre(m355_ctx* c, Resident& r, bool rotate, DecodeState& S, int mode, hipStream_t on_stream = nullptr)
{
  const bool with_intra = mode == PRE_ALL;
  if (r.sharded) return fail(M355_ERR_INVALID, "a sharded picture is decoded by phases (m355_decode_phase)");
  if (rotate && c->depth >= 2) select_lane(c, (c->active + 1) % c->depth);   /* consecutive pictures go round the lanes */
  /* which stream: an intra picture on lane 3.. takes the lane's class stream (lane_class_priority); the whole decode addresses
     c->stream, which is that stream until decode_post returns (a batch keeps to the lanes' ordinary streams: its pictures overlap
     inside one kernel, not through hardware queues) */
  {
    hipStream_t run = on_stream ? on_stream : c->stream;   /* (a batch on a stream of its own: its lanes lend their scratch only) */
    if (!on_stream && with_intra && r.dp.intra_dense && c->active >= 3 && lane_priorities_mode() == 2 && lane_class_priority(c->active) != 0) {
      if (!c->stream_hi) HIPCHK(hipStreamCreateWithPriority(&c->stream_hi, hipStreamNonBlocking, lane_class_priority(c->active)));
      run = c->stream_hi;
    }
    ev_wait(c, run, c->last);                              /* the lane's scratch and working planes (when its last decode ran on its other stream) */
    S.saved_stream = c->stream; S.swapped = run != c->stream;
    c->stream = run;
  }
  DevPic& d = S.d;
  int rc = prepare(c, r, d, S.want_sao);
  if (rc) return rc;
  const bool want_sao = S.want_sao;
  const m355_pic_params& pp = r.hdr.pp;
  const bool hbd = pp.bit_depth_luma > 8;
  const bool piped = c->depth >= 2;
  Frame* dstf = get_frame(c, r.hdr.dst_frame);
  if (piped) {
    /* read-after-write: the lists (uploaded on whichever lane was active); the reference frames' last writers: launch_prediction */
    ev_wait(c, c->stream, r.up);
  }
  hipStream_t st = c->stream;
  hipEvent_t* ev = nullptr;
  if (with_intra && c->timing_on) {                        /* (a batch's decodes are not stage-timed: their intra stage is shared) */
    if (c->ev_used >= 4096) c->ev_used = 0;                 /* bounded ring */
    while ((int)c->evs.size() < (c->ev_used + 1) * 7) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->evs.push_back(e); }
    ev = &c->evs[c->ev_used * 7];
    if ((int)c->ev_fused.size() <= c->ev_used) c->ev_fused.resize(c->ev_used + 1);
    c->ev_fused[c->ev_used] = d.res_map != nullptr;
    c->ev_used++;
    hipEventRecord(ev[0], st);
  }
  S.ev = ev;
  if (r.device_validate) m355_launch_validate(d, st);     /* a rejection gates THIS decode's kernels (epoch-tagged gate word) */
  if (!want_sao) dst_hazards(c, dstf, piped);
  if (pp.flags & M355_PF_CLEAR_DST) clear_target(c, d, want_sao ? &c->work : dstf, r.device_validate && !want_sao, st);
  if (mode != PRE_HAZARDS) launch_prediction(c, r, d, hbd, ev, with_intra);
  return M355_OK;
}

static int decode_post(m355_ctx* c, Resident& r, DecodeState& S, bool filters = true)
{
  struct StreamRestore { m355_ctx* c; DecodeState& S; ~StreamRestore() { if (S.swapped) c->stream = S.saved_stream; } } restore{c, S};
  const DevPic& d = S.d;
  const bool want_sao = S.want_sao;
  hipEvent_t* ev = S.ev;
  const m355_pic_params& pp = r.hdr.pp;
  const bool hbd = pp.bit_depth_luma > 8;
  const bool piped = c->depth >= 2;
  Frame* dstf = get_frame(c, r.hdr.dst_frame);
  hipStream_t st = c->stream;
  const bool deblock = filters && (c->stages & M355_STAGE_DEBLOCK) && (pp.flags & M355_PF_DEBLOCK_ENABLED);
  /* (the horizontal-edge pass inside the SAO kernel was built, is bit-exact on hardware and loses: C5 0.357 -> 0.392 ms per picture,
     SAO 50 -> 104 us for 24 us less deblocking — profiles/r05_a_switches_fuse_dbh.txt, tools/experiments/sao_fused_deblock_h.patch) */
  if (deblock) m355_launch_deblock(d, hbd, st);
  if (ev) hipEventRecord(ev[5], st);
  if (filters && want_sao) { dst_hazards(c, dstf, piped); m355_launch_sao(d, hbd, st); }
  if (ev) hipEventRecord(ev[6], st);
  /* ONE mark behind the decode's last kernel for everything that has to know when it is over: the lists' arenas, the destination
     frame's next reader / writer, the reference frames' next writer, the lane's next decode, the status slot */
  EvRef done;
  {
    const int rcm = ev_mark(c, st, &done);
    if (rcm) return rcm;
  }
  r.done = done; r.fresh = false;
#ifdef M355_X_TILED
  dstf->tiled_valid = false;
#endif
  dstf->wr_stream = st;
  dstf->wr = done;
  for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
    Frame* f = r.hdr.ref_frames[i] >= 0 ? get_frame(c, r.hdr.ref_frames[i]) : nullptr;
    if (f) f->rd[c->active] = done;
  }
  if (ev) c->timed = true;
  {
    /* this decode's status slot; a device-validated decode also brings its lane's gate words back — behind the mark the dependent
       decodes wait on, with a mark of its own: nobody waits for this copy but m355_decode_status / m355_wait */
    m355_ctx::Status& s = c->status[++c->serial % M355_STATUS_RING];
    if (s.serial && s.validated && !s.reported) {
      /* the slot's previous decode (M355_STATUS_RING submits ago) was never asked about: resolve it before its words are
         overwritten — a rejection must not get lost (m355_wait promises to report it) */
      ev_sync(c, s.done);
      if (c->status_words[4 * (s.serial % M355_STATUS_RING) + 1] == s.epoch) { if (!c->lost_count++) c->lost_first = s.serial; }
    }
    s.serial = c->serial; s.epoch = d.epoch; s.validated = r.device_validate; s.reported = false;
    s.done = done;
    if (r.device_validate) {
      if (!c->status_words) HIPCHK(hipHostMalloc(&c->status_words, 16 * M355_STATUS_RING, hipHostMallocDefault));
      hipMemcpyAsync(c->status_words + 4 * (c->serial % M355_STATUS_RING), c->timeout, 16, hipMemcpyDeviceToHost, st);
      const int rcm = ev_mark(c, st, &s.done);
      if (rcm) return rcm;
    }
  }
  c->last = done; c->last_stream = st;                       /* (the lane's next decode may run on the lane's other stream) */
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(M355_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  {
    /* M355_DEBUG_TIMEOUT=1: name the decode whose intra stage gave up a wait (diagnostic: serialises the pipeline) */
    static const bool dbg = getenv("M355_DEBUG_TIMEOUT") && atoi(getenv("M355_DEBUG_TIMEOUT"));
    if (dbg) {
      hipStreamSynchronize(st);
      uint32_t t = 0;
      hipMemcpy(&t, c->timeout, 4, hipMemcpyDeviceToHost);
      if (t) fprintf(stderr, "m355: decode %llu (epoch %u, %d pbs, %d ibs, %d cus, lane %d): intra wait gave up\n", c->serial, d.epoch, d.n_pbs, d.n_ibs, d.n_cus, c->active);
    }
  }
  return M355_OK;
}

static int decode(m355_ctx* c, Resident& r, bool rotate = true)
{
  DecodeState S;
  int rc = decode_pre(c, r, rotate, S, PRE_ALL);
  if (rc) { if (S.swapped) c->stream = S.saved_stream; return rc; }
  return decode_post(c, r, S);
}

/* status of one finished decode from its ring slot: M355_OK, or M355_ERR_INVALID with the rejected record in the message */
static int status_of(m355_ctx* c, m355_ctx::Status& s)
{
  if (!s.validated) return M355_OK;
  const uint32_t* w = c->status_words + 4 * (s.serial % M355_STATUS_RING);
  if (w[1] != s.epoch) return M355_OK;                       /* the lane's last rejected decode is another one */
  const unsigned long long key = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
  uint32_t bad = (uint32_t)key;
  if ((uint32_t)(key >> 32) != ~s.epoch) bad = 0;            /* (cannot happen: gate and key are written together) */
  static const char* const names[8] = {"?", "cu", "tu", "pb", "weight", "rb", "ib", "?"};
  s.reported = true;
  return fail(M355_ERR_INVALID, "picture %llu: %s %u rejected by the device-side list validation (the picture was not decoded)", s.serial, names[(bad >> 28) & 7], bad & 0x0FFFFFFFu);
}

unsigned long long m355_last_serial(m355_ctx* c) { return c->serial; }

int m355_decode_status(m355_ctx* c, unsigned long long serial)
{
  if (serial == 0 || serial > c->serial) return fail(M355_ERR_INVALID, "no decode with serial %llu", serial);
  m355_ctx::Status& s = c->status[serial % M355_STATUS_RING];
  if (s.serial != serial) return fail(M355_ERR_INVALID, "decode %llu is older than the last %d decodes: its status is no longer kept (m355_wait reports rejections)", serial, M355_STATUS_RING);
  hipSetDevice(c->device);
  const hipError_t q = ev_query(c, s.done);
  if (q == hipErrorNotReady) return M355_ERR_BUSY;
  if (q != hipSuccess) return fail(M355_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
  return status_of(c, s);
}

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
  for (int k = 0; k <= last; k++) {
    int rc = m355_decode_phase(c, h, k, k < 4 ? r.xb[k] : nullptr);
    if (rc) return rc;
    if (N <= 1 || k >= last) continue;                       /* a single rank owns every tile: nothing to exchange */
    if (k < 3) {
      if (!r.peers.empty() && (rc = c->comm.halo_sum(c->comm.user, r.xb[k], r.xb_bytes[k], r.peers.data(), (int)r.peers.size(), r.xscratch, (void*)c->stream)))
        return fail(M355_ERR_HIP, "halo exchange %d failed (%d)", k, rc);
    } else if ((rc = c->comm.all_gather(c->comm.user, r.xb[3], r.xb_bytes[3] / (size_t)N, r.shard_rank, N, (void*)c->stream)))
      return fail(M355_ERR_HIP, "tile all-gather failed (%d)", rc);
  }
  return M355_OK;
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
    if (which < 3) return r.peers.empty() ? 0 : c->comm.halo_sum(c->comm.user, r.xb[which], r.xb_bytes[which], r.peers.data(), (int)r.peers.size(), r.xscratch, (void*)c->stream);
    return c->comm.all_gather(c->comm.user, r.xb[3], r.xb_bytes[3] / (size_t)r.shard_n, r.shard_rank, r.shard_n, (void*)c->stream);
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
    for (void*& b : r.xb) { if (b) hipFree(b); b = nullptr; }
    if (r.xscratch) { hipFree(r.xscratch); r.xscratch = nullptr; }
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
  resident_free(c->resident[h]);
  return M355_OK;
}
int m355_decode_resident(m355_ctx* c, int h)
{
  if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used) return fail(M355_ERR_INVALID, "bad picture handle");
  return decode(c, c->resident[h]);
}
/* Several independent intra pictures as ONE intra stage: every picture's front part (validation, residuals, border plans) on its own
 * lane, then one k_intra<BATCH> launch over all their CTB wavefronts, then every picture's filters on its lane again.  What more
 * lanes buy an intra picture — other pictures' CTBs filling the GPU while its own wavefront is narrow — without one hardware queue per
 * picture. */
int m355_decode_batch(m355_ctx* c, const int* handles, int n)
{
  if (n < 1 || !handles) return fail(M355_ERR_INVALID, "m355_decode_batch: no pictures");
  if (n > std::max(1, c->depth)) return fail(M355_ERR_INVALID, "m355_decode_batch: %d pictures on %d lanes (m355_set_pipeline_depth)", n, c->depth);
  for (int k = 0; k < n; k++) {
    const int h = handles[k];
    if (h < 0 || h >= (int)c->resident.size() || !c->resident[h].used) return fail(M355_ERR_INVALID, "bad picture handle");
    for (int j = 0; j < k; j++) if (handles[j] == h) return fail(M355_ERR_INVALID, "m355_decode_batch: picture %d twice in one batch", h);
    const Resident& r = c->resident[h];
    const m355_pic_params &a = r.hdr.pp, &b = c->resident[handles[0]].hdr.pp;
    if (r.sharded) return fail(M355_ERR_INVALID, "a sharded picture is decoded by phases (m355_decode_phase)");
    /* (pictures of a batch must not reference one another: their frames' writer events are recorded behind the shared launch) */
    if (!r.dp.intra_dense || r.dp.n_pbs > 0) return fail(M355_ERR_INVALID, "m355_decode_batch: picture %d is not an intra picture", h);
    if (a.chroma_format_idc != b.chroma_format_idc || (a.bit_depth_luma > 8) != (b.bit_depth_luma > 8))
      return fail(M355_ERR_INVALID, "m355_decode_batch: the pictures differ in chroma format or sample type");
    for (int j = 0; j < k; j++)
      if (c->resident[handles[j]].hdr.dst_frame == r.hdr.dst_frame) return fail(M355_ERR_INVALID, "m355_decode_batch: two pictures into frame %d", r.hdr.dst_frame);
  }
  hipSetDevice(c->device);
  if (n == 1 || !(c->stages & M355_STAGE_INTRA)) {
    for (int k = 0; k < n; k++) { int rc = decode(c, c->resident[handles[k]]); if (rc) return rc; }
    return M355_OK;
  }
  DecodeState S[M355_MAX_LANES];
  int lane[M355_MAX_LANES];
  int rc_late = M355_OK, n_ok = 0;
  /* Where the batch runs.  M355_BATCH_STREAMS=N (default 4): whole batches go round N streams of their own — front parts, the shared
     launch and the filters of ONE batch are one stream's worth of work (they depend on one another anyway), consecutive batches on
     different lanes overlap on different hardware queues; the lanes lend their scratch and working planes.  =0: every picture's front
     part and filters on its own lane's stream, the shared launch on the first lane's (measured slower: 16 lanes' small kernels
     serialise on the runtime's four hardware queues AND with the batch, profiles/r04_n_c2_batch.txt). */
  static const int streams_env = getenv("M355_BATCH_STREAMS") ? std::min(4, std::max(0, atoi(getenv("M355_BATCH_STREAMS")))) : -1;
  /* as many streams as batches of this size fit the lanes side by side (batches that share lanes run one after the other anyway) */
  const int n_streams = streams_env >= 0 ? streams_env : std::min(4, std::max(1, c->depth / n));
  hipStream_t bs = nullptr;
  if (n_streams > 0) {
    const int j = (int)(c->batch_count++ % (unsigned)n_streams);
    if (!c->batch_stream[j]) HIPCHK(hipStreamCreateWithFlags(&c->batch_stream[j], hipStreamNonBlocking));
    bs = c->batch_stream[j];
  }
  for (int k = 0; k < n; k++) {
    Resident& r = c->resident[handles[k]];
    const int rc = decode_pre(c, r, true, S[k], bs ? PRE_HAZARDS : PRE_NO_INTRA, bs);
    if (S[k].swapped) { c->stream = S[k].saved_stream; S[k].swapped = false; }   /* (select_lane parks c->stream with the lane) */
    if (rc) { rc_late = rc; break; }                 /* the pictures in front of it are finished as a shorter batch */
    lane[k] = c->active;
    if (!bs) {
      if (!c->batch_ev_pre[k] && hipEventCreateWithFlags(&c->batch_ev_pre[k], hipEventDisableTiming) != hipSuccess) return fail(M355_ERR_HIP, "hipEventCreate failed");
      hipEventRecord(c->batch_ev_pre[k], c->stream);
    }
    n_ok++;
  }
  if (!n_ok) return rc_late;
  m355_ctx::BatchSlot& b = c->batch[c->batch_next];
  c->batch_next = (c->batch_next + 1) % M355_BATCH_RING;
  if (!b.dev) {
    HIPCHK(hipHostMalloc((void**)&b.host, sizeof(DevPic) * M355_MAX_LANES, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&b.dev, sizeof(DevPic) * M355_MAX_LANES + 64));
    b.ticket = (uint32_t*)((uint8_t*)b.dev + sizeof(DevPic) * M355_MAX_LANES);
    HIPCHK(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
  }
  if (b.pending) { hipEventSynchronize(b.ev); b.pending = false; }     /* (M355_BATCH_RING batches ago) */
  select_lane(c, lane[0]);
  hipStream_t st0 = bs ? bs : c->stream;
  int max_work = 0; long total = 0;
  for (int k = 0; k < n_ok; k++) {
    b.host[k] = S[k].d;
    max_work = std::max(max_work, S[k].d.n_intra_work); total += S[k].d.n_intra_work;
    if (k && !bs) hipStreamWaitEvent(st0, c->batch_ev_pre[k], 0);
  }
  hipMemcpyAsync(b.dev, b.host, sizeof(DevPic) * n_ok, hipMemcpyHostToDevice, st0);
  hipMemsetAsync(b.ticket, 0, 4, st0);
  const bool hbd = c->resident[handles[0]].hdr.pp.bit_depth_luma > 8;
  const HostBatch hb{b.host, b.dev, n_ok, n_ok >= 32 ? 0xFFFFFFFFu : (1u << n_ok) - 1u};
  if (bs) {
    /* the stages in front of the intra stage, each ONE launch over the batch's pictures (launch_prediction's order for an intra
       picture: metadata planes, border plans, 8x8 + 4x4 residuals, 32x32 + 16x16 residuals) */
    m355_launch_meta_planes_batch(hb, st0);
    if (c->stages & M355_STAGE_INTRA) m355_launch_intra_plan_batch(hb, st0);
    if (c->stages & M355_STAGE_RESIDUAL) { m355_launch_residual_batch(hb, hbd, false, st0); m355_launch_residual_batch(hb, hbd, true, st0); }
  }
  {
    static const int grid_env = getenv("M355_INTRA_GRID") ? atoi(getenv("M355_INTRA_GRID")) : 0;
    static int slots = 0;
    if (!slots) { hipDeviceProp_t prop; slots = (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ? 2 * prop.multiProcessorCount : 512; }
    /* persistent workgroups of the shared launch: a batch on its own takes every slot of the GPU (4 pictures, 64 -> 512 workgroups:
       0.573 -> 0.484 ms per picture); batches side by side take what covers their pictures' widest wavefronts (CTB (x, y) runs at
       step x + 2y: 16 for 1080p) or half their share of the slots — more only spin and crowd the other batches' kernels (32 pictures
       as 4 x 8, 512 -> 128 workgroups each: 0.161 -> 0.126; 16 as 4 x 4, 128 -> 64: 0.181 -> 0.160; profiles/r04_n_c2_batch.txt) */
    int widest = 0;
    for (int k = 0; k < n_ok; k++) widest += std::min(S[k].d.ctbH, (S[k].d.ctbW + 1) / 2) + 1;
    const int grid = (int)std::min<long>(std::max<long>(total, 1), grid_env > 0 ? grid_env : std::min(slots, n_streams <= 1 ? slots : std::max(widest, slots / (2 * n_streams))));
    m355_launch_intra_batch(S[0].d, hbd, b.dev, n_ok, max_work, b.ticket, grid, st0);
  }
  hipEventRecord(b.ev, st0); b.pending = true;
  if (bs) {
    /* the in-loop filters of the whole batch: two deblocking launches, one SAO launch (its pictures' destination hazards in front) */
    uint32_t dbk = 0, sao = 0;
    for (int k = 0; k < n_ok; k++) {
      if ((c->stages & M355_STAGE_DEBLOCK) && (c->resident[handles[k]].hdr.pp.flags & M355_PF_DEBLOCK_ENABLED)) dbk |= 1u << k;
      if (S[k].want_sao) sao |= 1u << k;
    }
    if (dbk) m355_launch_deblock_batch(HostBatch{b.host, b.dev, n_ok, dbk}, hbd, st0);
    if (sao) {
      hipStream_t keep = c->stream;
      c->stream = bs;
      for (int k = 0; k < n_ok; k++) if ((sao >> k) & 1u) dst_hazards(c, get_frame(c, c->resident[handles[k]].hdr.dst_frame), c->depth >= 2);
      c->stream = keep;
      m355_launch_sao_batch(HostBatch{b.host, b.dev, n_ok, sao}, hbd, st0);
    }
    hipEventRecord(b.ev, st0);     /* the filter launches read the slot's records too: the slot is free behind THEM */
  }
  for (int k = 0; k < n_ok; k++) {
    select_lane(c, lane[k]);
    if (bs) { S[k].saved_stream = c->stream; c->stream = bs; S[k].swapped = true; }      /* (decode_post puts the lane's stream back) */
    else if (k) hipStreamWaitEvent(c->stream, b.ev, 0);
    const int rc = decode_post(c, c->resident[handles[k]], S[k], !bs);
    if (rc && !rc_late) rc_late = rc;
  }
  return rc_late;
}
int m355_set_stages(m355_ctx* c, int mask) { c->stages = mask & M355_STAGE_ALL; return M355_OK; }

int m355_timing_reset(m355_ctx* c) { c->ev_used = 0; c->timing_on = true; return M355_OK; }

/* averages over every decode enqueued since m355_timing_reset(); waits for them to finish */
int m355_timing_collect(m355_ctx* c, int* n_decodes, float* total_ms, float stage_ms[6])
{
  c->timing_on = false;
  if (c->ev_used == 0) return fail(M355_ERR_INVALID, "nothing decoded since the last timing reset");
  hipSetDevice(c->device);
  HIPCHK(sync_all(c));
  double tot = 0, st[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < c->ev_used; k++) {
    hipEvent_t* ev = &c->evs[k * 7];
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[6])); tot += ms;
    /* stage order of the events: [meta, inter, residual, intra, deblock, sao], or with fused residuals [residual, meta, inter, ...] */
    static const int order[2][6] = {{0, 1, 2, 3, 4, 5}, {2, 0, 1, 3, 4, 5}};
    const int* o = order[k < (int)c->ev_fused.size() && c->ev_fused[k] ? 1 : 0];
    for (int i = 0; i < 6; i++) { HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); st[o[i]] += ms; }
  }
  if (n_decodes) *n_decodes = c->ev_used;
  if (total_ms) *total_ms = (float)(tot / c->ev_used);
  if (stage_ms) for (int i = 0; i < 6; i++) stage_ms[i] = (float)(st[i] / c->ev_used);
  return M355_OK;
}

} /* extern "C" */
