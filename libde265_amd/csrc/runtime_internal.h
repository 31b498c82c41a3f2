/*
 * runtime_internal.h — what the parts of the host runtime share: the context and its lanes, frames, resident pictures, the event
 * ring, the layout of an uploaded picture, and the prototypes of the functions one part calls in another.
 *   runtime.hip        context / lanes / frames / hashes / arenas / submit / wait / timing (the C ABI's entry points of those)
 *   runtime_upload.hip host pool, validation, intra schedule, upload (lists -> one pinned arena -> device)
 *   runtime_decode.hip prepare, the per-picture launch sequence, decode status, m355_decode_batch
 *   runtime_shard.hip  tile sharding: phases, exchanges, the in-process group, the built-in RCCL transport
 */
#ifndef M355_RUNTIME_INTERNAL_H
#define M355_RUNTIME_INTERNAL_H
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


extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);     /* sets the thread's last error text, returns `code` */
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(M355_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

#define M355_STATUS_RING 64
#define M355_BATCH_RING 16 /* m355_decode_batch: picture-record arrays in flight (the host runs this many batches ahead) */
#define M355_MAX_LANES 32  /* pictures in flight per context (m355_set_pipeline_depth) */
#define M355_TRANSIENT_MAX 12 /* staging arenas of m355_submit_picture (m355_ctx::transient_ring) */

/* ---- THE SCHEDULE TABLE: every size / depth at which the runtime picks another launch order, in one place, with the measurement that set it
 * (all A/Bs of two settings in ONE gpurun call on one box; boxes of the pool differ by +-5 %).  Each row is a separate correctness surface: the
 * forced-schedule tests (tests/test_gpu_chain_forced.py, tests/test_emu_chain.py, tests/test_chain_residuals_emu.py, tests/test_meta_merged_emu.py)
 * walk every one of them deterministically. ---- */
namespace sched {
/* pictures of up to this many luma samples run ALL their launches on the lane's main stream (no fork / join packets: six packets at ~2 us each are
   worth more than what the side stream buys once the kernels are short): C3 / C4 0.110 -> 0.098 / 0.100 ms three in flight, C5 0.347 -> 0.351
   (profiles/r04_al_*).  Above it the metadata planes + border plans and the small-block residual launch run on the lane's side stream. */
constexpr long long one_stream_max_samples = 16ll << 20;
/* from this many prediction blocks on, the zero fill of the metadata planes rides in k_job_count's workgroups (one per 256 PBs) and the planes + job list
   are roles of one launch; below it a fill of its own is faster (a handful of workgroups cannot spread it): profiles/r04_ak_*, r05_v29_* */
constexpr int clear_in_count_min_pbs = 64 * 256;
/* transform edges + border plans inside the residual launch: only when the context decodes ONE picture at a time (C3 0.1494 -> 0.1427 ms, C4 0.1627 ->
   0.1564; with lanes the separate launch runs beside other pictures' kernels: C3 0.0688 -> 0.0697, a chain's picture 0.132 -> 0.136, profiles/r05_v30_*) */
constexpr int tu_plan_in_residuals_max_depth = 1;
/* k_intra's halo keeper (13th wave): with one picture in flight, or for an intra picture when no other lane is busy (C2 1.22 -> 0.90 ms one at a time;
   with three in flight the 13-wave workgroup crowds the other pictures' kernels out: 0.468 -> 0.536 ms, profiles/r05_v14_*) */
constexpr int intra_keeper_max_depth = 1;
/* a dependent chain's picture: front part on a spare lane from three lanes on (two lanes: the whole picture follows its reference onto that lane: C3 0.162 ->
   0.178 ms when split there, profiles/r05_v27_*); its residual transforms move into the front part (int16 tiles + k_residual_add) on one-stream lanes only
   (C3 0.131 -> 0.124 ms, C4 0.148 -> 0.140; C5 0.42-0.44 -> 0.44-0.45: profiles/r05_v31_*) */
constexpr int chain_split_min_depth = 3;
constexpr int chain_residual_tiles_min_depth = 3;
/* lanes 3.. decode INTRA pictures on a stream of the next priority class (own hardware queues): C2 0.340 ms per picture at depth 9 (profiles/r03_v_*) */
constexpr int intra_class_first_lane = 3;
}

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
  /* m355_group_decode: x3_read[q] = rank q has copied this handle's gather buffer xb[3] (recorded by q on its own stream, created on q's
     device); the NEXT decode of this handle waits for them before it repacks the buffer — per buffer, not per rank: handles need not
     rotate in step with the lane streams */
  std::vector<hipEvent_t> x3_read;
  unsigned long long xb_epoch = 0;   /* which allocation xb[] is (counted per process: the interprocess transport exports a new one again) */
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
  uint32_t* job_base = nullptr; /* per 256-PB chunk the first job of each range + the three range ends (k_job_count / k_job_scan) */
  size_t cap_cb = 0, cap_u4 = 0, cap_edge = 0, cap_cuf = 0, cap_res = 0, cap_jobs = 0, cap_sao = 0, cap_iplan = 0, cap_jobbase = 0;
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
  uint32_t* job_base = nullptr;
  size_t cap_cb = 0, cap_u4 = 0, cap_edge = 0, cap_cuf = 0, cap_res = 0, cap_jobs = 0, cap_sao = 0, cap_iplan = 0, cap_jobbase = 0;
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
  void* ipc = nullptr;               /* interprocess transport (m355_shard_ipc_init, runtime_ipc.hip) */
  int xchg_h = -1, xchg_k = -1;      /* m355_decode_sharded: the picture handle / exchange the callbacks are being called for */
  std::vector<hipEvent_t> evs;  /* 7 events per timed decode (ring grows on demand) */
  int ev_used = 0;             /* decodes recorded since the last m355_timing_reset */
  bool timed = false;
  bool timing_on = false;      /* between m355_timing_reset and m355_timing_collect: decodes record their seven stage events */
  uint32_t* hash_acc = nullptr; /* m355_frame_hash accumulators */
  /* pinned staging buffer of the blocking frame transfers (m355_frame_upload / _download / _fill, MD5 of m355_frame_hash): the copy itself is queued on the
     stream that last wrote the frame, between pinned memory and the frame (frame_stage_* in runtime.hip) */
  void* stage = nullptr; size_t stage_bytes = 0;
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
  X(resbuf) X(jobs) X(sao_nb) X(iplan) X(job_base) X(cap_jobbase) X(cap_iplan) X(cap_cb) X(cap_u4) X(cap_edge) X(cap_cuf) X(cap_res) X(cap_jobs) X(cap_sao)

/* ---- shared between the parts ---- */
struct TileRect { int x0, y0, x1, y1; };   /* luma samples */
struct Seg { const void* src; size_t bytes; size_t ofs; };
struct Lay {
  Seg seg[32];
  int ns;
  size_t total;
  int i_sl, i_ct, i_cu, i_tu, i_pb, i_wt, i_rb[4], i_ibin, i_ib, i_il, i_co, i_pc, i_sc, i_ts, i_rs, i_ti, i_iw, i_dp, i_ow;
};
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
extern Rccl g_rccl;            /* runtime_shard.hip */
int ipc_before_repack(m355_ctx* c, int h, hipStream_t st);      /* runtime_ipc.hip: hooks of m355_decode_sharded */
int ipc_end_picture(m355_ctx* c, int rc_own);
int ipc_before_free(m355_ctx* c, int h);
extern "C" int m355_shard_ipc_close(m355_ctx* c);
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

extern "C" {
void clear_target(m355_ctx* c, const DevPic& d, Frame* tgt, bool gated, hipStream_t st);
int copy_tiles(m355_ctx* c, const m355_pic_params& pp, Frame* f, int k0, int k1, int skip, int nranks, char* xbuf, size_t slot, bool to_slot);
int decode(m355_ctx* c, Resident& r, bool rotate = true);
void dst_hazards(m355_ctx* c, Frame* dstf, bool piped);
int ev_mark(m355_ctx* c, hipStream_t st, EvRef* out);
hipError_t ev_query(m355_ctx* c, const EvRef& r);
hipError_t ev_sync(m355_ctx* c, const EvRef& r);
void ev_wait(m355_ctx* c, hipStream_t st, const EvRef& r);
Frame* get_frame(m355_ctx* c, int h);
void halo_layout(const m355_pic_params& pp, HaloLayout& h);
int lane_class_priority(int index);
int lane_priorities_mode();
void launch_prediction(m355_ctx* c, const Resident& r, const DevPic& d, bool hbd, hipEvent_t* ev, bool with_intra = true, hipStream_t chain = nullptr, Frame* hazard_dst = nullptr);
void make_layout(const m355_arena_caps& k, int nCtb, int halo_units, bool sharded, bool with_ib_input, Lay& L);
int prepare(m355_ctx* c, Resident& r, DevPic& d_out, bool& want_sao_out);
void select_lane(m355_ctx* c, int lane);
size_t slot_bytes(const m355_pic_params& pp, int nranks);
int status_of(m355_ctx* c, m355_ctx::Status& s);
hipError_t sync_all(m355_ctx* c);
int upload(m355_ctx* c, Resident& r, const m355_picture* pic);
hipError_t frame_event(hipEvent_t* e);
}

#endif
