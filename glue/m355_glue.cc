/*
 * m355_glue.cc — the reference-side binding of the MI355X backend (what a libde265 maintainer adds).
 *
 * Linked with the reference's OWN objects (compiled from /root/reference where it lies, see glue/Makefile)
 * into glue/_build/libde265.so: a libde265 with the unchanged public de265.h ABI whose pixel path runs on the
 * GPU.  Nothing of the reference is copied or edited; the reference functions below are REPLACED at link time
 * (their definitions in the reference objects are weakened with objcopy, ours win, the linker drops the originals):
 *
 *   scale_coefficients                  transform.cc:645   -> record one m355_rb + the parser's sparse levels
 *   decode_intra_prediction             intrapred.cc:321   -> record one m355_ib (decode order)
 *   generate_inter_prediction_samples   motion.cc:288      -> record one m355_pb with the host-side decisions
 *   decoder_context::run_postprocessing_filters_sequential / _parallel   decctx.cc:1783 / 1811
 *                                                          -> walk the picture's metadata, m355_submit_picture()
 *   process_sei                         sei.cc:436         -> decoded picture hash SEIs are checked against the DEVICE frame (m355_frame_hash)
 *   de265_new_decoder / de265_free_decoder / de265_get_image_plane   de265.cc:254-281, 729-738 (renamed in de265.o, wrapped here)
 *                                                          -> backend context; a picture is downloaded when its samples are asked for
 * deblock.cc, sao.cc, fallback-*.cc and x86/*.cc are not compiled at all (the entry points the parser's objects still name
 * exist here as traps; init_acceleration_functions_fallback fills the decoder's table with counting traps).
 *
 * The host keeps doing everything it did before — NAL / CABAC parsing, MV and QP derivation, DPB management,
 * output order — and NO pixel arithmetic: every slot of the decoder's acceleration table is replaced by a trap
 * that counts (m355_glue_cpu_pixel_calls(), 0 in the tests).  Pictures live on the device keyed by DPB index
 * (dpb.cc:194-281); the host planes, handed out from a pinned pool through de265_set_image_allocation_functions
 * (de265.h:350-368), are written only when a picture is output (or when the stream carries a picture hash SEI to check).
 *
 * Threading: the recording hooks run on the decoder's worker threads (threads.h); every thread appends to its own
 * lists.  When a picture is parsed (img->wait_for_completion has returned) the decoder's thread hands the lists and a snapshot
 * of the DPB to the glue's WORKER thread and goes on parsing the next picture; the worker walks the picture's metadata, writes
 * the lists into the backend's pinned arena and submits.  Whoever needs the picture afterwards (application, SEI check, the DPB
 * recycling the image) first waits for its job; all calls into the backend context are serialised by one mutex.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "libde265/de265.h"
#include "libde265/decctx.h"
#include "libde265/fallback.h"
#include "libde265/image.h"
#include "libde265/intrapred.h"
#include "libde265/motion.h"
#include "libde265/pps.h"
#include "libde265/sei.h"
#include "libde265/slice.h"
#include "libde265/sps.h"
#include "libde265/transform.h"
#include "libde265/deblock.h"
#include "libde265/sao.h"

#include "de265_mi355x.h"

/* the reference's own entry points, renamed in de265.o by glue/Makefile (objcopy --redefine-sym) */
extern "C" {
de265_decoder_context* m355ref_de265_new_decoder(void);
de265_error m355ref_de265_free_decoder(de265_decoder_context*);
const uint8_t* m355ref_de265_get_image_plane(const struct de265_image*, int, int*);
void m355ref_de265_set_parameter_bool(de265_decoder_context*, enum de265_param, int);
int m355ref_de265_get_parameter_bool(de265_decoder_context*, enum de265_param);
}

namespace {

/* ------------------------------------------------------------------ backend library (C ABI) ------ */

#define M355_FUNCS(X) \
  X(m355_last_error) X(m355_device_count) X(m355_create) X(m355_destroy) X(m355_frame_create) X(m355_frame_destroy) \
  X(m355_frame_upload) X(m355_frame_download) X(m355_submit_picture) X(m355_wait) X(m355_set_pipeline_depth) \
  X(m355_host_alloc) X(m355_host_free) X(m355_frame_hash) X(m355_arena_begin) X(m355_last_serial) X(m355_decode_status) \
  X(m355_frame_download_async) X(m355_frame_download_wait) \
  X(m355_group_create) X(m355_group_destroy) X(m355_group_decode) X(m355_picture_upload) X(m355_picture_replace) X(m355_shard_owner_of_tile) \
  X(m355_picture_arena_begin)

struct Api {
  void* handle = nullptr;
  std::string path;
#define DECL(n) decltype(&::n) n = nullptr;
  M355_FUNCS(DECL)
#undef DECL
};

Api* api()
{
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string path;
    if (const char* e = getenv("M355_LIB")) path = e;
    else {
      Dl_info info;
      if (dladdr((void*)&api, &info) && info.dli_fname) {
        path = info.dli_fname;                                   /* <repo>/glue/_build/libde265.so */
        for (int up = 0; up < 3; up++) { size_t k = path.rfind('/'); path = k == std::string::npos ? "." : path.substr(0, k); }
        path += "/libde265_amd/libde265_mi355x.so";
      }
    }
    a.path = path;
    a.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!a.handle) { fprintf(stderr, "libde265 (MI355X glue): cannot load the backend %s: %s\n", path.c_str(), dlerror()); return; }
#define LOAD(n) a.n = (decltype(a.n))dlsym(a.handle, #n); if (!a.n) { fprintf(stderr, "libde265 (MI355X glue): backend lacks %s\n", #n); a.handle = nullptr; return; }
    M355_FUNCS(LOAD)
#undef LOAD
  });
  return a.handle ? &a : nullptr;
}

std::atomic<long long> g_cpu_pixel_calls(0);

/* ------------------------------------------------------------------ per-thread recording --------- */

/* coverage counters of the recorder (tests: a stream that carries a feature must have driven the branch that maps it) */
enum { FEAT_PCM_CU, FEAT_WEIGHTED_PB, FEAT_BYPASS_RB, FEAT_SKIP_RB, FEAT_RDPCM_RB, FEAT_ROTATE_RB, FEAT_SCALING_RB, FEAT_CROSS_COMP_RB,
       FEAT_MULTI_SLICE_PIC, FEAT_WEIGHTED_PB_LATER_SLICE, FEAT_NO_BOUNDARY_FILTER_IB, FEAT_FILL_PB, FEAT_CHROMA_422_RB, FEAT_CHROMA_444_RB,
       FEAT_MONO_PIC, FEAT_DEBLOCK_OFF_SLICE, M355_GLUE_N_FEATURES };
std::atomic<long long> g_feat[M355_GLUE_N_FEATURES];
enum { RBF_ROTATE_PENDING = 0x80 };   /* m355_rb.flags bit the recorder alone uses (scale_coefficients): never reaches the library */

struct Run { uint32_t ctb, start, count; };

struct ThreadRec {
  const decoder_context* owner = nullptr;
  uint32_t img_id = 0xFFFFFFFFu;
  const void* thread = nullptr;     /* the one thread that appends to these lists */
  std::vector<m355_pb> pbs;
  std::vector<m355_rb> rbs[4];
  std::vector<m355_ib> ibs;
  std::vector<uint32_t> coeffs;
  std::vector<Run> runs;
  uint32_t res_len = 0;
  int last_ib = -1;                 /* the intra block just predicted: its residual (if any) comes next (slice.cc:3489-3508) */
  int luma_rb = -1;                 /* cross-component prediction: the transform unit's luma block in its size bin */
  int skipped_pbs = 0;              /* prediction units the reference leaves unwritten (motion.cc warnings) */
  bool any_weighted = false;
  long long feat[M355_GLUE_N_FEATURES] = {};   /* which recorder branches this thread's blocks took (m355_glue_feature_counts) */
  void clear()
  {
    pbs.clear(); for (auto& v : rbs) v.clear(); ibs.clear(); coeffs.clear(); runs.clear();
    res_len = 0; last_ib = -1; luma_rb = -1; skipped_pbs = 0; any_weighted = false; img_id = 0xFFFFFFFFu; owner = nullptr; thread = nullptr;
    for (long long& f : feat) f = 0;
  }
};

/* pinned plane pool (get_buffer / release_buffer run once per picture: dpb.cc:271 -> image.cc:211) */
struct PlanePool {
  std::mutex mu;
  std::multimap<size_t, void*> free_;
  std::map<void*, size_t> size_of;
  void* get(size_t bytes)
  {
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = free_.find(bytes);
      if (it != free_.end()) { void* p = it->second; free_.erase(it); return p; }
    }
    void* p = api()->m355_host_alloc(bytes);
    if (p) { std::lock_guard<std::mutex> g(mu); size_of[p] = bytes; }
    return p;
  }
  void put(void* p)
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = size_of.find(p);
    if (it != size_of.end()) free_.insert(std::make_pair(it->second, p));
  }
  void drain()
  {
    std::lock_guard<std::mutex> g(mu);
    for (auto& kv : size_of) api()->m355_host_free(kv.first);
    size_of.clear(); free_.clear();
  }
};

struct Glue {
  decoder_context* dctx = nullptr;
  m355_ctx* mctx = nullptr;
  std::mutex mu;
  uint32_t cur_id = 0xFFFFFFFFu;            /* image the thread lists belong to */
  std::vector<ThreadRec*> recs;             /* the lists of the picture being parsed */
  std::vector<ThreadRec*> pool;             /* idle */
  int frame_of_slot[M355_MAX_REF_FRAMES];
  uint32_t dev_id[M355_MAX_REF_FRAMES];     /* image ID whose pixels the slot's device frame holds */
  uint32_t host_id[M355_MAX_REF_FRAMES];    /* image ID whose pixels the host planes hold (downloaded) */
  int geom[M355_MAX_REF_FRAMES][5];
  /* asynchronous outcome of the submits (lists recorded in place are checked on the device, m355_decode_status): the serial of
     the decode that produced each slot's device frame while its outcome is unknown, and whether that frame is damaged (its own
     lists were rejected, or it was predicted from a damaged frame) */
  unsigned long long pending_serial[M355_MAX_REF_FRAMES];
  bool damaged[M355_MAX_REF_FRAMES];
  /* picture output: once the application has asked for a picture's planes, every later output picture's download is started right
     behind its decode (asynchronous, into the image's pinned planes) instead of when the application asks: dl_id = the image whose
     copy is in flight (or has landed) per slot */
  uint32_t dl_id[M355_MAX_REF_FRAMES];
  std::atomic<bool> app_reads{false};
  long long n_prefetched = 0;
  long long n_rejected = 0;
  PlanePool planes;
  /* The submit step (metadata walk, lists into the arena, m355_submit_picture) of a parsed picture runs on a WORKER thread while
     the decoder's own thread goes on to parse the next picture (m355_glue.cc picture_complete): one job per picture, in order.
     api_mu serialises every call into the backend context (worker: submit; application / decoder thread: download, hash, status) */
  struct Job {
    de265_image* img = nullptr; uint32_t id = 0; int dslot = -1;
    std::vector<ThreadRec*> recs;
    const de265_image* dpb_img[M355_MAX_REF_FRAMES]; uint32_t dpb_id[M355_MAX_REF_FRAMES];   /* the DPB as it was when the picture completed */
  };
  std::mutex job_mu, api_mu;
  std::condition_variable job_cv, idle_cv;
  std::deque<Job> jobs;
  bool busy = false, stop = false, sync_submit = false;
  uint32_t busy_id = 0xFFFFFFFFu;
  uint32_t busy_dpb_id[M355_MAX_REF_FRAMES];    /* the DPB snapshot of the job the worker is running */
  /* DE265_DECODER_PARAM_DISABLE_SAO as the application set it.  The decoder's own param_disable_sao is kept TRUE: that is what
     makes decctx.cc:1928 allocate every picture through de265_set_image_allocation_functions (pinned planes, and the release hook
     below that keeps the DPB from recycling an image the worker still reads) — with SAO on, the reference decodes into a private
     image and lets apply_sample_adaptive_offset write the output one; here SAO runs on the device and there is only one. */
  bool user_disable_sao = false;
  std::thread worker;
  std::vector<int> deferred_warnings;       /* raised by the worker, handed to the decoder on its own thread */
  double ms_main = 0;
  /* statistics (m355_glue_stats) */
  long long n_pictures = 0, n_uploads = 0, n_downloads = 0, n_hashed = 0;
  std::vector<std::vector<m355_cu>> w_bcu;   /* scratch of the worker's metadata walk (submit_picture) */
  std::vector<std::vector<m355_tu>> w_btu;
  std::vector<m355_cu> w_cus;
  std::vector<m355_tu> w_tus;
  double ms_walk = 0, ms_submit = 0, ms_download = 0;
  double ms_phase[6] = {0, 0, 0, 0, 0, 0};   /* of ms_walk: headers + CTBs, CU / TU walk, merge + sort, PCM / intra reorder, copies into the arena, final copies */
  /* M355_GLUE_RANKS=N > 1: ONE bitstream across N backend contexts (device r % device count; all on one device where there is
     only one) — the picture's tiles are split over the ranks (m355_shard_owner_of_tile), every rank reconstructs its tiles and the
     library's in-process group (m355_group_*: copies between the contexts, ordered by events) carries the deblocking / SAO halos and
     the finished tiles.  Rank 0 is `mctx` (pictures are downloaded / hashed there); rframe[r - 1][slot] = the other ranks' frames. */
  int n_ranks = 1;
  std::vector<m355_ctx*> rctx;              /* [0] = mctx */
  std::vector<std::vector<int>> rframe;     /* ranks 1.. */
  m355_group* group = nullptr;
  std::vector<std::vector<int>> rhandle;    /* [ring position][rank]: uploaded pictures, cycled (m355_picture_replace) */
  size_t rhandle_next = 0;
  std::string error;
  Glue()
  {
    for (int i = 0; i < M355_MAX_REF_FRAMES; i++) { frame_of_slot[i] = -1; dev_id[i] = host_id[i] = 0xFFFFFFFFu; pending_serial[i] = 0; damaged[i] = false; dl_id[i] = 0xFFFFFFFFu; }
  }
};

std::mutex g_reg_mu;
std::vector<Glue*> g_glues;

std::atomic<unsigned> g_reg_gen(1);         /* bumped whenever a decoder is created or freed: invalidates the per-thread caches */

Glue* glue_of(const decoder_context* d)
{
  static thread_local const decoder_context* t_d = nullptr;
  static thread_local Glue* t_g = nullptr;
  static thread_local unsigned t_gen = 0;
  if (t_d == d && t_g && t_gen == g_reg_gen.load(std::memory_order_acquire)) return t_g;
  std::lock_guard<std::mutex> g(g_reg_mu);
  for (Glue* x : g_glues) if (x->dctx == d) { t_d = d; t_g = x; t_gen = g_reg_gen.load(); return x; }
  return nullptr;
}

void install_traps(acceleration_functions& a);
void wait_submitted(Glue* g, uint32_t id, bool also_as_reference = false);
void flush_warnings(Glue* g);

/* the calling thread's lists for the picture `img` */
ThreadRec* rec_for(de265_image* img)
{
  static thread_local ThreadRec* t_rec = nullptr;
  static thread_local char t_token;          /* its address identifies the calling thread */
  const decoder_context* d = img->decctx;
  /* the cached lists are still this thread's own only if nobody recycled them since (a recycled ThreadRec may carry the
     same decoder and the same picture again — for ANOTHER thread) */
  if (t_rec && t_rec->thread == &t_token && t_rec->owner == d && t_rec->img_id == img->get_ID()) return t_rec;
  Glue* g = glue_of(d);
  if (!g) return nullptr;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->cur_id != img->get_ID()) {
    /* first block of a new picture; lists of a picture that was never completed (dropped by the decoder) are discarded */
    for (ThreadRec* r : g->recs) { r->clear(); g->pool.push_back(r); }
    g->recs.clear();
    g->cur_id = img->get_ID();
    /* de265_set_parameter_int(.., DE265_DECODER_PARAM_ACCELERATION_CODE, ..) refills the table (decctx.cc:239-270): keep
       the traps in place, so that "no CPU pixel work" stays a checked property, whatever the application selects */
    install_traps(g->dctx->acceleration);
  }
  ThreadRec* r;
  if (!g->pool.empty()) { r = g->pool.back(); g->pool.pop_back(); }
  else r = new ThreadRec;           /* never freed before process exit: other threads may still hold the pointer */
  r->owner = d; r->img_id = img->get_ID(); r->thread = &t_token;
  g->recs.push_back(r);
  t_rec = r;
  return r;
}

inline int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* ------------------------------------------------------------------ slot traps -------------------- */
/* Every slot of the decoder's table is replaced by one of these: the reference's pixel functions are not reachable
 * any more (their callers are the replaced functions), so a call here would mean CPU pixel work went unnoticed. */
template <class R, class... A> R trap(A...) { g_cpu_pixel_calls++; return R(); }
template <class R, class... A> void set_trap(R (*&slot)(A...)) { slot = &trap<R, A...>; }

void install_traps(acceleration_functions& a)
{
  set_trap(a.put_weighted_pred_avg_8); set_trap(a.put_unweighted_pred_8); set_trap(a.put_weighted_pred_8); set_trap(a.put_weighted_bipred_8);
  set_trap(a.put_weighted_pred_avg_16); set_trap(a.put_unweighted_pred_16); set_trap(a.put_weighted_pred_16); set_trap(a.put_weighted_bipred_16);
  set_trap(a.put_hevc_epel_8); set_trap(a.put_hevc_epel_h_8); set_trap(a.put_hevc_epel_v_8); set_trap(a.put_hevc_epel_hv_8);
  set_trap(a.put_hevc_epel_16); set_trap(a.put_hevc_epel_h_16); set_trap(a.put_hevc_epel_v_16); set_trap(a.put_hevc_epel_hv_16);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { set_trap(a.put_hevc_qpel_8[i][j]); set_trap(a.put_hevc_qpel_16[i][j]); }
  set_trap(a.transform_bypass); set_trap(a.transform_bypass_rdpcm_v); set_trap(a.transform_bypass_rdpcm_h);
  set_trap(a.transform_skip_8); set_trap(a.transform_skip_rdpcm_v_8); set_trap(a.transform_skip_rdpcm_h_8);
  set_trap(a.transform_4x4_dst_add_8); set_trap(a.transform_skip_16); set_trap(a.transform_4x4_dst_add_16);
  for (int i = 0; i < 4; i++) { set_trap(a.transform_add_8[i]); set_trap(a.transform_add_16[i]); }
  set_trap(a.rotate_coefficients);
  set_trap(a.transform_idst_4x4); set_trap(a.transform_idct_4x4); set_trap(a.transform_idct_8x8); set_trap(a.transform_idct_16x16); set_trap(a.transform_idct_32x32);
  set_trap(a.add_residual_8); set_trap(a.add_residual_16);
  set_trap(a.dequant_coeff_block);
  set_trap(a.deblock_luma_8); set_trap(a.deblock_chroma_8);
  set_trap(a.rdpcm_v); set_trap(a.rdpcm_h); set_trap(a.transform_skip_residual);
  set_trap(a.intra_pred_dc_8); set_trap(a.intra_pred_dc_16); set_trap(a.intra_pred_planar_8); set_trap(a.intra_pred_planar_16);
  set_trap(a.intra_pred_angular_8); set_trap(a.intra_pred_angular_16);
}

/* ------------------------------------------------------------------ image allocation hook --------- */

int glue_get_buffer(de265_decoder_context* ctx, de265_image_spec* spec, de265_image* img, void* userdata)
{
  /* same plane geometry as the default allocator (image.cc:110-160), from the pinned pool, NOT zero-filled: the host
     planes are only ever written by PCM sample reads, reference concealment (fill_image) and picture downloads */
  Glue* g = (Glue*)userdata;
  const int subw = (img->get_chroma_format() == de265_chroma_420 || img->get_chroma_format() == de265_chroma_422) ? 2 : 1;
  const int subh = img->get_chroma_format() == de265_chroma_420 ? 2 : 1;
  const uint32_t cw = spec->width / subw, ch = spec->height / subh;
  const uint32_t ls = (spec->width + spec->alignment - 1) / spec->alignment * spec->alignment;
  uint32_t cs = (cw + spec->alignment - 1) / spec->alignment * spec->alignment;
  const size_t lb = (size_t)ls * ((img->get_bit_depth(0) + 7) / 8) * spec->height + 64;
  const size_t cb = (size_t)cs * ((img->get_bit_depth(1) + 7) / 8) * ch + 64;
  void* p[3] = {g->planes.get(lb), nullptr, nullptr};
  bool ok = p[0] != nullptr;
  if (img->get_chroma_format() != de265_chroma_mono) {
    p[1] = g->planes.get(cb); p[2] = g->planes.get(cb);
    ok = ok && p[1] && p[2];
  } else cs = 0;
  if (!ok) { for (void* q : p) if (q) g->planes.put(q); return 0; }
  /* de265_set_image_plane takes the stride in BYTES (de265.cc:747-752 divides by the bytes per sample) */
  de265_set_image_plane(img, 0, p[0], (int)(ls * ((img->get_bit_depth(0) + 7) / 8)), nullptr);
  de265_set_image_plane(img, 1, p[1], (int)(cs * ((img->get_bit_depth(1) + 7) / 8)), nullptr);
  de265_set_image_plane(img, 2, p[2], (int)(cs * ((img->get_bit_depth(1) + 7) / 8)), nullptr);
  (void)ctx;
  return 1;
}
void glue_release_buffer(de265_decoder_context* ctx, de265_image* img, void* userdata)
{
  Glue* g = (Glue*)userdata;
  wait_submitted(g, img->get_ID(), true);         /* the DPB recycles the image (dpb.cc:206-216): the worker may still be walking its metadata,
                                                     or reading it as a reference of a later picture's job */
  {
    /* a download started behind the picture's decode that nobody waited for (the application never looked at the picture): it must
       have landed before the planes go back to the pool — the next image's parser may write PCM samples into them */
    std::lock_guard<std::mutex> api_lock(g->api_mu);
    for (int s = 0; s < M355_MAX_REF_FRAMES; s++)
      if (g->dl_id[s] == img->get_ID() && g->host_id[s] != img->get_ID() && g->frame_of_slot[s] >= 0) {
        api()->m355_frame_download_wait(g->mctx, g->frame_of_slot[s]);
        g->dl_id[s] = 0xFFFFFFFFu;
      }
  }
  for (int c = 0; c < 3; c++) {
    void* p = (void*)img->get_image_plane(c);
    if (p) g->planes.put(p);
  }
  (void)ctx;
}

/* ------------------------------------------------------------------ device frames ---------------- */

int slot_of(decoder_context* d, const de265_image* img)
{
  for (int i = 0; i < M355_MAX_REF_FRAMES && d->has_image(i); i++)
    if (d->get_image(i) == img) return i;
  return -1;
}

bool ensure_frame(Glue* g, int slot, const de265_image* img)
{
  const seq_parameter_set& sps = img->get_sps();
  const int geo[5] = {sps.pic_width_in_luma_samples, sps.pic_height_in_luma_samples, sps.chroma_format_idc, sps.BitDepth_Y, sps.BitDepth_C};
  if (g->frame_of_slot[slot] >= 0 && memcmp(geo, g->geom[slot], sizeof(geo)) == 0) return true;
  if (g->frame_of_slot[slot] >= 0) api()->m355_frame_destroy(g->mctx, g->frame_of_slot[slot]);
  g->frame_of_slot[slot] = api()->m355_frame_create(g->mctx, geo[0], geo[1], geo[2], geo[3], geo[4]);
  g->dev_id[slot] = 0xFFFFFFFFu;
  if (g->frame_of_slot[slot] < 0) { g->error = api()->m355_last_error(); return false; }
  for (int r = 1; r < g->n_ranks; r++) {
    int& f = g->rframe[(size_t)r - 1][(size_t)slot];
    if (f >= 0) api()->m355_frame_destroy(g->rctx[(size_t)r], f);
    f = api()->m355_frame_create(g->rctx[(size_t)r], geo[0], geo[1], geo[2], geo[3], geo[4]);
    if (f < 0) { g->error = api()->m355_last_error(); return false; }
  }
  memcpy(g->geom[slot], geo, sizeof(geo));
  return true;
}

/* a DPB picture the backend did not produce (the decoder's concealment pictures, decctx.cc:1294-1321): host -> device */
bool upload_host_planes(Glue* g, int slot, const de265_image* img)
{
  if (!ensure_frame(g, slot, img)) return false;
  const int nc = img->get_chroma_format() == de265_chroma_mono ? 1 : 3;
  for (int c = 0; c < nc; c++)
    for (int r = 0; r < g->n_ranks; r++)
      if (api()->m355_frame_upload(g->rctx[(size_t)r], r ? g->rframe[(size_t)r - 1][(size_t)slot] : g->frame_of_slot[slot], c, img->get_image_plane(c), img->get_image_stride(c)) != M355_OK) { g->error = api()->m355_last_error(); return false; }
  g->dev_id[slot] = img->get_ID();
  g->host_id[slot] = img->get_ID();
  g->n_uploads++;
  return true;
}

/* Collect the outcome of the decodes whose status is still open (non-blocking unless `wait_for_slot` >= 0: then that slot's
 * decode is waited for).  A picture whose lists the device rejected was not decoded: it is marked as the reference marks a
 * picture with decoding errors (image.h:347 integrity; decctx.cc checks it when the picture is output), pictures predicted
 * from it are marked when they are submitted. */
void collect_status(Glue* g, int wait_for_slot, const Glue::Job* job = nullptr)
{
  for (int s = 0; s < M355_MAX_REF_FRAMES; s++) {
    if (!g->pending_serial[s]) continue;
    int st = api()->m355_decode_status(g->mctx, g->pending_serial[s]);
    while (st == M355_ERR_BUSY && s == wait_for_slot) { std::this_thread::yield(); st = api()->m355_decode_status(g->mctx, g->pending_serial[s]); }
    if (st == M355_ERR_BUSY) continue;
    if (st == M355_ERR_STALE) st = M355_OK;   /* (older than the backend's status ring: a rejection would surface in m355_wait) */
    if (st != M355_OK) {
      g->error = api()->m355_last_error();
      fprintf(stderr, "libde265 (MI355X glue): %s\n", g->error.c_str());
      g->damaged[s] = true;
      g->n_rejected++;
      /* (on the worker the DPB is looked at through the job's snapshot: the decoder's thread may be reshaping it) */
      de265_image* im = job ? const_cast<de265_image*>(job->dpb_img[s]) : (g->dctx->has_image(s) ? g->dctx->get_image(s) : nullptr);
      if (im && (job ? job->dpb_id[s] : im->get_ID()) == g->dev_id[s]) im->integrity = INTEGRITY_DECODING_ERRORS;
      { std::lock_guard<std::mutex> lk(g->job_mu); g->deferred_warnings.push_back(DE265_WARNING_INCORRECT_ENTRY_POINT_OFFSET); }   /* there is no backend-specific warning code in de265.h */
    }
    g->pending_serial[s] = 0;
  }
}

void download_if_needed(Glue* g, de265_image* img)
{
  wait_submitted(g, img->get_ID());
  flush_warnings(g);
  std::lock_guard<std::mutex> api_lock(g->api_mu);
  const int slot = slot_of(g->dctx, img);
  if (slot < 0 || g->frame_of_slot[slot] < 0) return;
  if (g->dev_id[slot] != img->get_ID() || g->host_id[slot] == img->get_ID()) return;
  const auto t0 = std::chrono::steady_clock::now();
  collect_status(g, slot);                         /* this picture's own outcome (waits for its decode) */
  if (g->damaged[slot] && img->integrity == INTEGRITY_CORRECT) img->integrity = INTEGRITY_DERIVED_FROM_FAULTY_REFERENCE;
  g->app_reads.store(true);
  if (g->dl_id[slot] == img->get_ID()) {           /* its copy was started behind its decode: wait for that copy only */
    if (api()->m355_frame_download_wait(g->mctx, g->frame_of_slot[slot]) == M355_OK) {
      g->host_id[slot] = img->get_ID();
      g->n_downloads++;
      g->ms_download += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      return;
    }
  }
  const int nc = img->get_chroma_format() == de265_chroma_mono ? 1 : 3;
  for (int c = 0; c < nc; c++)
    if (api()->m355_frame_download(g->mctx, g->frame_of_slot[slot], c, img->get_image_plane(c), img->get_image_stride(c)) != M355_OK) {
      g->error = api()->m355_last_error();
      fprintf(stderr, "libde265 (MI355X glue): download failed: %s\n", g->error.c_str());
      return;
    }
  g->host_id[slot] = img->get_ID();
  g->n_downloads++;
  g->ms_download += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

/* ------------------------------------------------------------------ picture submission ----------- */

void walk_tu(const de265_image* img, int x0, int y0, int log2, int depth, std::vector<m355_tu>& out)
{
  /* the recursion of markTransformBlockBoundary (deblock.cc:33-63) */
  if (img->get_split_transform_flag(x0, y0, depth)) {
    const int h = 1 << (log2 - 1);
    walk_tu(img, x0, y0, log2 - 1, depth + 1, out);
    walk_tu(img, x0 + h, y0, log2 - 1, depth + 1, out);
    walk_tu(img, x0, y0 + h, log2 - 1, depth + 1, out);
    walk_tu(img, x0 + h, y0 + h, log2 - 1, depth + 1, out);
  } else {
    m355_tu tu; memset(&tu, 0, sizeof(tu));
    tu.x = (uint16_t)x0; tu.y = (uint16_t)y0; tu.log2_size = (uint8_t)log2;
    tu.flags = img->get_nonzero_coefficient(x0, y0) ? M355_TUF_NONZERO_COEFF : 0;
    out.push_back(tu);
  }
}

/* z-scan rank of a luma position inside its CTB at 8x8 granularity: the decode order of coding units */
inline uint32_t z_rank(int x, int y)
{
  uint32_t r = 0;
  for (int b = 0; b < 4; b++) r |= (uint32_t)(((x >> (3 + b)) & 1) << (2 * b)) | (uint32_t)(((y >> (3 + b)) & 1) << (2 * b + 1));
  return r;
}

/* the submit step's own data-parallel work (metadata walk, copies into the arena): a handful of short-lived threads per picture */
int glue_threads()
{
  static int n = 0;
  if (!n) {
    const unsigned hc = std::thread::hardware_concurrency();
    n = hc >= 32 ? 16 : (hc >= 8 ? 8 : (hc >= 4 ? 4 : 1));
    if (const char* e = getenv("M355_GLUE_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = v; }
  }
  return n;
}
/* persistent helper threads: a thread created per parallel phase costs this process 0.3-0.5 ms each while the parser's threads are
   busy (stack mmap / clone against their page faults) — sixteen of them made the 8K metadata walk take 10 ms against 4 ms on one
   thread (profiles/r03_z11_*) */
struct TaskPool {
  std::vector<std::thread> th;
  std::mutex mu, run_mu;
  std::condition_variable cv_go, cv_done;
  const std::function<void(size_t)>* job = nullptr;
  size_t n_tasks = 0;
  std::atomic<size_t> next{0};
  size_t done = 0;
  int active = 0;                                              /* workers that hold the current job's function */
  unsigned long long gen = 0;
  bool stop = false;
  explicit TaskPool(int workers) { for (int i = 0; i < workers; i++) th.emplace_back([this]() { work(); }); }
  ~TaskPool() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv_go.notify_all(); for (auto& t : th) t.join(); }
  void drain(const std::function<void(size_t)>& f, size_t n)
  {
    size_t mine = 0;
    for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; f(i); mine++; }
    if (mine) { std::lock_guard<std::mutex> g(mu); done += mine; if (done >= n) cv_done.notify_all(); }
  }
  void work()
  {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(size_t)>* f; size_t n;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&]() { return stop || gen != seen; });
        if (stop) return;
        seen = gen; f = job; n = n_tasks;
        if (f) active++;
      }
      if (f) {
        drain(*f, n);
        std::lock_guard<std::mutex> g(mu);
        if (--active == 0) cv_done.notify_all();
      }
    }
  }
  void run(size_t n, const std::function<void(size_t)>& f)
  {
    std::lock_guard<std::mutex> one(run_mu);                   /* one parallel phase at a time */
    { std::lock_guard<std::mutex> g(mu); job = &f; n_tasks = n; next.store(0); done = 0; gen++; }
    cv_go.notify_all();
    drain(f, n);
    std::unique_lock<std::mutex> lk(mu);
    job = nullptr;                                             /* a worker that wakes up from here on takes nothing ... */
    cv_done.wait(lk, [&]() { return done >= n && active == 0; });   /* ... and none still holds `f` when this returns */
  }
};
void parallel_tasks(size_t n_tasks, const std::function<void(size_t)>& f)
{
  const int T = (int)std::min<size_t>((size_t)glue_threads(), n_tasks);
  if (T <= 1) { for (size_t i = 0; i < n_tasks; i++) f(i); return; }
  static TaskPool* pool = new TaskPool(glue_threads() - 1);    /* lives until process exit */
  pool->run(n_tasks, f);
}

/* One picture over the ranks of the glue's group.  `pic` = the whole picture's lists in host memory (copying mode).  Rank r gets
 * the coding units, transform leaves, prediction blocks, residual and intra blocks of ITS tiles (the owner of a block is the owner
 * of its CTB's tile); slices, CTB table (SAO parameters, slice indices), weights, PCM samples, coefficients and the residual
 * numbering stay picture-wide — what libde265_amd/shard.py shard_picture does for the Python harness. */
int submit_sharded(Glue* g, const m355_picture& pic, int dslot)
{
  Api* A = api();
  const m355_pic_params& pp = pic.pp;
  const int N = g->n_ranks, l2 = pp.log2_ctb_size, cs = 1 << l2;
  const int ctbW = (pp.width + cs - 1) >> l2, ctbH = (pp.height + cs - 1) >> l2, n_tiles = pp.num_tile_cols * pp.num_tile_rows;
  std::vector<uint8_t> owner((size_t)ctbW * ctbH, 0);
  for (int ty = 0, t = 0; ty < pp.num_tile_rows; ty++)
    for (int tx = 0; tx < pp.num_tile_cols; tx++, t++) {
      const uint8_t o = (uint8_t)A->m355_shard_owner_of_tile(t, n_tiles, N);
      for (int y = pp.row_bd[ty]; y < pp.row_bd[ty + 1] && y < ctbH; y++)
        for (int x = pp.col_bd[tx]; x < pp.col_bd[tx + 1] && x < ctbW; x++) owner[(size_t)y * ctbW + x] = o;
    }
  const int sw = (pp.chroma_format_idc == 1 || pp.chroma_format_idc == 2) ? 2 : 1, sh = pp.chroma_format_idc == 1 ? 2 : 1;
  auto own = [&](int x, int y) { return owner[(size_t)(y >> l2) * ctbW + (x >> l2)]; };
  if (g->rhandle.size() < 4) g->rhandle.resize(4);
  std::vector<int>& hs = g->rhandle[g->rhandle_next];
  g->rhandle_next = (g->rhandle_next + 1) % g->rhandle.size();
  const bool fresh = hs.empty();
  if (fresh) hs.assign((size_t)N, -1);
  /* The ranks' lists are cut out side by side, IN PLACE: a counting pass per rank, then m355_picture_arena_begin hands out room in
     the pinned arena of the rank's handle (with the border units of the other ranks behind cus[] / pbs[], sized from the picture
     parameters) and the cut writes there — m355_picture_replace then copies nothing on the host (what the per-rank vectors and the
     library's staging copy cost before: one more pass over every list).  Each task reads the whole picture's lists and keeps its
     rank's records; the coefficients of the kept residual blocks are compacted.  M355_GLUE_RANKS_COPY=1: the vectors again. */
  static const bool ranks_copy = getenv("M355_GLUE_RANKS_COPY") != nullptr;
  struct RankLists { std::vector<m355_cu> cus; std::vector<m355_tu> tus; std::vector<m355_pb> pbs; std::vector<m355_rb> rbs; std::vector<m355_ib> ibs;
                     std::vector<m355_ctb> ctbs; std::vector<uint32_t> coeffs; int rb_count[4]; };
  static std::vector<RankLists> L;                          /* (kept between pictures: one worker thread submits) */
  if ((int)L.size() < N) L.resize((size_t)N);
  std::vector<uint32_t> order((size_t)pic.n_ctbs);
  for (int i = 0; i < pic.n_ctbs; i++) order[(size_t)i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pic.ctbs[a].ib_start < pic.ctbs[b].ib_start; });
  auto own_rb = [&](const m355_rb& rb) { return own(rb.cidx ? rb.x * sw : rb.x, rb.cidx ? rb.y * sh : rb.y); };
  if (!ranks_copy) {
    struct Cnt { int cus = 0, tus = 0, pbs = 0, rbs[4] = {0, 0, 0, 0}, ibs = 0; uint32_t coeffs = 0; };
    std::vector<Cnt> cnt((size_t)N);
    parallel_tasks((size_t)N, [&](size_t rr) {
      const int r = (int)rr; Cnt& c = cnt[rr];
      for (int i = 0; i < pic.n_cus; i++) c.cus += own(pic.cus[i].x, pic.cus[i].y) == r;
      for (int i = 0; i < pic.n_tus; i++) c.tus += own(pic.tus[i].x, pic.tus[i].y) == r;
      for (int i = 0; i < pic.n_pbs; i++) c.pbs += own(pic.pbs[i].x, pic.pbs[i].y) == r;
      const m355_rb* src = pic.rbs;
      for (int s = 0; s < 4; s++)
        for (int i = 0; i < pic.rb_count[s]; i++, src++) if (own_rb(*src) == r) { c.rbs[s]++; c.coeffs += src->ncoeff; }
      for (int ci = 0; ci < pic.n_ctbs; ci++) if (owner[(size_t)ci] == r) c.ibs += (int)pic.ctbs[ci].ib_count;
    });
    std::vector<m355_picture> rp((size_t)N);
    std::vector<m355_arena_caps> caps((size_t)N);
    for (int r = 0; r < N; r++) {                           /* (one context after the other: the arenas may have to grow) */
      m355_arena_caps& k = caps[(size_t)r]; memset(&k, 0, sizeof(k));
      const Cnt& c = cnt[(size_t)r];
      k.n_slices = pic.n_slices; k.n_ctbs = pic.n_ctbs; k.n_cus = c.cus + 1; k.n_tus = c.tus + 1; k.n_pbs = c.pbs + 1; k.n_wts = pic.n_wts + 1; k.n_ibs = c.ibs + 1;
      for (int s = 0; s < 4; s++) k.n_rbs[s] = c.rbs[s] + 1;
      k.n_coeffs = c.coeffs + 1; k.n_pcm = pic.n_pcm + 1; k.scaling = pic.scaling_factors ? 1 : 0;
      const int h = A->m355_picture_arena_begin(g->rctx[(size_t)r], fresh ? -1 : hs[(size_t)r], &k, &pic.pp, &rp[(size_t)r]);
      if (h < 0) return -h;
      hs[(size_t)r] = h;
    }
    parallel_tasks((size_t)N, [&](size_t rr) {
      const int r = (int)rr;
      m355_picture& q = rp[rr];
      const m355_arena_caps& k = caps[rr];
      /* picture-wide lists: slices, the CTB table (renumbered below), weights, PCM samples, scaling factors */
      memcpy((void*)q.slices, pic.slices, sizeof(m355_slice) * (size_t)pic.n_slices);
      if (pic.n_wts) memcpy((void*)q.wts, pic.wts, sizeof(m355_wt) * (size_t)pic.n_wts);
      if (pic.n_pcm) memcpy((void*)q.pcm, pic.pcm, 2 * (size_t)pic.n_pcm);
      if (pic.scaling_factors) memcpy((void*)q.scaling_factors, pic.scaling_factors, 6 * (16 + 64 + 256 + 1024));
      m355_cu* cu = (m355_cu*)q.cus; m355_tu* tu = (m355_tu*)q.tus; m355_pb* pb = (m355_pb*)q.pbs; uint32_t* co = (uint32_t*)q.coeffs;
      int n = 0;
      for (int i = 0; i < pic.n_cus; i++) if (own(pic.cus[i].x, pic.cus[i].y) == r) cu[n++] = pic.cus[i];
      q.n_cus = n; n = 0;
      for (int i = 0; i < pic.n_tus; i++) if (own(pic.tus[i].x, pic.tus[i].y) == r) tu[n++] = pic.tus[i];
      q.n_tus = n; n = 0;
      for (int i = 0; i < pic.n_pbs; i++) if (own(pic.pbs[i].x, pic.pbs[i].y) == r) pb[n++] = pic.pbs[i];
      q.n_pbs = n;
      const m355_rb* src = pic.rbs;
      uint32_t nco = 0;
      for (int s = 0; s < 4; s++) {
        m355_rb* dst = k.rb_bin[s];
        int kept = 0;
        for (int i = 0; i < pic.rb_count[s]; i++, src++)
          if (own_rb(*src) == r) {
            m355_rb rb = *src;
            rb.coeff_ofs = nco;
            memcpy(co + nco, pic.coeffs + src->coeff_ofs, 4 * (size_t)src->ncoeff);
            nco += src->ncoeff;
            dst[kept++] = rb;
          }
        q.rb_count[s] = kept;
      }
      q.n_coeffs = nco;
      /* intra blocks: whole CTBs are kept or dropped; the kept CTBs' runs are renumbered (ascending ib_start = the order they lie in) */
      m355_ctb* ct = (m355_ctb*)q.ctbs; m355_ib* ib = (m355_ib*)q.ibs;
      memcpy(ct, pic.ctbs, sizeof(m355_ctb) * (size_t)pic.n_ctbs);
      uint32_t nib = 0;
      for (uint32_t ci : order) {
        m355_ctb& c = ct[ci];
        if (owner[ci] != r || !c.ib_count) { c.ib_start = 0; c.ib_count = 0; continue; }
        memcpy(ib + nib, pic.ibs + c.ib_start, sizeof(m355_ib) * (size_t)c.ib_count);
        c.ib_start = nib; nib += c.ib_count;
      }
      q.n_ibs = (int32_t)nib;
      q.pp = pic.pp; q.n_slices = pic.n_slices; q.n_ctbs = pic.n_ctbs; q.n_wts = pic.n_wts; q.n_pcm = pic.n_pcm; q.res_len = pic.res_len;
      if (!pic.scaling_factors) q.scaling_factors = nullptr;
      q.dst_frame = pic.dst_frame;
      for (int i = 0; i < M355_MAX_REF_FRAMES; i++) q.ref_frames[i] = pic.ref_frames[i];
      if (r) {
        q.dst_frame = g->rframe[(size_t)r - 1][(size_t)dslot];
        for (int i = 0; i < M355_MAX_REF_FRAMES; i++) if (pic.ref_frames[i] >= 0) q.ref_frames[i] = g->rframe[(size_t)r - 1][(size_t)i];
      }
    });
    for (int r = 0; r < N; r++) {
      const int rc = A->m355_picture_replace(g->rctx[(size_t)r], hs[(size_t)r], &rp[(size_t)r]);
      if (rc != M355_OK) return rc;
    }
    return A->m355_group_decode(g->group, hs.data(), 1);    /* (every rank ends up with the whole picture: it is a reference everywhere) */
  }
  parallel_tasks((size_t)N, [&](size_t rr) {
    const int r = (int)rr;
    RankLists& l = L[rr];
    l.cus.clear(); l.tus.clear(); l.pbs.clear(); l.rbs.clear(); l.ibs.clear(); l.coeffs.clear();
    for (int i = 0; i < pic.n_cus; i++) if (own(pic.cus[i].x, pic.cus[i].y) == r) l.cus.push_back(pic.cus[i]);
    for (int i = 0; i < pic.n_tus; i++) if (own(pic.tus[i].x, pic.tus[i].y) == r) l.tus.push_back(pic.tus[i]);
    for (int i = 0; i < pic.n_pbs; i++) if (own(pic.pbs[i].x, pic.pbs[i].y) == r) l.pbs.push_back(pic.pbs[i]);
    const m355_rb* src = pic.rbs;
    for (int s = 0; s < 4; s++) {
      int kept = 0;
      for (int i = 0; i < pic.rb_count[s]; i++, src++)
        if (own_rb(*src) == r) {
          m355_rb rb = *src;
          rb.coeff_ofs = (uint32_t)l.coeffs.size();
          l.coeffs.insert(l.coeffs.end(), pic.coeffs + src->coeff_ofs, pic.coeffs + src->coeff_ofs + src->ncoeff);
          l.rbs.push_back(rb); kept++;
        }
      l.rb_count[s] = kept;
    }
    /* intra blocks: whole CTBs are kept or dropped; the kept CTBs' runs are renumbered (ascending ib_start = the order they lie in) */
    l.ctbs.assign(pic.ctbs, pic.ctbs + pic.n_ctbs);
    for (uint32_t ci : order) {
      m355_ctb& c = l.ctbs[ci];
      if (owner[ci] != r || !c.ib_count) { c.ib_start = 0; c.ib_count = 0; continue; }
      const uint32_t start = (uint32_t)l.ibs.size();
      l.ibs.insert(l.ibs.end(), pic.ibs + c.ib_start, pic.ibs + c.ib_start + c.ib_count);
      c.ib_start = start;
    }
  });
  for (int r = 0; r < N; r++) {
    RankLists& l = L[(size_t)r];
    m355_picture rp = pic;
    for (int s = 0; s < 4; s++) rp.rb_count[s] = l.rb_count[s];
    rp.cus = l.cus.data(); rp.n_cus = (int32_t)l.cus.size(); rp.tus = l.tus.data(); rp.n_tus = (int32_t)l.tus.size();
    rp.pbs = l.pbs.data(); rp.n_pbs = (int32_t)l.pbs.size(); rp.rbs = l.rbs.data(); rp.ibs = l.ibs.data(); rp.n_ibs = (int32_t)l.ibs.size();
    rp.ctbs = l.ctbs.data(); rp.coeffs = l.coeffs.data(); rp.n_coeffs = (uint32_t)l.coeffs.size();
    if (r) {
      rp.dst_frame = g->rframe[(size_t)r - 1][(size_t)dslot];
      for (int i = 0; i < M355_MAX_REF_FRAMES; i++) if (pic.ref_frames[i] >= 0) rp.ref_frames[i] = g->rframe[(size_t)r - 1][(size_t)i];
    }
    const int h = fresh ? A->m355_picture_upload(g->rctx[(size_t)r], &rp) : A->m355_picture_replace(g->rctx[(size_t)r], hs[(size_t)r], &rp);
    if (fresh) { if (h < 0) return -h; hs[(size_t)r] = h; }
    else if (h != M355_OK) return h;
  }
  return A->m355_group_decode(g->group, hs.data(), 1);      /* (every rank ends up with the whole picture: it is a reference everywhere) */
}

bool submit_picture(Glue* g, Glue::Job& job)
{
  de265_image* img = job.img;
  Api* A = api();
  decoder_context* d = g->dctx;
  /* (the release hook keeps the DPB from recycling an image with a job in the queue: glue_release_buffer) */
  if (img->get_ID() != job.id) { g->error = "the picture's image was recycled before its submit step ran"; return false; }
  const auto t0 = std::chrono::steady_clock::now();
  auto tp = t0;
  auto lap = [&](int k) { const auto n = std::chrono::steady_clock::now(); g->ms_phase[k] += std::chrono::duration<double, std::milli>(n - tp).count(); tp = n; };
  const seq_parameter_set& sps = img->get_sps();
  const pic_parameter_set& pps = img->get_pps();

  m355_picture pic; memset(&pic, 0, sizeof(pic));
  m355_pic_params& pp = pic.pp;
  pp.width = sps.pic_width_in_luma_samples; pp.height = sps.pic_height_in_luma_samples;
  pp.chroma_format_idc = (uint8_t)sps.chroma_format_idc;
  pp.bit_depth_luma = (uint8_t)sps.BitDepth_Y; pp.bit_depth_chroma = (uint8_t)sps.BitDepth_C;
  pp.log2_ctb_size = (uint8_t)sps.Log2CtbSizeY; pp.log2_min_tb_size = (uint8_t)sps.Log2MinTrafoSize;
  pp.log2_min_cb_size = (uint8_t)sps.Log2MinCbSizeY;
  pp.pic_cb_qp_offset = (int8_t)pps.pic_cb_qp_offset; pp.pic_cr_qp_offset = (int8_t)pps.pic_cr_qp_offset;
  if (pps.constrained_intra_pred_flag) pp.flags |= M355_PF_CONSTRAINED_INTRA_PRED;
  if (sps.strong_intra_smoothing_enable_flag) pp.flags |= M355_PF_STRONG_INTRA_SMOOTHING;
  if (sps.pcm_loop_filter_disable_flag) pp.flags |= M355_PF_PCM_LOOP_FILTER_DISABLE;
  if (pps.loop_filter_across_tiles_enabled_flag) pp.flags |= M355_PF_LF_ACROSS_TILES;
  if (sps.sample_adaptive_offset_enabled_flag && !g->user_disable_sao) pp.flags |= M355_PF_SAO_ENABLED;
  if (sps.range_extension.intra_smoothing_disabled_flag) pp.flags |= M355_PF_INTRA_SMOOTHING_DISABLED;
  if (sps.range_extension.implicit_rdpcm_enabled_flag) pp.flags |= M355_PF_IMPLICIT_RDPCM;
  if (sps.range_extension.transform_skip_rotation_enabled_flag) pp.flags |= M355_PF_TRANSFORM_SKIP_ROTATION;
  if (sps.scaling_list_enable_flag) pp.flags |= M355_PF_SCALING_LIST;
  if (!d->param_disable_deblocking) pp.flags |= M355_PF_DEBLOCK_ENABLED;
  if (pps.range_extension.cross_component_prediction_enabled_flag) pp.flags |= M355_PF_CROSS_COMPONENT_PRED;
  pp.num_tile_cols = (uint8_t)pps.num_tile_columns; pp.num_tile_rows = (uint8_t)pps.num_tile_rows;
  if (pps.num_tile_columns > M355_MAX_TILE_COLS || pps.num_tile_rows > M355_MAX_TILE_ROWS) { g->error = "tile grid larger than the backend's tables"; return false; }
  for (int i = 0; i <= pps.num_tile_columns; i++) pp.col_bd[i] = (uint16_t)pps.colBd[i];
  for (int i = 0; i <= pps.num_tile_rows; i++) pp.row_bd[i] = (uint16_t)pps.rowBd[i];

  /* ---- slice headers (what deblocking / SAO read of them) ---- */
  std::vector<m355_slice> slices;
  for (size_t i = 0; i < img->slices.size(); i++) {
    const slice_segment_header* sh = img->slices[i];
    m355_slice s; memset(&s, 0, sizeof(s));
    s.slice_addr_rs = (int32_t)sh->SliceAddrRS;
    s.beta_offset = (int8_t)sh->slice_beta_offset; s.tc_offset = (int8_t)sh->slice_tc_offset;
    if (sh->slice_deblocking_filter_disabled_flag) s.flags |= M355_SF_DEBLOCK_DISABLED;
    if (sh->slice_loop_filter_across_slices_enabled_flag) s.flags |= M355_SF_LF_ACROSS_SLICES;
    if (sh->slice_sao_luma_flag) s.flags |= M355_SF_SAO_LUMA;
    if (sh->slice_sao_chroma_flag) s.flags |= M355_SF_SAO_CHROMA;
    slices.push_back(s);
  }
  if (slices.empty()) { g->error = "picture without slices"; return false; }

  /* ---- CTBs: slice index, SAO parameters (image.h:160-170, slice.h:268-276) ---- */
  const int ctbW = sps.PicWidthInCtbsY, ctbH = sps.PicHeightInCtbsY;
  std::vector<m355_ctb> ctbs((size_t)ctbW * ctbH);
  bool any_pcm = false;
  for (int y = 0; y < ctbH; y++)
    for (int x = 0; x < ctbW; x++) {
      m355_ctb& c = ctbs[(size_t)y * ctbW + x]; memset(&c, 0, sizeof(c));
      const unsigned si = img->get_SliceHeaderIndexCtb(x, y);
      c.slice_idx = (uint16_t)(si < slices.size() ? si : 0);
      const sao_info* sao = img->get_sao_info(x, y);
      c.sao_type = sao->SaoTypeIdx; c.sao_eo_class = sao->SaoEoClass;
      memcpy(c.sao_band_pos, sao->sao_band_position, 3);
      memcpy(c.sao_offset, sao->saoOffsetVal, 12);
      if (img->get_CTB_has_pcm_or_cu_transquant_bypass(x, y)) c.flags |= M355_CTBF_HAS_PCM_OR_BYPASS;
    }

  lap(0);
  /* ---- coding units + transform-tree leaves (image.h:173-195, deblock.cc:33-63) ---- */
  /* (the walk's vectors live in the Glue and keep their capacity from picture to picture: growing sixty vectors on sixteen
     threads at once made the parallel walk slower than a serial one — 10 ms against 5 ms per 8K picture, the allocator and the
     kernel's page-table lock serialise it —, profiles/r03_z11_*) */
  std::vector<m355_cu>& cus = g->w_cus; cus.clear();
  std::vector<m355_tu>& tus = g->w_tus; tus.clear();
  std::vector<uint32_t> pcm_cus;
  const int minCb = sps.MinCbSizeY;
  long long covered = 0;
  {
    /* bands of min-CB rows, walked in parallel, concatenated in raster order */
    const int rows = sps.PicHeightInMinCbsY, per = std::max(1, (rows + 4 * glue_threads() - 1) / (4 * glue_threads()));
    const size_t nb = (size_t)((rows + per - 1) / per);
    std::vector<std::vector<m355_cu>>& bcu = g->w_bcu;
    std::vector<std::vector<m355_tu>>& btu = g->w_btu;
    if (bcu.size() < nb) { bcu.resize(nb); btu.resize(nb); }
    std::vector<long long> bcov(nb, 0);
    parallel_tasks(nb, [&](size_t b) {
      std::vector<m355_cu>& vc = bcu[b]; vc.clear();
      std::vector<m355_tu>& vt = btu[b]; vt.clear();
      long long cov = 0;
      for (int cy = (int)b * per; cy < rows && cy < ((int)b + 1) * per; cy++)
        for (int cx = 0; cx < sps.PicWidthInMinCbsY; cx++) {
          const int l2 = img->get_log2CbSize_cbUnits(cx, cy);
          if (l2 == 0) continue;
          const int x0 = cx * minCb, y0 = cy * minCb;
          m355_cu cu; memset(&cu, 0, sizeof(cu));
          cu.x = (uint16_t)x0; cu.y = (uint16_t)y0; cu.log2_size = (uint8_t)l2;
          cu.pred_mode = (uint8_t)img->get_pred_mode(x0, y0);
          cu.part_mode = (uint8_t)img->get_PartMode(x0, y0);
          cu.qp_y = (int8_t)img->get_QPY(x0, y0);
          if (img->get_pcm_flag(x0, y0)) cu.flags |= M355_CUF_PCM;
          if (img->get_cu_transquant_bypass(x0, y0)) cu.flags |= M355_CUF_TRANSQUANT_BYPASS;
          vc.push_back(cu);
          cov += 1ll << (2 * l2);
          walk_tu(img, x0, y0, l2, 0, vt);
        }
      bcov[b] = cov;
    });
    size_t ncu = 0, ntu = 0;
    for (size_t b = 0; b < nb; b++) { ncu += bcu[b].size(); ntu += btu[b].size(); covered += bcov[b]; }
    cus.reserve(ncu); tus.reserve(ntu);
    for (size_t b = 0; b < nb; b++) { cus.insert(cus.end(), bcu[b].begin(), bcu[b].end()); tus.insert(tus.end(), btu[b].begin(), btu[b].end()); }
    for (size_t i = 0; i < cus.size(); i++) if (cus[i].flags & M355_CUF_PCM) { pcm_cus.push_back((uint32_t)i); any_pcm = true; }
  }

  lap(1);
  /* ---- concatenate the threads' lists ---- */
  const std::vector<ThreadRec*>& recs = job.recs;          /* (empty: a picture without a single coded block) */
  size_t n_pb = 0, n_ib = 0, n_co = 0, n_rb[4] = {0, 0, 0, 0};
  int skipped = 0;
  std::vector<uint32_t> co_base(recs.size()), res_base(recs.size());
  uint32_t res_len = 0;
  for (size_t t = 0; t < recs.size(); t++) {
    co_base[t] = (uint32_t)n_co; res_base[t] = res_len;
    n_pb += recs[t]->pbs.size(); n_ib += recs[t]->ibs.size(); n_co += recs[t]->coeffs.size(); res_len += recs[t]->res_len;
    for (int s = 0; s < 4; s++) n_rb[s] += recs[t]->rbs[s].size();
    skipped += recs[t]->skipped_pbs;
  }
  for (int s = 0; s < 4; s++) pic.rb_count[s] = (int32_t)n_rb[s];
  bool any_weighted = false;
  for (ThreadRec* r : recs) any_weighted = any_weighted || r->any_weighted;
  for (ThreadRec* r : recs) for (int k = 0; k < M355_GLUE_N_FEATURES; k++) if (r->feat[k]) g_feat[k] += r->feat[k];
  g_feat[FEAT_PCM_CU] += (long long)pcm_cus.size();
  if (slices.size() > 1) g_feat[FEAT_MULTI_SLICE_PIC]++;
  if (sps.chroma_format_idc == 0) g_feat[FEAT_MONO_PIC]++;
  for (const m355_slice& sl : slices) if (sl.flags & M355_SF_DEBLOCK_DISABLED) g_feat[FEAT_DEBLOCK_OFF_SLICE]++;

  /* intra blocks: each CTB's run (one thread decodes a whole CTB), CTBs in raster order */
  struct Ref { uint32_t ctb; uint16_t t; uint32_t start, count; };
  std::vector<Ref> order;
  size_t n_ib_final = 0;
  {
    std::vector<Ref> all;
    for (size_t t = 0; t < recs.size(); t++)
      for (const Run& rn : recs[t]->runs) all.push_back(Ref{rn.ctb, (uint16_t)t, rn.start, rn.count});
    std::stable_sort(all.begin(), all.end(), [](const Ref& a, const Ref& b) { return a.ctb < b.ctb; });
    for (size_t k = 0; k < all.size(); k++) {
      if (k + 1 < all.size() && all[k + 1].ctb == all[k].ctb) continue;     /* a CTB coded twice (damaged stream): the last one stands */
      if (all[k].ctb >= ctbs.size()) continue;
      m355_ctb& c = ctbs[all[k].ctb];
      c.ib_start = (uint32_t)n_ib_final; c.ib_count = all[k].count;
      n_ib_final += all[k].count;
      order.push_back(all[k]);
    }
  }

  lap(2);
  /* Where the lists go.  Normally straight into the backend's pinned arena (m355_arena_begin: the submit then copies nothing
     and checks the records on the device); pictures with PCM units (their raw blocks are merged into the intra lists below)
     and M355_GLUE_COPY=1 take the copying m355_submit_picture through host vectors. */
  static const bool force_copy = getenv("M355_GLUE_COPY") != nullptr;
  const bool in_place = !any_pcm && !force_copy && g->n_ranks == 1;   /* (a tile-sharded context takes whole lists: submit_sharded) */
  std::unique_lock<std::mutex> api_lock(g->api_mu, std::defer_lock);
  std::vector<m355_pb> pbs;
  std::vector<uint32_t> coeffs;
  std::vector<m355_rb> rbs;
  std::vector<m355_ib> ibs;
  std::vector<uint16_t> pcm;
  m355_picture apic; memset(&apic, 0, sizeof(apic));
  m355_arena_caps caps; memset(&caps, 0, sizeof(caps));
  m355_pb* d_pbs = nullptr; uint32_t* d_co = nullptr; m355_ib* d_ibs = nullptr; m355_rb* d_rb[4] = {nullptr, nullptr, nullptr, nullptr};
  if (in_place) {
    caps.n_slices = (int32_t)slices.size(); caps.n_ctbs = (int32_t)ctbs.size(); caps.n_cus = (int32_t)cus.size() + 1; caps.n_tus = (int32_t)tus.size() + 1;
    caps.n_pbs = (int32_t)n_pb + 1; caps.n_wts = any_weighted ? (int32_t)img->slices.size() * 32 : 1; caps.n_ibs = (int32_t)n_ib_final + 1;
    for (int s = 0; s < 4; s++) caps.n_rbs[s] = (int32_t)n_rb[s] + 1;
    caps.n_coeffs = (uint32_t)n_co + 1; caps.n_pcm = 1; caps.scaling = sps.scaling_list_enable_flag ? 1 : 0;
    api_lock.lock();                                        /* the backend context from here to the end of the submit */
    if (A->m355_arena_begin(g->mctx, &caps, &apic) != M355_OK) { g->error = A->m355_last_error(); return false; }
    d_pbs = (m355_pb*)apic.pbs; d_co = (uint32_t*)apic.coeffs; d_ibs = (m355_ib*)apic.ibs;
    for (int s = 0; s < 4; s++) d_rb[s] = caps.rb_bin[s];
  } else {
    pbs.resize(n_pb); coeffs.resize(n_co); rbs.resize(n_rb[0] + n_rb[1] + n_rb[2] + n_rb[3]); ibs.resize(n_ib_final);
    d_pbs = pbs.data(); d_co = coeffs.data(); d_ibs = ibs.data();
    size_t o = 0;
    for (int s = 0; s < 4; s++) { d_rb[s] = rbs.data() + o; o += n_rb[s]; }
  }
  {
    /* one task per (thread list, kind) + slices of the intra runs: plain copies, with each thread's coefficient / residual
       offsets rebased into the picture-wide arrays */
    std::vector<size_t> pb_base(recs.size()), rb_base[4];
    size_t acc = 0;
    for (size_t t = 0; t < recs.size(); t++) { pb_base[t] = acc; acc += recs[t]->pbs.size(); }
    for (int s = 0; s < 4; s++) { rb_base[s].resize(recs.size()); acc = 0; for (size_t t = 0; t < recs.size(); t++) { rb_base[s][t] = acc; acc += recs[t]->rbs[s].size(); } }
    const size_t ib_chunk = 256, n_ib_tasks = (order.size() + ib_chunk - 1) / ib_chunk;
    const size_t per_rec = 6, n_tasks = recs.size() * per_rec + n_ib_tasks;
    parallel_tasks(n_tasks, [&](size_t k) {
      if (k < recs.size() * per_rec) {
        const size_t t = k / per_rec; const int what = (int)(k % per_rec);
        ThreadRec* r = recs[t];
        if (what == 0) { if (!r->pbs.empty()) memcpy(d_pbs + pb_base[t], r->pbs.data(), r->pbs.size() * sizeof(m355_pb)); }
        else if (what == 1) { if (!r->coeffs.empty()) memcpy(d_co + co_base[t], r->coeffs.data(), r->coeffs.size() * 4); }
        else {
          const int sb = what - 2;
          m355_rb* dst = d_rb[sb] + rb_base[sb][t];
          const std::vector<m355_rb>& v = r->rbs[sb];
          for (size_t i = 0; i < v.size(); i++) {
            m355_rb rb = v[i];
            if (rb.flags & RBF_ROTATE_PENDING) {                  /* (scale_coefficients: the chroma block's look-up, now that every tile is parsed) */
              rb.flags &= (uint8_t)~RBF_ROTATE_PENDING;
              if (img->get_pred_mode(rb.x, rb.y) == MODE_INTRA) { rb.flags |= M355_RBF_ROTATE; g_feat[FEAT_ROTATE_RB]++; }
            }
            rb.coeff_ofs += co_base[t];
            if (rb.flags & M355_RBF_DEFERRED) rb.res_ofs += res_base[t];
            dst[i] = rb;
          }
        }
      } else {
        const size_t c0 = (k - recs.size() * per_rec) * ib_chunk, c1 = std::min(order.size(), c0 + ib_chunk);
        for (size_t q = c0; q < c1; q++) {
          const Ref& o = order[q];
          m355_ib* dst = d_ibs + ctbs[o.ctb].ib_start;
          for (uint32_t i = 0; i < o.count; i++) {
            m355_ib ib = recs[o.t]->ibs[o.start + i];
            if (ib.flags & M355_IBF_HAS_RESIDUAL) ib.res_ofs += res_base[o.t];
            dst[i] = ib;
          }
        }
      }
    });
  }
  lap(3);
  if (any_pcm) {
    /* PCM coding units (slice.cc:4211-4255): the parser has stored the raw samples in the host planes (bitstream reading,
       not arithmetic); lift them into pcm[] and list one raw block per component at the unit's place in decode order */
    const int sw = sps.SubWidthC, sh = sps.SubHeightC, nc = sps.chroma_format_idc ? 3 : 1;
    std::vector<std::vector<std::pair<uint32_t, m355_ib>>> extra(ctbs.size());   /* per CTB: (z rank of the CU, block) */
    for (uint32_t ci : pcm_cus) {
      const m355_cu& cu = cus[ci];
      const uint32_t ctb = (uint32_t)((cu.y >> sps.Log2CtbSizeY) * ctbW + (cu.x >> sps.Log2CtbSizeY));
      for (int c = 0; c < nc; c++) {
        const int subw = c ? (sw == 2) : 0, subh = c ? (sh == 2) : 0;
        const int l2 = cu.log2_size - subw, nblk = (subw && !subh) ? 2 : 1, n = 1 << l2;    /* 4:2:2 chroma: two stacked squares */
        if (l2 < 2) continue;
        for (int b = 0; b < nblk; b++) {
          m355_ib ib; memset(&ib, 0, sizeof(ib));
          ib.x = (uint16_t)(cu.x >> subw); ib.y = (uint16_t)((cu.y >> subh) + b * n); ib.cidx = (uint8_t)c; ib.log2_size = (uint8_t)l2;
          ib.mode = 1; ib.flags = M355_IBF_PCM; ib.res_ofs = (uint32_t)pcm.size();
          const int bpp = img->get_bytes_per_pixel(c);
          const uint8_t* base = img->get_image_plane(c);
          const ptrdiff_t st = img->get_image_stride(c);
          for (int yy = 0; yy < n; yy++)
            for (int xx = 0; xx < n; xx++) {
              const uint8_t* q = base + ((size_t)(ib.y + yy) * st + ib.x + xx) * bpp;
              pcm.push_back(bpp == 1 ? (uint16_t)*q : *(const uint16_t*)q);
            }
          extra[ctb].push_back(std::make_pair(z_rank(cu.x, cu.y), ib));
        }
      }
    }
    std::vector<m355_ib> merged; merged.reserve(ibs.size() + 3 * pcm_cus.size());
    for (size_t ci = 0; ci < ctbs.size(); ci++) {
      m355_ctb& c = ctbs[ci];
      const uint32_t start = (uint32_t)merged.size();
      if (extra[ci].empty()) { merged.insert(merged.end(), ibs.begin() + c.ib_start, ibs.begin() + c.ib_start + c.ib_count); }
      else {
        /* recorded blocks are in decode order; key = z rank of their coding unit */
        std::vector<std::pair<uint32_t, m355_ib>> all;
        for (uint32_t i = 0; i < c.ib_count; i++) {
          const m355_ib& ib = ibs[c.ib_start + i];
          const int xl = ib.cidx ? ib.x * (sw == 2 ? 2 : 1) : ib.x, yl = ib.cidx ? ib.y * (sh == 2 ? 2 : 1) : ib.y;
          const int cl2 = img->get_log2CbSize(xl, yl);
          all.push_back(std::make_pair(z_rank((xl >> cl2) << cl2, (yl >> cl2) << cl2), ib));
        }
        all.insert(all.end(), extra[ci].begin(), extra[ci].end());
        std::stable_sort(all.begin(), all.end(), [](const std::pair<uint32_t, m355_ib>& a, const std::pair<uint32_t, m355_ib>& b) { return a.first < b.first; });
        for (auto& e : all) merged.push_back(e.second);
      }
      c.ib_start = start; c.ib_count = (uint32_t)merged.size() - start;
    }
    ibs.swap(merged);
  }

  /* ---- explicit weights: one entry per (slice, list, refIdx) (motion.cc:508-529, 571-598, 633-652) ---- */
  std::vector<m355_wt> wts;
  if (any_weighted) {
    if (img->slices.size() * 32 > 65535) { g->error = "too many slices for the weight table index"; return false; }
    const int shift1_L = std::max(2, 14 - sps.BitDepth_Y), shift1_C = std::max(2, 14 - sps.BitDepth_C);
    wts.resize(img->slices.size() * 32);
    for (size_t si = 0; si < img->slices.size(); si++) {
      const slice_segment_header* sh = img->slices[si];
      for (int l = 0; l < 2; l++)
        for (int r = 0; r < 16; r++) {
          m355_wt& w = wts[si * 32 + l * 16 + r]; memset(&w, 0, sizeof(w));
          w.w[0] = sh->LumaWeight[l][r]; w.o[0] = (int16_t)(sh->luma_offset[l][r] * (1 << sps.WpOffsetBdShiftY));
          for (int k = 0; k < 2; k++) { w.w[1 + k] = sh->ChromaWeight[l][r][k]; w.o[1 + k] = (int16_t)(sh->ChromaOffset[l][r][k] * (1 << sps.WpOffsetBdShiftC)); }
          w.log2wd_luma = (uint8_t)(sh->luma_log2_weight_denom + shift1_L);
          w.log2wd_chroma = (uint8_t)(sh->ChromaLog2WeightDenom + shift1_C);
        }
    }
  }

  lap(4);
  /* ---- frames: destination + every DPB slot a prediction block reads ---- */
  if (!api_lock.owns_lock()) api_lock.lock();
  const int dslot = job.dslot;
  if (dslot < 0) { g->error = "picture is not in the DPB"; return false; }
  if (!ensure_frame(g, dslot, img)) return false;
  pic.dst_frame = g->frame_of_slot[dslot];
  for (int i = 0; i < M355_MAX_REF_FRAMES; i++) pic.ref_frames[i] = -1;
  bool used[M355_MAX_REF_FRAMES] = {};
  for (size_t i = 0; i < n_pb; i++) {
    const m355_pb& pb = d_pbs[i];
    for (int l = 0; l < 2; l++)
      if ((pb.flags & (M355_PBF_MC_L0 << l)) && !(pb.flags & (M355_PBF_FILL_L0 << l))) used[pb.ref_slot[l]] = true;
  }
  for (int s = 0; s < M355_MAX_REF_FRAMES; s++) {
    if (!used[s]) continue;
    const de265_image* rp = job.dpb_img[s];
    if (s == dslot || !rp) { g->error = "prediction block references its own picture / an empty slot"; return false; }
    if (g->frame_of_slot[s] < 0 || g->dev_id[s] != job.dpb_id[s]) {
      if (!upload_host_planes(g, s, rp)) return false;
    }
    pic.ref_frames[s] = g->frame_of_slot[s];
  }
  collect_status(g, -1, &job);                     /* outcomes known by now (non-blocking) */
  bool from_damaged = false;
  for (int s = 0; s < M355_MAX_REF_FRAMES; s++) if (used[s] && g->damaged[s]) from_damaged = true;

  long long area = 0;
  for (int y = 0; y < sps.PicHeightInMinCbsY; y++) area += (long long)sps.PicWidthInMinCbsY;
  area *= (long long)minCb * minCb;
  if (covered != area || skipped) pp.flags |= M355_PF_CLEAR_DST;

  uint8_t sf[6 * (16 + 64 + 256 + 1024)];
  if (sps.scaling_list_enable_flag) {
    /* [sizeId][matrixID][y][x] as transform.cc:505-508 reads them from the PPS */
    uint8_t* f = sf;
    memcpy(f, pps.scaling_list.ScalingFactor_Size0, 6 * 16); f += 6 * 16;
    memcpy(f, pps.scaling_list.ScalingFactor_Size1, 6 * 64); f += 6 * 64;
    memcpy(f, pps.scaling_list.ScalingFactor_Size2, 6 * 256); f += 6 * 256;
    memcpy(f, pps.scaling_list.ScalingFactor_Size3, 6 * 1024);
    pic.scaling_factors = sf;
  }

  pic.n_slices = (int32_t)slices.size(); pic.n_ctbs = (int32_t)ctbs.size(); pic.n_cus = (int32_t)cus.size(); pic.n_tus = (int32_t)tus.size();
  pic.n_pbs = (int32_t)n_pb; pic.n_wts = (int32_t)wts.size(); pic.n_ibs = (int32_t)(in_place ? n_ib_final : ibs.size());
  pic.n_coeffs = (uint32_t)n_co; pic.n_pcm = (uint32_t)pcm.size(); pic.res_len = res_len;
  if (in_place) {
    /* the small lists and the metadata walk's output go next to the block lists in the arena */
    parallel_tasks(5, [&](size_t k) {
      if (k == 0) memcpy((void*)apic.cus, cus.data(), cus.size() * sizeof(m355_cu));
      else if (k == 1) memcpy((void*)apic.tus, tus.data(), tus.size() * sizeof(m355_tu));
      else if (k == 2) memcpy((void*)apic.ctbs, ctbs.data(), ctbs.size() * sizeof(m355_ctb));
      else if (k == 3) { memcpy((void*)apic.slices, slices.data(), slices.size() * sizeof(m355_slice)); if (!wts.empty()) memcpy((void*)apic.wts, wts.data(), wts.size() * sizeof(m355_wt)); }
      else if (pic.scaling_factors) memcpy((void*)apic.scaling_factors, sf, sizeof(sf));
    });
    pic.slices = apic.slices; pic.ctbs = apic.ctbs; pic.cus = apic.cus; pic.tus = apic.tus; pic.pbs = apic.pbs; pic.wts = apic.wts;
    pic.rbs = apic.rbs; pic.ibs = apic.ibs; pic.coeffs = apic.coeffs; pic.pcm = apic.pcm;
    if (pic.scaling_factors) pic.scaling_factors = apic.scaling_factors;
  } else {
    pic.slices = slices.data(); pic.ctbs = ctbs.data(); pic.cus = cus.data(); pic.tus = tus.data(); pic.pbs = pbs.data();
    pic.wts = wts.data(); pic.rbs = rbs.data(); pic.ibs = ibs.data(); pic.coeffs = coeffs.data(); pic.pcm = pcm.data();
  }

  lap(5);
  const auto t1 = std::chrono::steady_clock::now();
  /* M355_GLUE_ATTRIB (time attribution only — pictures are NOT decoded; profiles/r04_e2e_attribution.txt): 2 = everything but the submit */
  static const int attrib = getenv("M355_GLUE_ATTRIB") ? atoi(getenv("M355_GLUE_ATTRIB")) : 0;
  if (attrib == 2) { g->ms_walk += std::chrono::duration<double, std::milli>(t1 - t0).count(); g->n_pictures++; g->dev_id[dslot] = img->get_ID(); g->host_id[dslot] = 0xFFFFFFFFu; return true; }
  const int rc = g->n_ranks > 1 ? submit_sharded(g, pic, dslot) : A->m355_submit_picture(g->mctx, &pic);
  const auto t2 = std::chrono::steady_clock::now();
  g->ms_walk += std::chrono::duration<double, std::milli>(t1 - t0).count();
  g->ms_submit += std::chrono::duration<double, std::milli>(t2 - t1).count();
  if (rc != M355_OK) { g->error = A->m355_last_error(); return false; }
  g->dev_id[dslot] = img->get_ID();
  g->host_id[dslot] = 0xFFFFFFFFu;
  g->pending_serial[dslot] = g->n_ranks > 1 ? 0 : A->m355_last_serial(g->mctx);   /* (sharded: the lists were checked on the host, at the upload) */
  g->damaged[dslot] = from_damaged;
  if (from_damaged && img->integrity == INTEGRITY_CORRECT) img->integrity = INTEGRITY_DERIVED_FROM_FAULTY_REFERENCE;
  g->n_pictures++;
  g->dl_id[dslot] = 0xFFFFFFFFu;
  if (g->app_reads.load() && img->PicOutputFlag && !getenv("M355_GLUE_NO_PREFETCH")) {
    void* dst[3] = {nullptr, nullptr, nullptr};
    ptrdiff_t str[3] = {0, 0, 0};
    const int nc = img->get_chroma_format() == de265_chroma_mono ? 1 : 3;
    for (int c = 0; c < nc; c++) { dst[c] = img->get_image_plane(c); str[c] = img->get_image_stride(c); }
    if (A->m355_frame_download_async(g->mctx, g->frame_of_slot[dslot], dst, str) == M355_OK) { g->dl_id[dslot] = img->get_ID(); g->n_prefetched++; }
  }
  return true;
}

/* one picture's submit step, on whichever thread runs it */
void run_job(Glue* g, Glue::Job& job)
{
  if (!submit_picture(g, job)) {
    fprintf(stderr, "libde265 (MI355X glue): picture POC %d not decoded: %s\n", job.img->PicOrderCntVal, g->error.c_str());
    job.img->integrity = INTEGRITY_DECODING_ERRORS;
    std::lock_guard<std::mutex> lk(g->job_mu);
    g->deferred_warnings.push_back(DE265_WARNING_INCORRECT_ENTRY_POINT_OFFSET);   /* there is no backend-specific warning code in de265.h */
  }
  std::lock_guard<std::mutex> lk(g->mu);
  for (ThreadRec* r : job.recs) { r->clear(); g->pool.push_back(r); }
  job.recs.clear();
}

void worker_main(Glue* g)
{
  std::unique_lock<std::mutex> lk(g->job_mu);
  for (;;) {
    g->job_cv.wait(lk, [&]() { return g->stop || !g->jobs.empty(); });
    if (g->jobs.empty()) return;                             /* (stop: after the queue has drained) */
    Glue::Job job = std::move(g->jobs.front());
    g->jobs.pop_front();
    g->busy = true; g->busy_id = job.id;
    memcpy(g->busy_dpb_id, job.dpb_id, sizeof(g->busy_dpb_id));
    lk.unlock();
    run_job(g, job);
    lk.lock();
    g->busy = false; g->busy_id = 0xFFFFFFFFu;
    g->idle_cv.notify_all();
  }
}

/* wait until the worker is done with picture `id` (0xFFFFFFFF: with everything): whoever is about to touch the picture's device
   frame, its host planes or its metadata arrays (application, SEI check, the DPB recycling the image) comes through here */
void wait_submitted(Glue* g, uint32_t id, bool also_as_reference)
{
  std::unique_lock<std::mutex> lk(g->job_mu);
  g->idle_cv.wait(lk, [&]() {
    if (id == 0xFFFFFFFFu) return g->jobs.empty() && !g->busy;
    if (g->busy && g->busy_id == id) return false;
    for (const Glue::Job& j : g->jobs) if (j.id == id) return false;
    if (also_as_reference) {
      /* the submit step of a LATER picture reads this one through its DPB snapshot (integrity, host planes of a picture that
         has to be re-uploaded): the image must outlive those jobs too */
      if (g->busy) for (int s = 0; s < M355_MAX_REF_FRAMES; s++) if (g->busy_dpb_id[s] == id) return false;
      for (const Glue::Job& j : g->jobs) for (int s = 0; s < M355_MAX_REF_FRAMES; s++) if (j.dpb_id[s] == id) return false;
    }
    return true;
  });
}

/* warnings raised on the worker reach the decoder on the decoder's own thread (its error queue is not synchronised) */
void flush_warnings(Glue* g)
{
  std::vector<int> w;
  { std::lock_guard<std::mutex> lk(g->job_mu); w.swap(g->deferred_warnings); }
  for (int code : w) g->dctx->add_warning((de265_error)code, false);
}

/* picture complete (decode_some, decctx.cc:605-630): hand it to the device.  The decoder's thread only takes the picture's lists
   and a snapshot of the DPB and goes back to parsing; the submit step runs on the glue's worker (M355_GLUE_SYNC=1: here). */
void picture_complete(decoder_context* d, de265_image* img)
{
  Glue* g = glue_of(d);
  if (!g) { fprintf(stderr, "libde265 (MI355X glue): decoder without a backend context\n"); abort(); }
  const auto t0 = std::chrono::steady_clock::now();
  Glue::Job job;
  job.img = img; job.id = img->get_ID(); job.dslot = slot_of(d, img);
  for (int s = 0; s < M355_MAX_REF_FRAMES; s++) {
    job.dpb_img[s] = d->has_image(s) ? d->get_image(s) : nullptr;
    job.dpb_id[s] = job.dpb_img[s] ? job.dpb_img[s]->get_ID() : 0xFFFFFFFFu;
  }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->cur_id == img->get_ID()) { job.recs.swap(g->recs); g->cur_id = 0xFFFFFFFFu; }   /* else: a picture without a single coded block */
  }
  flush_warnings(g);
  static const int attrib = getenv("M355_GLUE_ATTRIB") ? atoi(getenv("M355_GLUE_ATTRIB")) : 0;
  if (attrib == 1) {                              /* 1 = parser + recorders only: the picture's lists are dropped here */
    std::lock_guard<std::mutex> lk(g->mu);
    for (ThreadRec* r : job.recs) { r->clear(); g->pool.push_back(r); }
    g->n_pictures++;
    return;
  }
  if (g->sync_submit) run_job(g, job);
  else {
    { std::lock_guard<std::mutex> lk(g->job_mu); g->jobs.push_back(std::move(job)); }
    g->job_cv.notify_one();
  }
  g->ms_main += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  /* (a decoded-picture-hash SEI is checked right after this call, decctx.cc:634-641: process_sei below hashes the DEVICE frame) */
}

} // namespace

/* =============================================================== replaced reference functions ===== */

/* transform.cc:645 — one coded transform block.  What scale_coefficients_internal (transform.cc:361-642) decides on the
 * host is recorded; its arithmetic is k_residual's. */
void scale_coefficients(thread_context* tctx, int xT, int yT, int x0, int y0, int nT, int cIdx, bool transform_skip_flag, bool intra, int rdpcmMode)
{
  (void)x0; (void)y0;
  de265_image* img = tctx->img;
  ThreadRec* r = rec_for(img);
  if (!r) return;
  const seq_parameter_set& sps = img->get_sps();
  const pic_parameter_set& pps = img->get_pps();
  const int log2 = ilog2(nT);
  m355_rb rb; memset(&rb, 0, sizeof(rb));
  rb.x = (uint16_t)xT; rb.y = (uint16_t)yT; rb.cidx = (uint8_t)cIdx; rb.log2_size = (uint8_t)log2;
  rb.qp = (uint8_t)(cIdx == 0 ? tctx->qPYPrime : (cIdx == 1 ? tctx->qPCbPrime : tctx->qPCrPrime));       /* transform.cc:371-377 */
  const bool cuIntra = img->get_pred_mode(xT, yT) == MODE_INTRA;      /* transform.cc:398: looked up at (xT,yT) as given — restated literally */
  /* ... which for a CHROMA block of a 4:2:0 / 4:2:2 picture is a position up / left of the block (its chroma coordinates in the luma-indexed array):
     in single-threaded decoding order a unit parsed earlier — but with tile (or WPP) threads possibly a unit of ANOTHER tile that its thread has not
     reached yet, and the reference's own output then depends on the race (tools/soak_streams.py found it: 4:2:2, tile columns, transform_skip_rotation,
     profiles/r06_v29_*).  The recorder leaves that one decision open (RBF_ROTATE_PENDING, glue-internal) and build_lists takes it once the whole picture is
     parsed: always what the reference decodes single-threaded. */
  const bool rot_on = sps.range_extension.transform_skip_rotation_enabled_flag && nT == 4;
  const bool rot_later = rot_on && cIdx != 0 && (sps.SubWidthC != 1 || sps.SubHeightC != 1);
  const bool rotate = rot_on && !rot_later && cuIntra;                /* :400-402 */
  const uint8_t rot_flag = rot_later ? RBF_ROTATE_PENDING : (rotate ? M355_RBF_ROTATE : 0);
  if (tctx->cu_transquant_bypass_flag) {
    rb.kind = M355_RK_BYPASS;
    rb.flags |= rot_flag;
    r->feat[FEAT_BYPASS_RB]++;
  } else if (transform_skip_flag) {
    rb.kind = M355_RK_SKIP;
    rb.flags |= rot_flag;
    r->feat[FEAT_SKIP_RB]++;
  } else rb.kind = (nT == 4 && cIdx == 0 && cuIntra) ? M355_RK_DST : M355_RK_DCT;                          /* :601-606 */
  if (rb.kind == M355_RK_BYPASS || rb.kind == M355_RK_SKIP) {
    if (rdpcmMode == 1) rb.flags |= M355_RBF_RDPCM_H; else if (rdpcmMode == 2) rb.flags |= M355_RBF_RDPCM_V;
    if (rdpcmMode) r->feat[FEAT_RDPCM_RB]++;
    if (rb.flags & M355_RBF_ROTATE) r->feat[FEAT_ROTATE_RB]++;
  }
  if (cIdx && sps.chroma_format_idc >= 2) r->feat[sps.chroma_format_idc == 2 ? FEAT_CHROMA_422_RB : FEAT_CHROMA_444_RB]++;
  if (sps.scaling_list_enable_flag) {                                 /* matrixID, transform.cc:493-502 */
    int m = nT == 32 ? 0 : cIdx;
    if (!intra) m += nT < 32 ? 3 : 1;
    rb.matrix_id = (uint8_t)m;
    r->feat[FEAT_SCALING_RB]++;
  }
  std::vector<m355_rb>& bin = r->rbs[log2 - 2];
  if (pps.range_extension.cross_component_prediction_enabled_flag) {
    /* slice.cc:3721-3760: ResScaleVal of a chroma block, and where its transform unit's luma block sits in the size bin */
    if (cIdx == 0) r->luma_rb = (int)bin.size();
    else if (tctx->ResScaleVal != 0 && r->luma_rb >= 0) {
      const int a = tctx->ResScaleVal < 0 ? -tctx->ResScaleVal : tctx->ResScaleVal;
      const int back = (int)bin.size() - r->luma_rb;                  /* 1 or 2 */
      if (back == 1 || back == 2)
      { rb.matrix_id |= (uint8_t)(((ilog2(a) + 1) << 4) | (tctx->ResScaleVal < 0 ? 0x80 : 0) | (back == 2 ? 8 : 0)); r->feat[FEAT_CROSS_COMP_RB]++; }
    }
  }
  const int n = tctx->nCoeff[cIdx];
  rb.coeff_ofs = (uint32_t)r->coeffs.size();
  rb.ncoeff = (uint16_t)n;
  const int16_t* lvl = tctx->coeffList[cIdx];
  const int16_t* pos = tctx->coeffPos[cIdx];
  for (int i = 0; i < n; i++) r->coeffs.push_back((uint32_t)(uint16_t)pos[i] | ((uint32_t)(uint16_t)lvl[i] << 16));
  if (intra && r->last_ib >= 0) {
    m355_ib& ib = r->ibs[r->last_ib];
    if (ib.cidx == rb.cidx && ib.x == rb.x && ib.y == rb.y && ib.log2_size == rb.log2_size && !(ib.flags & M355_IBF_HAS_RESIDUAL)) {
      /* an intra block's residual is added after its prediction, which needs its neighbours first: deferred */
      ib.flags |= M355_IBF_HAS_RESIDUAL; ib.res_ofs = r->res_len;
      rb.flags |= M355_RBF_DEFERRED; rb.res_ofs = r->res_len;
      r->res_len += (uint32_t)nT * nT;
    }
  }
  bin.push_back(rb);
}

/* intrapred.cc:321 — one intra-predicted block, in decode order */
void decode_intra_prediction(de265_image* img, int xB0, int yB0, enum IntraPredMode intraPredMode, int nT, int cIdx)
{
  ThreadRec* r = rec_for(img);
  if (!r) return;
  const seq_parameter_set& sps = img->get_sps();
  m355_ib ib; memset(&ib, 0, sizeof(ib));
  ib.x = (uint16_t)xB0; ib.y = (uint16_t)yB0; ib.cidx = (uint8_t)cIdx; ib.log2_size = (uint8_t)ilog2(nT); ib.mode = (uint8_t)intraPredMode;
  if (sps.range_extension.implicit_rdpcm_enabled_flag && img->get_cu_transquant_bypass(xB0, yB0)) { ib.flags |= M355_IBF_DISABLE_BOUNDARY_FILTER; r->feat[FEAT_NO_BOUNDARY_FILTER_IB]++; }   /* intrapred.cc:306-308 */
  const int xl = cIdx ? xB0 * sps.SubWidthC : xB0, yl = cIdx ? yB0 * sps.SubHeightC : yB0;
  const uint32_t ctb = (uint32_t)((yl >> sps.Log2CtbSizeY) * sps.PicWidthInCtbsY + (xl >> sps.Log2CtbSizeY));
  if (r->runs.empty() || r->runs.back().ctb != ctb) r->runs.push_back(Run{ctb, (uint32_t)r->ibs.size(), 0});
  r->runs.back().count++;
  r->last_ib = (int)r->ibs.size();
  r->ibs.push_back(ib);
}

/* motion.cc:288 — one prediction block.  The decisions of motion.cc:340-688 (bi -> uni demotion, unusable reference ->
 * 1 << 13 fill, which weighted-prediction branch) are taken here, on the host; the arithmetic is k_inter's. */
void generate_inter_prediction_samples(base_context* ctx, const slice_segment_header* shdr, de265_image* img, int xC, int yC, int xB, int yB,
                                       int nCS, int nPbW, int nPbH, const PBMotion* vi)
{
  (void)nCS;
  ThreadRec* r = rec_for(img);
  if (!r) return;
  const pic_parameter_set* pps = shdr->pps.get();
  const seq_parameter_set* sps = pps->sps.get();
  if (sps->BitDepth_Y != img->get_bit_depth(0) || sps->BitDepth_C != img->get_bit_depth(1)) {           /* motion.cc:304-309 */
    img->integrity = INTEGRITY_DECODING_ERRORS;
    ctx->add_warning(DE265_WARNING_BIT_DEPTH_OF_CURRENT_IMAGE_DOES_NOT_MATCH_SPS, false);
    r->skipped_pbs++;
    return;
  }
  if (sps->chroma_format_idc != img->get_chroma_format()) {                                               /* :311-315 */
    img->integrity = INTEGRITY_DECODING_ERRORS;
    ctx->add_warning(DE265_WARNING_CHROMA_OF_CURRENT_IMAGE_DOES_NOT_MATCH_SPS, false);
    r->skipped_pbs++;
    return;
  }
  const int xP = xC + xB, yP = yC + yB;
  int predFlag[2] = {vi->predFlag[0], vi->predFlag[1]};
  if (pps->weighted_pred_flag == 0 && predFlag[0] && predFlag[1] && vi->mv[0].x == vi->mv[1].x && vi->mv[0].y == vi->mv[1].y &&
      shdr->RefPicList[0][vi->refIdx[0]] == shdr->RefPicList[1][vi->refIdx[1]])
    predFlag[1] = 0;                                                                                      /* :348-357 */

  m355_pb pb; memset(&pb, 0, sizeof(pb));
  pb.x = (uint16_t)xP; pb.y = (uint16_t)yP; pb.w = (uint8_t)nPbW; pb.h = (uint8_t)nPbH;
  pb.ref_slot[0] = pb.ref_slot[1] = -1;
  for (int l = 0; l < 2; l++) {
    if (vi->predFlag[l]) {                       /* as stored: what boundary-strength derivation compares (deblock.cc:294-366) */
      pb.flags |= (uint8_t)(M355_PBF_PRED_L0 << l);
      const int idx = shdr->RefPicList[l][vi->refIdx[l]];
      pb.ref_slot[l] = (int8_t)((idx >= 0 && idx < M355_MAX_REF_FRAMES) ? idx : -1);
      pb.mv[l][0] = vi->mv[l].x; pb.mv[l][1] = vi->mv[l].y;
    }
    if (!predFlag[l]) continue;
    const int idx = shdr->RefPicList[l][vi->refIdx[l]];
    const de265_image* refPic = (idx >= 0 && idx < M355_MAX_REF_FRAMES) ? ctx->get_image((uint16_t)idx) : nullptr;
    bool usable = true;
    if (!refPic || refPic->PicState == UnusedForReference) { ctx->add_warning(DE265_WARNING_NONEXISTING_REFERENCE_PICTURE_ACCESSED, false); usable = false; }           /* :362-367 */
    else if (refPic->get_width(0) != sps->pic_width_in_luma_samples || refPic->get_height(0) != sps->pic_height_in_luma_samples ||
             img->get_chroma_format() != refPic->get_chroma_format()) { ctx->add_warning(DE265_WARNING_REFERENCE_IMAGE_SIZE_DOES_NOT_MATCH_SPS, false); usable = false; }   /* :368-374 */
    else if (img->get_bit_depth(0) != refPic->get_bit_depth(0) || img->get_bit_depth(1) != refPic->get_bit_depth(1)) {
      ctx->add_warning(DE265_WARNING_REFERENCE_IMAGE_BIT_DEPTH_DOES_NOT_MATCH, false); usable = false;                                                               /* :375-380 */
    }
    if (!usable) { img->integrity = INTEGRITY_DECODING_ERRORS; pb.flags |= (uint8_t)(M355_PBF_FILL_L0 << l); r->feat[FEAT_FILL_PB]++; }
  }
  /* weighted sample prediction: which branch of motion.cc:493-688 */
  bool weighted = false, write = true;
  if (shdr->slice_type == SLICE_TYPE_P) {
    if (predFlag[0] == 1 && predFlag[1] == 0) { pb.flags |= M355_PBF_MC_L0; weighted = pps->weighted_pred_flag != 0; }
    else write = false;
  } else {
    if (predFlag[0] == 1 && predFlag[1] == 1) { pb.flags |= M355_PBF_MC_L0 | M355_PBF_MC_L1; weighted = pps->weighted_bipred_flag != 0; }
    else if (predFlag[0] == 1 || predFlag[1] == 1) { pb.flags |= predFlag[0] ? M355_PBF_MC_L0 : M355_PBF_MC_L1; weighted = pps->weighted_bipred_flag != 0; }
    else write = false;
  }
  if (!write) {
    ctx->add_warning(DE265_WARNING_BOTH_PREDFLAGS_ZERO, false);
    img->integrity = INTEGRITY_DECODING_ERRORS;
    r->skipped_pbs++;
    return;
  }
  /* a FILL of a list that is not interpolated has no meaning for the kernels */
  for (int l = 0; l < 2; l++) if (!(pb.flags & (M355_PBF_MC_L0 << l))) pb.flags &= (uint8_t)~(M355_PBF_FILL_L0 << l);
  if (weighted) {
    r->any_weighted = true;
    pb.flags |= M355_PBF_WEIGHTED;
    const unsigned si = img->get_SliceHeaderIndex(xP, yP);
    for (int l = 0; l < 2; l++) pb.wt_idx[l] = (uint16_t)(si * 32 + l * 16 + (vi->refIdx[l] & 15));
    r->feat[FEAT_WEIGHTED_PB]++;
    if (si > 0) r->feat[FEAT_WEIGHTED_PB_LATER_SLICE]++;
  }
  r->pbs.push_back(pb);
}

/* decctx.cc:1783 / 1811 — the picture is parsed: instead of filtering it on the host, submit it */
void decoder_context::run_postprocessing_filters_sequential(de265_image* img) { picture_complete(this, img); }
void decoder_context::run_postprocessing_filters_parallel(image_unit* imgunit) { picture_complete(this, imgunit->img); }

/* sei.cc:436 — SEI messages attached to a decoded picture.  The reference hashes the host planes (compute_MD5 /
 * compute_CRC_8bit_fast / compute_checksum, sei.cc:161-274, called from process_sei_decoded_picture_hash :276-356); here the
 * picture lives on the device and is hashed THERE (m355_frame_hash: CRC and checksum by reduction kernels, nothing is copied back;
 * MD5 — one serial chain per plane — by the backend's host threads on a copy it brings back itself): a stream that carries hash
 * SEIs costs no download into the application's planes either. */
de265_error process_sei(const sei_message* sei, de265_image* img)
{
  if (sei->payload_type != sei_payload_type_decoded_picture_hash || !img->decctx->param_sei_check_hash) return DE265_OK;
  if (img->PicOutputFlag == false) return DE265_OK;                          /* sei.cc:280-287 */
  Glue* g = glue_of(img->decctx);
  if (!g) return DE265_OK;
  wait_submitted(g, img->get_ID());
  std::lock_guard<std::mutex> api_lock(g->api_mu);
  const int slot = slot_of(g->dctx, img);
  if (slot < 0 || g->frame_of_slot[slot] < 0 || g->dev_id[slot] != img->get_ID()) return DE265_OK;   /* not a picture the backend decoded */
  collect_status(g, slot);                                                    /* waits for its decode; a rejected picture is reported there */
  const sei_decoded_picture_hash* want = &sei->data.decoded_picture_hash;
  const int type = want->hash_type == sei_decoded_picture_hash_type_MD5 ? M355_HASH_MD5 : (want->hash_type == sei_decoded_picture_hash_type_CRC ? M355_HASH_CRC : M355_HASH_CHECKSUM);
  if (want->hash_type > sei_decoded_picture_hash_type_checksum) return DE265_OK;
  m355_picture_hash got; memset(&got, 0, sizeof(got));
  if (api()->m355_frame_hash(g->mctx, g->frame_of_slot[slot], type, &got) != M355_OK) {
    fprintf(stderr, "libde265 (MI355X glue): picture hash: %s\n", api()->m355_last_error());
    return DE265_ERROR_CHECKSUM_MISMATCH;
  }
  g->n_hashed++;
  const int n = img->get_sps().chroma_format_idc == 0 ? 1 : 3;
  for (int c = 0; c < n; c++) {
    if (type == M355_HASH_MD5 && memcmp(got.md5[c], want->md5[c], 16) != 0) return DE265_ERROR_CHECKSUM_MISMATCH;
    if (type == M355_HASH_CRC && got.crc[c] != want->crc[c]) return DE265_ERROR_CHECKSUM_MISMATCH;
    if (type == M355_HASH_CHECKSUM && got.checksum[c] != want->checksum[c]) return DE265_ERROR_CHECKSUM_MISMATCH;
  }
  return DE265_OK;
}

/* fallback.cc:28 — the only table filler this build's decctx.cc knows (no SIMD switches in its config.h).  The reference's pixel
 * kernels (fallback-*.cc, x86/*) are not compiled into this library at all: the table is filled with the counting traps. */
void init_acceleration_functions_fallback(struct acceleration_functions* accel) { install_traps(*accel); }

/* deblock.cc / sao.cc are not part of this build; their entry points only exist so that the (unreachable) weakened
 * originals of the two functions above still link */
static void pixel_path_trap(const char* what) { fprintf(stderr, "libde265 (MI355X glue): CPU pixel path reached: %s\n", what); abort(); }
void apply_deblocking_filter(de265_image*) { pixel_path_trap("apply_deblocking_filter"); }
void add_deblocking_tasks(image_unit*) { pixel_path_trap("add_deblocking_tasks"); }
void apply_sample_adaptive_offset(de265_image*) { pixel_path_trap("apply_sample_adaptive_offset"); }
void apply_sample_adaptive_offset_sequential(de265_image*) { pixel_path_trap("apply_sample_adaptive_offset_sequential"); }
bool add_sao_tasks(image_unit*, int) { pixel_path_trap("add_sao_tasks"); return false; }

/* =============================================================== public API wrappers (de265.h) ==== */

extern "C" {

LIBDE265_API de265_decoder_context* de265_new_decoder()
{
  Api* A = api();
  if (!A) return nullptr;
  if (A->m355_device_count() < 1) { fprintf(stderr, "libde265 (MI355X glue): no HIP device — this build has no CPU pixel path\n"); return nullptr; }
  de265_decoder_context* c = m355ref_de265_new_decoder();
  if (!c) return nullptr;
  Glue* g = new Glue;
  g->dctx = (decoder_context*)c;
  g->dctx->param_disable_sao = true;              /* (Glue::user_disable_sao holds the application's setting) */
  int dev = 0;
  if (const char* e = getenv("M355_DEVICE")) dev = atoi(e);
  if (A->m355_create(dev, &g->mctx) != M355_OK) {
    fprintf(stderr, "libde265 (MI355X glue): %s\n", A->m355_last_error());
    delete g; m355ref_de265_free_decoder(c);
    return nullptr;
  }
  g->rctx.push_back(g->mctx);
  if (const char* e = getenv("M355_GLUE_RANKS")) {
    const int n = atoi(e), ndev = A->m355_device_count();
    if (n >= 2 && n <= 64) {
      for (int r = 1; r < n; r++) {
        m355_ctx* x = nullptr;
        if (A->m355_create((dev + r) % ndev, &x) != M355_OK) { fprintf(stderr, "libde265 (MI355X glue): rank %d: %s\n", r, A->m355_last_error()); break; }
        g->rctx.push_back(x);
        g->rframe.push_back(std::vector<int>(M355_MAX_REF_FRAMES, -1));
      }
      if ((int)g->rctx.size() == n && A->m355_group_create(g->rctx.data(), n, &g->group) == M355_OK) g->n_ranks = n;
      else {
        fprintf(stderr, "libde265 (MI355X glue): M355_GLUE_RANKS=%d could not be set up (%s): one context\n", n, A->m355_last_error());
        for (size_t r = 1; r < g->rctx.size(); r++) A->m355_destroy(g->rctx[r]);
        g->rctx.resize(1); g->rframe.clear();
      }
    }
  }
  /* three lanes: the depth every measurement of the backend favours (independent 4K pictures 0.096 ms with two lanes, 0.070 with three, more do not help:
     the runtime's streams share four hardware queues) and the one from which a dependent chain's picture can run its front part beside its reference's tail
      */
  int depth = 3;
  if (const char* e = getenv("M355_PIPELINE_DEPTH")) depth = atoi(e);
  if (depth >= 1 && depth <= 16) for (m355_ctx* x : g->rctx) A->m355_set_pipeline_depth(x, g->n_ranks > 1 ? std::min(depth, 3) : depth);
  g->sync_submit = getenv("M355_GLUE_SYNC") != nullptr;
  if (!g->sync_submit) g->worker = std::thread(worker_main, g);
  install_traps(g->dctx->acceleration);
  de265_image_allocation alloc = {glue_get_buffer, glue_release_buffer};
  de265_set_image_allocation_functions(c, &alloc, g);
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_glues.push_back(g);
  g_reg_gen++;
  return c;
}

/* de265.cc:515-560 / 589-620, with DISABLE_SAO kept in the glue (see Glue::user_disable_sao) */
LIBDE265_API void de265_set_parameter_bool(de265_decoder_context* c, enum de265_param param, int value)
{
  Glue* g = glue_of((decoder_context*)c);
  if (g && param == DE265_DECODER_PARAM_DISABLE_SAO) { g->user_disable_sao = !!value; return; }
  m355ref_de265_set_parameter_bool(c, param, value);
}
LIBDE265_API int de265_get_parameter_bool(de265_decoder_context* c, enum de265_param param)
{
  Glue* g = glue_of((decoder_context*)c);
  if (g && param == DE265_DECODER_PARAM_DISABLE_SAO) return g->user_disable_sao;
  return m355ref_de265_get_parameter_bool(c, param);
}

LIBDE265_API de265_error de265_free_decoder(de265_decoder_context* c)
{
  Glue* g = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_glues.size(); i++)
      if ((de265_decoder_context*)g_glues[i]->dctx == c) { g = g_glues[i]; g_glues.erase(g_glues.begin() + i); g_reg_gen++; break; }
  }
  if (g) {
    wait_submitted(g, 0xFFFFFFFFu);
    std::lock_guard<std::mutex> api_lock(g->api_mu);
    collect_status(g, -1);
    for (size_t r = 1; r < g->rctx.size(); r++) api()->m355_wait(g->rctx[r]);
    while (api()->m355_wait(g->mctx) != M355_OK) {           /* (one rejected picture per call) */
      fprintf(stderr, "libde265 (MI355X glue): at shutdown: %s\n", api()->m355_last_error());
      if (++g->n_rejected > 1000) break;
    }
  }
  if (g && getenv("M355_GLUE_STATS"))
    fprintf(stderr, "m355 glue: %lld pictures submitted, %lld uploaded, %lld downloaded; host ms per picture: on the decoder's thread %.3f, on the worker: lists %.3f, submit %.3f; download %.3f ms each; cpu pixel calls %lld\n",
            g->n_pictures, g->n_uploads, g->n_downloads, g->n_pictures ? g->ms_main / g->n_pictures : 0.0, g->n_pictures ? g->ms_walk / g->n_pictures : 0.0, g->n_pictures ? g->ms_submit / g->n_pictures : 0.0,
            g->n_downloads ? g->ms_download / g->n_downloads : 0.0, g_cpu_pixel_calls.load());
  if (g && getenv("M355_GLUE_STATS") && atoi(getenv("M355_GLUE_STATS")) >= 2 && g->n_pictures)
    fprintf(stderr, "m355 glue: lists per picture: headers + CTBs %.3f, CU / TU walk %.3f, merge + sort %.3f, arena + block lists %.3f, PCM + weights %.3f, frames + small lists %.3f ms\n",
            g->ms_phase[0] / g->n_pictures, g->ms_phase[1] / g->n_pictures, g->ms_phase[2] / g->n_pictures, g->ms_phase[3] / g->n_pictures, g->ms_phase[4] / g->n_pictures, g->ms_phase[5] / g->n_pictures);
  const de265_error e = m355ref_de265_free_decoder(c);      /* releases the images into the pool */
  if (g) {
    { std::lock_guard<std::mutex> lk(g->job_mu); g->stop = true; }
    g->job_cv.notify_all();
    if (g->worker.joinable()) g->worker.join();
    if (g->group) api()->m355_group_destroy(g->group);
    for (size_t r = 1; r < g->rctx.size(); r++) api()->m355_destroy(g->rctx[r]);
    api()->m355_destroy(g->mctx);
    g->planes.drain();
    for (ThreadRec* r : g->recs) { r->clear(); }
    delete g;                                                /* ThreadRecs stay allocated (see rec_for) */
  }
  return e;
}

/* The application touches a decoded picture's samples through this call (de265.cc:729-738; dec265's writer, a player's
 * upload): only then is the picture brought back from the device — an application that never looks (dec265 -q without -o, or one
 * that takes the device frame by other means) costs no transfer at all. */
LIBDE265_API const uint8_t* de265_get_image_plane(const struct de265_image* img, int channel, int* stride)
{
  if (img && img->decctx) {
    Glue* g = glue_of(img->decctx);
    if (g) download_if_needed(g, const_cast<de265_image*>(img));
  }
  return m355ref_de265_get_image_plane(img, channel, stride);
}

/* test / diagnostics hooks of this build (not part of de265.h) */
LIBDE265_API long long m355_glue_cpu_pixel_calls(void) { return g_cpu_pixel_calls.load(); }
LIBDE265_API const char* m355_glue_backend_path(void) { Api* A = api(); return A ? A->path.c_str() : ""; }
/* how often the recorder took each feature branch since the process started (order: FEAT_* above) -> number of counters */
LIBDE265_API int m355_glue_feature_counts(long long* out, int n)
{
  for (int k = 0; k < n && k < M355_GLUE_N_FEATURES; k++) out[k] = g_feat[k].load();
  return M355_GLUE_N_FEATURES;
}
/* output pictures whose download was started behind their decode (before the application asked) */
LIBDE265_API long long m355_glue_prefetched_pictures(de265_decoder_context* c) { Glue* g = glue_of((decoder_context*)c); if (g) wait_submitted(g, 0xFFFFFFFFu); return g ? g->n_prefetched : -1; }
LIBDE265_API long long m355_glue_hashed_pictures(de265_decoder_context* c) { Glue* g = glue_of((decoder_context*)c); if (g) wait_submitted(g, 0xFFFFFFFFu); return g ? g->n_hashed : -1; }
LIBDE265_API long long m355_glue_rejected_pictures(de265_decoder_context* c) { Glue* g = glue_of((decoder_context*)c); if (g) wait_submitted(g, 0xFFFFFFFFu); return g ? g->n_rejected : -1; }
LIBDE265_API int m355_glue_stats(de265_decoder_context* c, long long* pictures, long long* uploads, long long* downloads)
{
  Glue* g = glue_of((decoder_context*)c);
  if (!g) return -1;
  wait_submitted(g, 0xFFFFFFFFu);
  if (pictures) *pictures = g->n_pictures;
  if (uploads) *uploads = g->n_uploads;
  if (downloads) *downloads = g->n_downloads;
  return 0;
}

} /* extern "C" */
