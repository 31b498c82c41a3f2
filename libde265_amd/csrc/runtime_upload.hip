/*
 * runtime_upload.hip — host side of the picture layer, part 2 of 4: the host thread pool, validation of the work lists, the intra schedule
 * (dependency levels, exec records, work list) and upload(): lists -> one pinned arena -> the device.
 */
#include "runtime_internal.h"

extern "C" {
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
  /* INTRA PICTURES (8 or more intra blocks per CTB, as *dense below) get their levels from a picture-wide CLOCK instead of the
     dependency depth inside the CTB: t(block) = the latest of its producers' t + cost — producers in its own CTB AND the blocks of
     neighbour CTBs whose samples its mode reads (those a hand-off later) —, in units of one small block (4x4 / 8x8: 1, 16x16: 2,
     32x32: 2 — k_intra shares those among a component's waves —, CTB-to-CTB hand-off: 6 — the times of profiles/r03_*_intra_level_profile_* and r05_v14, rounded); a CTB's levels are the
     distinct t of its blocks, in order.  Why: with depth-only levels a block at the CTB's left edge that reads nothing of its own
     CTB sits in level 0 and polls there for a sample the left CTB produces in its LAST level — and the level's barrier holds the
     whole workgroup for it, i.e. neighbouring CTBs run one after the other instead of half a CTB apart (tools/intra_sim.py models
     k_intra's chain on the 1080p picture of config 2: 981 -> 758 us with the hand-offs as measured).  Blocks now come up when the
     clock says their inputs have arrived — which is also what lets the halo keeper (k_intra.hip) have them in LDS by then.  The CTBs
     are walked in wavefront order (x + 2y) by the pool's threads, each waiting for its left and top-right neighbour's clock. */
  const bool timed = one_sided && (size_t)pic->n_ibs >= 8 * (size_t)pic->n_ctbs;
  static const bool level_stats = getenv("M355_INTRA_LEVEL_STATS") != nullptr;
  double span[4] = {0, 0, 0, 0};
  std::mutex span_mu;
  /* (an intra picture has hundreds of blocks per CTB: smaller shares, so that a 1080p picture's 510 CTBs still use the whole pool) */
  /* (timed) per CTB and component: t + cost of the block behind each 4x4 unit of the CTB's right column [0..15] and bottom row
     [16..31], -1 = no intra block there; tile column / row of every CTB column / row */
  std::vector<int32_t> edge_t;
  std::vector<uint16_t> tcol, trow;
  if (timed) {
    edge_t.assign((size_t)pic->n_ctbs * 96, -1);
    tcol.resize((size_t)ctbW); trow.resize((size_t)ctbH);
    for (int i = 0; i < pp.num_tile_cols; i++) for (int x = pp.col_bd[i]; x < pp.col_bd[i + 1] && x < ctbW; x++) tcol[(size_t)x] = (uint16_t)i;
    for (int i = 0; i < pp.num_tile_rows; i++) for (int y = pp.row_bd[i]; y < pp.row_bd[i + 1] && y < ctbH; y++) trow[(size_t)y] = (uint16_t)i;
  }
  /* (hand-offs of 3 / 6 / 10 units and a 32x32 block of 4 / 5 measured the same within 0.5 %: profiles/r05_v16_*) */
  constexpr int T_HANDOFF = 6, T_16 = 2, T_32 = 2;          /* (k_intra shares a 32x32 block among its component's waves) */
  struct Scratch {
    std::vector<std::pair<uint32_t, uint32_t>> key, sorted;      /* (level << 2 | cidx, index) */
    std::vector<int32_t> tstart, tvals;
    std::vector<uint32_t> histv;                         /* stable counting sort of a CTB's keys (grows once) */
    long long my_blocks = 0, my_ctbs = 0, my_levels = 0;
    double my_span[4] = {0, 0, 0, 0};
  };
  auto schedule_ctb = [&](size_t c, Scratch& S) {
    auto& key = S.key; auto& sorted = S.sorted;
    long long& my_blocks = S.my_blocks; long long& my_ctbs = S.my_ctbs; long long& my_levels = S.my_levels; double* const my_span = S.my_span;
    {
      const m355_ctb& ctb = pic->ctbs[c];
      log2_waves[c] = 0; plan_count[c] = 0; touch[c] = 0; need[c] = 0;
      if (!ctb.ib_count) return;
      my_ctbs++; my_blocks += ctb.ib_count;
      const int cx = (int)c % ctbW, cy = (int)c / ctbW;
      int8_t grid[3][16][16];                            /* level of the block covering each 4x4 unit (a chain in a CTB is < 64 long) */
      int32_t tgrid[3][16][16];                          /* (timed) t + cost of the block covering each unit */
      memset(grid, 0xFF, sizeof(grid));                  /* -1: no intra block of this CTB there (yet) */
      if (timed) { memset(tgrid, 0xFF, sizeof(tgrid)); S.tstart.assign(ctb.ib_count, 0); }
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
            else if (m > 26 && m <= 34) { te = std::min(2 * nT, nT + ((nT * mag[m - 26]) >> 5) + 2); le = uy > 0 ? 0 : 2 * nT; }
            else if (m >= 2 && m <= 9) { le = 2 * nT; te = ux > 0 ? 0 : 2 * nT; }
            /* (a mode outside 0..34 — lists validated on the device only — keeps the full 2nT range, as m355_intra_used_entries does) */
            if (filt) { if (te) te = std::min(2 * nT, te + 1); if (le) le = 2 * nT; }
            top_e = te; left_e = le;
          }
          const int top_u = (top_e + 3) >> 2, left_u = (left_e + 3) >> 2;   /* units beside the corner */
          if (!timed) for (int t = -1; t < 2 * n4; t++) {
            if (t < left_u && ux - 1 >= 0 && uy + t >= 0 && uy + t < 16) level = std::max(level, grid[ib.cidx][uy + t][ux - 1] + 1);
            if (t < top_u && uy - 1 >= 0 && ux + t >= 0 && ux + t < 16) level = std::max(level, grid[ib.cidx][uy - 1][ux + t] + 1);
          }
          if (timed) {
            /* the clock: in-CTB producers over the same (conservative) ranges as the levels above, producers in neighbour CTBs over
               the entries the mode READS (m355_intra_used_entries: what k_intra's plan keeps — the rest is never waited for) */
            int ute, ule;
            m355_intra_used_entries(ib.mode, ib.log2_size, ib.cidx, pp.chroma_format_idc, pp.flags, ib.flags, &ute, &ule);
            const int utop = (ute + 3) >> 2, uleft = (ule + 3) >> 2;
            const int cu = ((1 << pp.log2_ctb_size) >> csw) >> 2, cv = ((1 << pp.log2_ctb_size) >> csh) >> 2;   /* the CTB in units */
            int32_t ts = 0;
            auto nb_edge = [&](int dx, int dy, int slot) -> int32_t {       /* neighbour CTB (cx + dx, cy + dy), same tile, or -1 */
              const int nx = cx + dx, ny = cy + dy;
              if (nx < 0 || ny < 0 || nx >= ctbW || tcol[(size_t)nx] != tcol[(size_t)cx] || trow[(size_t)ny] != trow[(size_t)cy]) return -1;
              return edge_t[((size_t)ny * ctbW + nx) * 96 + (size_t)ib.cidx * 32 + slot];
            };
            for (int t = -1; t < 2 * n4; t++) {
              const int ly_ = uy + t, lx_ = ux + t;
              if (ux - 1 >= 0) { if (t < left_u && ly_ >= 0 && ly_ < 16) ts = std::max(ts, tgrid[ib.cidx][ly_][ux - 1]); }
              else if ((t < uleft || t == -1) && ly_ < cv) {
                const int32_t e = ly_ >= 0 ? nb_edge(-1, 0, ly_) : nb_edge(-1, -1, 16 + cu - 1);
                if (e >= 0) ts = std::max(ts, e + T_HANDOFF);
              }
              if (uy - 1 >= 0) { if (t < top_u && lx_ >= 0 && lx_ < 16) ts = std::max(ts, tgrid[ib.cidx][uy - 1][lx_]); }
              else if (t < utop || t == -1) {
                const int32_t e = lx_ < 0 ? nb_edge(-1, -1, 16 + cu - 1) : (lx_ < cu ? nb_edge(0, -1, 16 + lx_) : (lx_ - cu < cu ? nb_edge(1, -1, 16 + lx_ - cu) : -1));
                if (e >= 0) ts = std::max(ts, e + T_HANDOFF);
              }
            }
            S.tstart[k] = ts;
          }
        }
        if (timed) {
          const int32_t td = S.tstart[k] + (ib.log2_size >= 5 ? T_32 : (ib.log2_size == 4 ? T_16 : 1));
          for (int y = uy; y < uy + n4 && y < 16; y++) for (int x = ux; x < ux + n4 && x < 16; x++) tgrid[ib.cidx][y][x] = td;
        }
        level = std::min(level, 126);                    /* (only overlapping blocks — rejected below — could get there) */
        for (int y = uy; y < uy + n4 && y < 16; y++)
          for (int x = ux; x < ux + n4 && x < 16; x++) { if (grid[ib.cidx][y][x] >= 0) clash = true; grid[ib.cidx][y][x] = (int8_t)level; }
        key.push_back(std::make_pair(((uint32_t)level << 2) | ib.cidx, k));
      }
      if (clash) { int e = -1; overlap.compare_exchange_strong(e, (int)c); }
      if (timed) {
        /* levels = the distinct start times, in order (a CTB's times span a few hundred units: ranks by counting); the CTB's edges
           for the neighbours that follow */
        int32_t tmin = INT32_MAX, tmax = 0;
        for (uint32_t k = 0; k < ctb.ib_count; k++) { tmin = std::min(tmin, S.tstart[k]); tmax = std::max(tmax, S.tstart[k]); }
        const size_t span_t = (size_t)(tmax - tmin) + 1;
        if (span_t <= 8192) {
          S.tvals.assign(span_t + 1, 0);
          for (uint32_t k = 0; k < ctb.ib_count; k++) S.tvals[(size_t)(S.tstart[k] - tmin)] = 1;
          int32_t r = 0;
          for (size_t i = 0; i < span_t; i++) { const int32_t u = S.tvals[i]; S.tvals[i] = r; r += u; }
          for (auto& e : key) e.first = ((uint32_t)S.tvals[(size_t)(S.tstart[e.second] - tmin)] << 2) | (e.first & 3u);
        } else {
          S.tvals.assign(S.tstart.begin(), S.tstart.end());
          std::sort(S.tvals.begin(), S.tvals.end());
          S.tvals.erase(std::unique(S.tvals.begin(), S.tvals.end()), S.tvals.end());
          for (auto& e : key) e.first = ((uint32_t)(std::lower_bound(S.tvals.begin(), S.tvals.end(), S.tstart[e.second]) - S.tvals.begin()) << 2) | (e.first & 3u);
        }
        int32_t* const et = &edge_t[c * 96];
        for (int q = 0; q < 3; q++) {
          const int csw = q ? (sw == 2) : 0, csh = q ? (sh == 2) : 0;
          const int cu = ((1 << pp.log2_ctb_size) >> csw) >> 2, cv = ((1 << pp.log2_ctb_size) >> csh) >> 2;
          for (int i = 0; i < 16; i++) { et[q * 32 + i] = i < cv ? tgrid[q][i][cu - 1] : -1; et[q * 32 + 16 + i] = i < cu ? tgrid[q][cv - 1][i] : -1; }
        }
      }
      {
        uint32_t kmax = 0;
        for (const auto& e : key) kmax = std::max(kmax, e.first);
        if (S.histv.size() < (size_t)kmax + 2) S.histv.resize((size_t)kmax + 2);
        uint32_t* const hist = S.histv.data();
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
        /* a 32x32 block is SHARED by its component's waves (k_intra.hip: each predicts its rows): all the waves there are */
        if (ib.log2_size >= M355_INTRA_SHARE_MIN_LOG2 && !(ib.flags & M355_IBF_PCM)) widest = std::max(widest, 8u);
      }
      plan_count[c] = rel;
      log2_waves[c] = widest >= 5 ? 3 : (widest >= 3 ? 2 : (widest == 2 ? 1 : 0));
      if (level_stats) {
        /* makespan of the CTB's levels in block units (4x4 / 8x8 = 1, 16x16 = 2, 32x32 = 4.5: the per-block times of
           profiles/r03_*_intra_level_profile_*) under k_intra's wave policies: blocks of a (level, component) go round the
           component's waves (8 + 2 + 2, 8 + 4 + 4) or round all waves (12, 16) in record order */
        static const int GW[4][3] = {{8, 2, 2}, {8, 4, 4}, {12, 0, 0}, {16, 16, 16}};
        double ms[4] = {0, 0, 0, 0};
        uint32_t k0 = 0;
        while (k0 < ctb.ib_count) {
          uint32_t k1 = k0;
          while (k1 < ctb.ib_count && (key[k1].first >> 2) == (key[k0].first >> 2)) k1++;
          for (int pol = 0; pol < 4; pol++) {
            double load[3][16]; memset(load, 0, sizeof(load));
            int cnt[3] = {0, 0, 0};
            for (uint32_t k = k0; k < k1; k++) {
              const m355_ib& ib = pic->ibs[ctb.ib_start + key[k].second];
              const double cost = ib.log2_size >= 5 ? 4.5 : (ib.log2_size == 4 ? 2.0 : 1.0);
              const int comp = GW[pol][1] ? ib.cidx : 0;
              load[comp][cnt[comp]++ % GW[pol][comp]] += cost;
            }
            double m = 0;
            for (int a = 0; a < 3; a++) for (int b = 0; b < 16; b++) m = std::max(m, load[a][b]);
            ms[pol] += m;
          }
          k0 = k1;
        }
        for (int pol = 0; pol < 4; pol++) my_span[pol] += ms[pol];
      }
      if (ctb.ib_count) my_levels += (key[ctb.ib_count - 1].first >> 2) + 1;
    }
  };
  auto fold_stats = [&](const Scratch& S) {
    n_blocks += S.my_blocks; n_intra_ctbs += S.my_ctbs; n_levels += S.my_levels;
    if (level_stats) { std::lock_guard<std::mutex> lk(span_mu); for (int q = 0; q < 4; q++) span[q] += S.my_span[q]; }
  };
  if (!timed) {
    parallel_ranges((size_t)pic->n_ctbs, (size_t)pic->n_ibs >= 8 * (size_t)pic->n_ctbs ? 16 : 256, [&](size_t cb, size_t ce) {
      Scratch S;
      for (size_t c = cb; c < ce; c++) schedule_ctb(c, S);
      fold_stats(S);
    });
  } else {
    /* wavefront order: a CTB's clock needs those of its left and top-right neighbours (which bring top-left and top with them);
       the pool's threads claim CTBs in that order and wait for the two flags — a thread only ever waits for a CTB claimed before
       its own, i.e. one that another running thread is working on */
    const size_t n = (size_t)pic->n_ctbs;
    std::vector<uint32_t> order(n);
    {
      std::vector<uint32_t> cnt((size_t)ctbW + 2 * (size_t)ctbH + 2, 0);
      for (size_t c = 0; c < n; c++) cnt[(c % ctbW) + 2 * (c / ctbW) + 1]++;
      for (size_t i = 1; i < cnt.size(); i++) cnt[i] += cnt[i - 1];
      for (size_t c = 0; c < n; c++) order[cnt[(c % ctbW) + 2 * (c / ctbW)]++] = (uint32_t)c;
    }
    std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[n]);
    for (size_t c = 0; c < n; c++) done[c].store(0, std::memory_order_relaxed);
    std::atomic<size_t> next(0);
    const size_t parts = std::min<size_t>((size_t)std::max(1, host_threads()), std::max<size_t>(1, (size_t)std::min(ctbW, ctbH)));
    parallel_ranges(parts, 1, [&](size_t, size_t) {
      Scratch S;
      for (;;) {
        const size_t i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= n) break;
        const size_t c = order[i];
        const int cx = (int)(c % ctbW), cy = (int)(c / ctbW);
        const size_t dep[2] = {cx > 0 ? c - 1 : n, cy > 0 ? (cx + 1 < ctbW ? c - ctbW + 1 : c - ctbW) : n};
        for (int d = 0; d < 2; d++)
          if (dep[d] < n) while (!done[dep[d]].load(std::memory_order_acquire)) std::this_thread::yield();
        schedule_ctb(c, S);
        done[c].store(1, std::memory_order_release);
      }
      fold_stats(S);
    });
  }
  {
    const bool stats = level_stats;
    if (stats) fprintf(stderr, "intra_schedule: level makespans in block units, waves 8+2+2: %.0f, 8+4+4: %.0f, 12 shared: %.0f, 16 shared: %.0f\n", span[0], span[1], span[2], span[3]);
    if (stats) fprintf(stderr, "intra_schedule: %lld blocks in %lld CTBs, %lld levels (one-sided %d)\n", n_blocks.load(), n_intra_ctbs.load(), n_levels.load(), (int)one_sided);
  }
  *dense = (n_intra_ctbs.load() && n_blocks.load() >= 8 * (long)ctbW * ctbH) ? 1 : 0;
  return overlap.load();
}

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
/* canonical exchange-buffer layout of a picture (k_common.h HaloLayout); depends on the picture parameters only */
void halo_layout(const m355_pic_params& pp, HaloLayout& h) {
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

static void caps_of(const m355_picture* pic, m355_arena_caps& k)
{
  memset(&k, 0, sizeof(k));
  k.n_slices = pic->n_slices; k.n_ctbs = pic->n_ctbs; k.n_cus = pic->n_cus; k.n_tus = pic->n_tus; k.n_pbs = pic->n_pbs; k.n_wts = pic->n_wts;
  for (int b = 0; b < 4; b++) k.n_rbs[b] = pic->rb_count[b];
  k.n_ibs = pic->n_ibs; k.n_coeffs = pic->n_coeffs; k.n_pcm = pic->n_pcm; k.scaling = pic->scaling_factors != nullptr;
}
/* where everything of one picture sits in the (pinned host / device) arena, for given list capacities */
void make_layout(const m355_arena_caps& k, int nCtb, int halo_units, bool sharded, bool with_ib_input, Lay& L) {
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

int upload(m355_ctx* c, Resident& r, const m355_picture* pic) {
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
  int nw = 0, n_free = 0, n_ticket = 0;
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
    order.resize(n_cand);
    std::vector<uint32_t>& bucket = r.sched_bucket;
    if (!intra_dense) {
      /* INTER pictures (one workgroup per CTB with intra blocks, more of them than the GPU holds at once at 8K): the CTBs that take
         part in a dependency CHAIN — they read a neighbour's intra samples, or a neighbour reads theirs — come FIRST, by the length of
         the chain still hanging on them (rem = the CTB's own levels + a hand-off + the longest rem among its readers; readers follow
         their producers in decode order, so one pass backwards computes it): the stage ends with its longest chain, which therefore
         has to start with the launch, not when the CTBs without dependencies have drained (C5: the last chains were claimed 30 us
         into a 65 us launch, profiles/r05_g_intra_sparse_timeline.txt).  These go through the ticket, in this order: a producer's rem
         exceeds its readers', ties keep decode order, so a workgroup still only waits on lower tickets.  The CTBs without
         dependencies follow, by workgroup index (no ticket), longest first. */
      std::vector<uint32_t>& rem = r.sched_u32b;            /* (rebuilt as the plan bases below) */
      rem.assign((size_t)nCtb, 0);
      const uint32_t* const aux = (const uint32_t*)(r.host + seg[i_il].ofs);
      constexpr uint32_t REM_CAP = 4095, T_HANDOFF_LEVELS = 6;
      for (size_t i = n_cand; i-- > 0;) {
        const uint32_t rs = cand[2 * i];
        if (!(dep[rs] & 31)) continue;
        const m355_ctb& cb = pic->ctbs[rs];
        const uint32_t levels = ((aux[4 * (size_t)(cb.ib_start + cb.ib_count - 1) + 3] >> 16) & 0x3FFFu) + 1u;
        const uint32_t mine = std::min(REM_CAP, rem[rs] + levels + T_HANDOFF_LEVELS);
        rem[rs] = mine;
        const int cx = (int)rs % ctbW, cy = (int)rs / ctbW;
        const int dx[4] = {-1, -1, 0, 1}, dy[4] = {0, -1, -1, -1};
        for (int n = 0; n < 4; n++)
          if ((dep[rs] >> n) & 1) { uint32_t& pr = rem[(size_t)(cy + dy[n]) * ctbW + (cx + dx[n])]; pr = std::max(pr, mine); }
      }
      /* (a producer's entry held its readers' maximum until its own turn came: every entry of a chain member is final now) */
      bucket.assign((size_t)REM_CAP + 2 + (size_t)max_cnt + 2, 0);
      size_t n_chain = 0;
      auto slot_of = [&](size_t i) -> size_t {               /* chain members by falling rem, then the others by falling block count */
        const uint32_t rs = cand[2 * i];
        return (dep[rs] & 31) ? (size_t)(REM_CAP - rem[rs]) : (size_t)REM_CAP + 1 + (size_t)(max_cnt - (cand[2 * i + 1] & 0x7FFFFFFFu));
      };
      for (size_t i = 0; i < n_cand; i++) { bucket[slot_of(i) + 1]++; if (dep[cand[2 * i]] & 31) n_chain++; }
      for (size_t i = 1; i < bucket.size(); i++) bucket[i] += bucket[i - 1];
      for (size_t i = 0; i < n_cand; i++) order[bucket[slot_of(i)]++] = cand[2 * i];
      n_ticket = (int)n_chain;
    } else {
    n_free = (int)(n_cand - n_dep);
    n_ticket = (int)n_cand;                                  /* (an intra picture's persistent workgroups claim every item through the ticket) */
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
  d.n_intra_work = nw; d.n_intra_ticket = n_ticket;
  d.ctb_dep = (const uint8_t*)(r.dev + seg[i_dp].ofs);
  d.ctb_owner = sharded ? (const uint8_t*)(r.dev + seg[i_ow].ofs) : nullptr;
  d.halo_cu_base = pic->n_cus; d.halo_pb_base = pic->n_pbs;
  d.n_pb_records = pic->n_pbs + halo.n_units;
  d.device_validate = r.device_validate ? 1 : 0;
  d.n_wts = pic->n_wts; d.n_coeffs = pic->n_coeffs; d.n_pcm = pic->n_pcm; d.res_len = pic->res_len;
  r.used = true;
  return M355_OK;
}

} /* extern "C" */
