/*
 * runtime_decode.hip — host side of the picture layer, part 3 of 4: prepare() (scratch, DevPic), the per-picture launch sequence
 * (launch_prediction, decode_pre / decode_post), the status ring, m355_decode_batch.
 */
#include "runtime_internal.h"

extern "C" {
/* ----------------------------------------------------------------------- decode --------------- */

/* frames, scratch and the device descriptor of one decode of `r` */
bool chain_residuals_forced();

int prepare(m355_ctx* c, Resident& r, DevPic& d_out, bool& want_sao_out) {
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
  /* (Inter residuals are added to the prediction samples in place, behind k_inter_jobs.  Handing them to k_inter_jobs' write-back as int16
     tiles — round 4's "fused" order, the default with one picture in flight then — saves 80 MB per 8K picture and was measured again with
     round 5's lean k_inter_jobs: it loses at every depth, C5 0.362-0.368 against 0.346-0.356 ms with three pictures in flight, 0.486-0.494
     against 0.470-0.486 one at a time, C3 0.087 against 0.0805 — profiles/r05_h_fused_order_ab.txt — and left the product.) */
  /* behind the intra residuals: room for the residual tiles of every other block, which a dependent chain's picture leaves there in its front part
     (launch_prediction; with three lanes and more) — block i of size bin s at tile_base[s] + i * nT^2 */
  const size_t res_intra = ((size_t)pic.res_len + 64) & ~(size_t)63;
  size_t res_tiles = 0;
  for (int s = 0; s < 4; s++) { d.res_tile_base[s] = (uint32_t)res_tiles; res_tiles += (size_t)pic.rb_count[s] << (2 * (s + 2)); }
  /* (pictures of a one-stream lane only — up to 4K, launch_prediction: an 8K picture's kernels fill the GPU, its transforms gain nothing in front) */
  const bool tiles = (c->depth >= sched::chain_residual_tiles_min_depth || chain_residuals_forced()) && (long long)pic.pp.width * pic.pp.height <= sched::one_stream_max_samples;
  const size_t res_need = res_intra + (tiles ? res_tiles : 0) + 1;
  if ((rc = grow(&c->resbuf, &c->cap_res, res_need, c->stream, false))) return rc;
  d.res_tiles = tiles ? c->resbuf + res_intra : nullptr;     /* (resbuf itself: below, with the lane's other scratch) */
  d.res_front = 0;
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
  d.fill_pb_of_in_meta = ((c->stages & M355_STAGE_INTER) && m355_inter_uses_jobs(d)) ? 0 : 1;   /* else k_inter_jobs writes it */
  d.jobs = c->jobs; d.sao_nb = c->sao_nb; d.iplan = c->iplan;
  d.resbuf = c->resbuf; d.edge = c->edge; d.ticket = c->ticket; d.timeout = c->timeout;
  {
    /* intra pictures: k_intra's workgroups are persistent (k_intra.hip); with several pictures in flight every picture gets a
       share of the GPU's workgroup slots (2 per CU for this kernel) — enough for its active wavefront, not a slot per CTB */
    static const int grid_env = getenv("M355_INTRA_GRID") ? atoi(getenv("M355_INTRA_GRID")) : 0;
    static int slots = 0;
    if (!slots) { hipDeviceProp_t prop; slots = (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ? 2 * prop.multiProcessorCount : 512; }
    d.intra_grid = grid_env > 0 ? grid_env : std::max(64, slots / std::max(1, c->depth));
    /* the halo keeper (a 13th wave per workgroup, k_intra.hip) pays when the picture's own chain is all there is: with more pictures
       in flight their kernels fill the waits, and a 13-wave workgroup at 128 registers leaves no room on its CU for the 4-wave
       workgroups of the other pictures' kernels (C2, three in flight: 0.468 -> 0.536 ms per picture, profiles/r05_v14_*) */
    d.intra_keeper = c->depth <= sched::intra_keeper_max_depth;
    if (!d.intra_keeper && r.dp.intra_dense) {
      /* ... or when nothing else is in flight right now, whatever the depth: the intra picture a stream's other pictures wait for */
      bool idle = true;
      for (int i = 0; i < c->depth && idle; i++) if (i != c->active && ev_query(c, c->lanes[i].last) != hipSuccess) idle = false;
      d.intra_keeper = idle;
    }
    /* (test hook: the interpreter finishes every launch before the next call, so its pipeline is always idle — M355_TEST_NO_KEEPER=1
       sends intra pictures through the 12-wave kernel behind the planner's launch all the same) */
    static const bool no_keeper = getenv("M355_TEST_NO_KEEPER") != nullptr;
    if (no_keeper) d.intra_keeper = 0;
    /* (test hook: the interpreter runs k_intra's workgroups one after the other, so a neighbour's samples are always there when a
       CTB is staged — this sends them down the paths a CTB takes on the hardware, where they arrive later) */
    static const int halo_late = getenv("M355_TEST_HALO_LATE") ? atoi(getenv("M355_TEST_HALO_LATE")) : 0;
    d.test_halo_late = halo_late;
  }
  d.epoch = ++c->epoch;
  if (d.epoch == 0) d.epoch = ++c->epoch;
  d_out = d; want_sao_out = want_sao;
  return M355_OK;
}

hipError_t frame_event(hipEvent_t* e) {
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
void clear_target(m355_ctx* c, const DevPic& d, Frame* tgt, bool gated, hipStream_t st) {
  for (int cc = 0; cc < 3; cc++) {
    if (!tgt->pw[cc]) continue;
    const size_t bytes = (size_t)tgt->stride[cc] * tgt->ph[cc] * tgt->bpp[cc];
    if (gated) m355_launch_clear_gated(d, tgt->plane[cc], bytes, st);
    else hipMemsetAsync(tgt->plane[cc], 0, bytes, st);
  }
}

/* test hook (M355_TEST_CHAIN_RESIDUALS=1): every picture takes the residual order of a dependent chain's picture — k_residual in front of k_inter, storing tiles,
   k_residual_add behind it —, chain or not: the CPU tier's interpreter finishes every launch before the next call, so no reference is ever "still being written" there */
bool chain_residuals_forced() { static const bool on = getenv("M355_TEST_CHAIN_RESIDUALS") != nullptr; return on; }

void launch_prediction(m355_ctx* c, const Resident& r, const DevPic& d, bool hbd, hipEvent_t* ev, bool with_intra, hipStream_t chain, Frame* hazard_dst) {
  hipStream_t st = c->stream;
  /* an intra picture keeps to its lane's main stream: its side work (metadata planes, border plans: 0.07 ms) is nothing beside k_intra,
     and half as many streams compete for the runtime's hardware queues when many such pictures are in flight (C2 0.340 ms per picture
     = 1.50 M CTB64/s at depth 9, profiles/r03_v_*; forked: 0.59 at depth 8) — and so does a picture of up to 4K: the fork / join of the
     side stream is six packets (three event records, three waits) at about 2 us of pipeline time each, and what they buy — the metadata
     scatters and the second residual launch beside the main stream — is worth less than that once the kernels are short (three in
     flight, profiles/r04_al_*: C3 / C4 0.110 -> 0.098 / 0.100 ms on one stream, C5 0.347 -> 0.351); one intra picture at a time with its
     planes forked beside the residuals: 0.881 -> 0.896 ms (profiles/r05_v20_*) */
  const bool single = d.intra_dense || (long long)d.pp.width * d.pp.height <= sched::one_stream_max_samples;
  hipStream_t s2 = single ? st : c->stream2;
  /* the zero fill of the metadata planes rides in the picture's first main-stream launch (k_job_count), in FRONT of the fork: the
     side stream's scatters then start behind it — one launch less per inter picture */
  /* (the fill is shared out over the launch's workgroups, one per 256 PBs: with a handful of them a fill of its own is faster;
     M355_CLEAR_IN_COUNT_MIN=<PBs> moves the threshold: tests/test_meta_merged_emu.py sends the CPU tier's small pictures down this path —
     and through the merged planes + job-list launch behind it — with 1) */
  static const int clear_min = getenv("M355_CLEAR_IN_COUNT_MIN") ? atoi(getenv("M355_CLEAR_IN_COUNT_MIN")) : sched::clear_in_count_min_pbs;
  const bool clear_in_count = d.n_pbs >= std::max(1, clear_min);
  bool tu_plan_with_residuals = false;
  /* a dependent chain's picture (`chain`: decode_pre) transforms its residuals in its FRONT part — they do not depend on the reference —, as int16 tiles
     (k_residual with res_front), and adds them behind k_inter (k_residual_add: two round trips instead of the transform's chain of them): a chain's picture
     0.131 -> 0.124 ms at C3, 0.148 -> 0.140 at C4; at C5 0.42-0.44 -> 0.44-0.45 — pictures of a two-stream lane keep the one order (profiles/r05_v31_*).  Not at
     16 bits per sample: a tile holds a residual clipped to int16, which is the same sum only while a sample needs no more than 15 bits. */
  const bool res_front = (chain || chain_residuals_forced()) && d.res_tiles && (c->stages & M355_STAGE_RESIDUAL) && d.pp.bit_depth_luma <= 15 && d.pp.bit_depth_chroma <= 15 &&
                         d.rb_count[0] + d.rb_count[1] + d.rb_count[2] + d.rb_count[3] > 0;
  if (clear_in_count) m355_launch_job_count(d, true, st);
  if (!single) { hipEventRecord(c->ev_fork, st); hipStreamWaitEvent(s2, c->ev_fork, 0); }
  /* transform edges and border plans in ONE launch (a packet less per picture: C3 0.098 -> 0.093 ms, profiles/r05_a_switches_merge.txt) */
  if (single && clear_in_count && (c->stages & M355_STAGE_INTRA)) {
    /* one stream: the planes' scatters and the job list are independent roles of ONE launch (k_meta_planes_jobs) */
    m355_launch_meta_planes_jobs(d, st);
    /* (the transform edges + border plans — read by k_intra and the deblocking filter only — ride in the residual launch behind k_inter when the context
       decodes one picture at a time: C3 0.1494 -> 0.1427 ms, C4 0.1627 -> 0.1564.  With lanes they stay a launch of their own in FRONT of k_inter, where
       other pictures' kernels — or, for a chain's picture, its reference's last stages — run beside them: merged, C3 0.0688 -> 0.0697 ms with three in
       flight and a chain's picture 0.132 -> 0.136, C4 0.147 -> 0.156: profiles/r05_v30_*) */
    if (c->depth <= sched::tu_plan_in_residuals_max_depth && (c->stages & M355_STAGE_RESIDUAL) && !res_front) tu_plan_with_residuals = true; else m355_launch_tu_plan(d, st);
  } else {
    if (c->stages & M355_STAGE_INTRA) {
      m355_launch_meta_planes(d, s2, clear_in_count, false);
      m355_launch_tu_plan(d, s2);
    } else m355_launch_meta_planes(d, s2, clear_in_count);
    if (clear_in_count) m355_launch_job_list(d, st); else m355_launch_meta_jobs(d, st);
  }
  auto launch_residuals = [&](const DevPic& dd) {
    /* inter residuals are added to the prediction samples: behind k_inter_jobs; the two launches side by side on the lane's two
       streams (one after the other on the main stream, without the second fork, was measured 1 % slower at C5 with three
       pictures in flight: 0.3573-0.3605 against 0.3538-0.3580 ms, profiles/r04_am_residual_streams_ab.txt) */
    hipStream_t sr = !single ? s2 : st;
    if (sr != st) { hipEventRecord(c->ev_fork2, st); hipStreamWaitEvent(s2, c->ev_fork2, 0); }
    if (tu_plan_with_residuals) m355_launch_residual_tu_plan(dd, hbd, st);
    else if (sr == st) m355_launch_residual_both(dd, hbd, st);       /* (one stream: one launch, k_residual.hip) */
    else {
      m355_launch_residual(dd, hbd, false, sr);
      m355_launch_residual(dd, hbd, true, st);
    }
  };
  if (res_front) { DevPic dq = d; dq.res_front = 1; launch_residuals(dq); }
  if (ev) hipEventRecord(ev[1], st);
  /* read-after-write on the reference frames: their last writers are waited for HERE, in front of the first kernel that reads a
     reference — the list copy, validation, metadata planes and job list of a picture run beside the tail
     (filters) of the picture it references */
  if (chain && chain != st) {
    /* a picture of a dependent CHAIN (decode_pre: its newest reference is still being written, on `chain`): everything up to here — validation,
       metadata planes, job list, plans: what does not read a reference — ran on the lane's own stream, beside the reference's last stages; from
       here on the decode continues ON THE REFERENCE'S STREAM, where stream order stands for the wait.  The one cross-queue wait left is for this
       decode's own front part, which is long over when the reference's filters are (a wait on a mark that has passed costs a packet; a wait the
       queue really sleeps on cost a dependent 4K picture 25-40 us: profiles/r05_v25_*) */
    EvRef front;
    if (ev_mark(c, st, &front) == M355_OK) {
      c->stream = st = chain; ev_wait(c, st, front);
      if (ev) hipEventRecord(ev[1], st);      /* (timed decode: the 'inter' interval starts behind the hand-over, the wait for the reference's stages is booked on 'meta') */
    }
  }
  /* (a picture without SAO writes its destination from here on: its readers / last writer are waited for HERE, not in front of the metadata kernels —
     the destination of a chain's picture is often a frame the picture before it still reads) */
  if (hazard_dst) dst_hazards(c, hazard_dst, true);
  if (c->depth >= 2)
    for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
      Frame* f = r.hdr.ref_frames[i] >= 0 ? get_frame(c, r.hdr.ref_frames[i]) : nullptr;
      if (f) ev_wait(c, st, f->wr);
    }
  if (c->stages & M355_STAGE_INTER) m355_launch_inter(d, hbd, st);
  if (ev) hipEventRecord(ev[2], st);
  if ((c->stages & M355_STAGE_RESIDUAL) && !res_front) launch_residuals(d);
  else if (res_front) m355_launch_residual_add(d, hbd, st);
  if (!single) { hipEventRecord(c->ev_join, s2); hipStreamWaitEvent(st, c->ev_join, 0); }     /* join */
  if (ev) hipEventRecord(ev[3], st);
  if (with_intra && (c->stages & M355_STAGE_INTRA)) m355_launch_intra(d, hbd, st, clear_in_count);   /* (m355_decode_batch launches several pictures' intra stage as one kernel) */
  if (ev) hipEventRecord(ev[4], st);
}

/* write-after-write / write-after-read on the destination: waited for right before the first kernel that writes it — the SAO
   stage when SAO runs (everything before writes this lane's working planes), else the first stage */
void dst_hazards(m355_ctx* c, Frame* dstf, bool piped) {
  if (dstf->dl_pending) hipStreamWaitEvent(c->stream, dstf->ev_dl, 0);     /* (stays pending for the HOST until m355_frame_download_wait / m355_wait) */
  if (!piped) return;
  ev_wait(c, c->stream, dstf->wr);
  for (int k = 0; k < M355_MAX_LANES; k++) ev_wait(c, c->stream, dstf->rd[k]);
}

/* One decode = decode_pre (lane, hazards, validation, every stage in front of the intra stage [and, with_intra, that stage]) +
 * decode_post (in-loop filters, events, status slot).  m355_decode_batch runs the pre part of several intra pictures on their lanes,
 * ONE k_intra launch for all of them, then their post parts. */
struct DecodeState { DevPic d; bool want_sao = false; hipEvent_t* ev = nullptr; hipStream_t saved_stream = nullptr; bool swapped = false; hipStream_t chain = nullptr; };

/* front: PRE_ALL = everything up to and including the intra stage; PRE_NO_INTRA = without k_intra; PRE_HAZARDS = lane, hazards, validation and
   clearing only (m355_decode_batch launches the stages itself, one launch per stage for all its pictures) */
enum { PRE_ALL = 0, PRE_NO_INTRA = 1, PRE_HAZARDS = 2 };
static int decode_pre(m355_ctx* c, Resident& r, bool rotate, DecodeState& S, int mode, hipStream_t on_stream = nullptr)
{
  const bool with_intra = mode == PRE_ALL;
  if (r.sharded) return fail(M355_ERR_INVALID, "a sharded picture is decoded by phases (m355_decode_phase)");
  hipStream_t chain = nullptr;
  if (rotate && c->depth >= 2) {
    /* consecutive pictures go round the lanes.  A picture whose (newest) reference is still being written is the next link of a dependent CHAIN
       (low-delay P / B: every picture references the one before): its front part goes to a lane whose stream is NOT the reference's — it runs
       beside the reference's last stages —, everything from k_inter on to the reference's stream (launch_prediction).  On rotating lanes alone
       such a picture sleeps on a cross-queue event per picture: C3 0.182-0.200 ms per picture of a chain, 0.160 with the whole picture on its
       reference's lane, 0.139 with the front part beside the reference's tail; C4 0.184 / 0.180 / 0.155; C5 0.456-0.466 / 0.438 / 0.439 — an 8K picture's
       kernels fill the GPU, nothing runs beside them for free (profiles/r05_v25_*, r05_v27_*) */
    unsigned long long newest = 0;
    for (int i = 0; i < M355_MAX_REF_FRAMES; i++) {
      Frame* f = r.hdr.ref_frames[i] >= 0 ? get_frame(c, r.hdr.ref_frames[i]) : nullptr;
      /* (test hook M355_TEST_CHAIN_LANES=1: every reference this context has decoded counts as still being written — the interpreter finishes every launch
         before the next call, so the CPU tier would never walk the chain's bookkeeping: lane choice, change of stream, marks, late hazards) */
      static const bool always = getenv("M355_TEST_CHAIN_LANES") != nullptr;
      if (!f || f->wr.ticket <= newest || (!always && ev_query(c, f->wr) == hipSuccess)) continue;
      newest = f->wr.ticket; chain = f->wr.stream;
    }
    int lane = (c->active + 1) % c->depth;
    if (chain) {
      /* three lanes and more: the front part alternates between the lanes that are not the chain's (two scratch sets: the one a picture's back part
         still uses and the one the next front part fills); two lanes: the whole picture on the chain's own lane — with one lane to spare the front
         part would wait for the scratch of the picture before it and bring the cross-queue waits back (C3 0.162 -> 0.178 ms, profiles/r05_v27_*) */
      const bool split = c->depth >= sched::chain_split_min_depth;
      for (int k = 0; k < c->depth; k++) {
        const int l = (c->active + 1 + k) % c->depth;
        if (((l == c->active ? c->stream : c->lanes[l].stream) != chain) == split) { lane = l; break; }
      }
    }
    select_lane(c, lane);
  }
  /* which stream: an intra picture on lane 3.. takes the lane's class stream (lane_class_priority); the whole decode addresses
     c->stream, which is that stream until decode_post returns (a batch keeps to the lanes' ordinary streams: its pictures overlap
     inside one kernel, not through hardware queues) */
  {
    hipStream_t run = on_stream ? on_stream : c->stream;   /* (a batch on a stream of its own: its lanes lend their scratch only) */
    if (!on_stream && with_intra && r.dp.intra_dense && c->active >= sched::intra_class_first_lane && lane_priorities_mode() == 2 && lane_class_priority(c->active) != 0) {
      if (!c->stream_hi) HIPCHK(hipStreamCreateWithPriority(&c->stream_hi, hipStreamNonBlocking, lane_class_priority(c->active)));
      run = c->stream_hi;
    }
    ev_wait(c, run, c->last);                              /* the lane's scratch and working planes (when its last decode ran on its other stream) */
    S.saved_stream = c->stream; S.swapped = run != c->stream;
    c->stream = run;
    if (chain && chain != run && mode == PRE_ALL) { S.chain = chain; S.swapped = true; }   /* (launch_prediction moves the decode onto `chain`; decode_post restores) */
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
    c->ev_used++;
    hipEventRecord(ev[0], st);
  }
  S.ev = ev;
  if (r.device_validate) m355_launch_validate(d, st);     /* a rejection gates THIS decode's kernels (epoch-tagged gate word) */
  /* write-after-read / -write on the destination of a picture without SAO: in front of its first writer — the clearing fill if there is one, else
     k_inter (launch_prediction, behind a chain picture's change of stream) */
  const bool hazards_late = !want_sao && piped && mode != PRE_HAZARDS && !(pp.flags & M355_PF_CLEAR_DST);
  if (!want_sao && !hazards_late) dst_hazards(c, dstf, piped);
  if (pp.flags & M355_PF_CLEAR_DST) clear_target(c, d, want_sao ? &c->work : dstf, r.device_validate && !want_sao, st);
  if (!with_intra) d.intra_keeper = 0;                     /* (a batch's shared intra stage is the 12-wave kernel: it needs the planner's launch) */
  if (mode != PRE_HAZARDS) launch_prediction(c, r, d, hbd, ev, with_intra, S.chain, hazards_late ? dstf : nullptr);
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

int decode(m355_ctx* c, Resident& r, bool rotate) {
  DecodeState S;
  int rc = decode_pre(c, r, rotate, S, PRE_ALL);
  if (rc) { if (S.swapped) c->stream = S.saved_stream; return rc; }
  return decode_post(c, r, S);
}

/* status of one finished decode from its ring slot: M355_OK, or M355_ERR_INVALID with the rejected record in the message */
int status_of(m355_ctx* c, m355_ctx::Status& s) {
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
  if (s.serial != serial) return fail(M355_ERR_STALE, "decode %llu is older than the last %d decodes: its status is no longer kept (m355_wait reports rejections)", serial, M355_STATUS_RING);
  hipSetDevice(c->device);
  const hipError_t q = ev_query(c, s.done);
  if (q == hipErrorNotReady) return M355_ERR_BUSY;
  if (q != hipSuccess) return fail(M355_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
  return status_of(c, s);
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
} /* extern "C" */
