/*
 * k_common.h — device-side picture descriptor, small device helpers and the launch entry points of
 * the MI355X HEVC reconstruction kernels (one .hip file per stage).
 *
 * Data layout in HBM:
 *   frames      : planar, one allocation per plane, element = uint8 (8 bit) / uint16 (9..16 bit),
 *                 row pitch in SAMPLES padded to a multiple of 128 bytes;
 *   work lists  : the PODs of include/de265_mi355x.h, uploaded verbatim (SoA per list);
 *   metadata    : per-min-CB CU index plane, per-4x4 transform/prediction edge bytes and PB index
 *                 plane — rasterised on the device from the CU / TU-leaf / PB lists (k_meta.hip);
 *   residuals   : int16, one contiguous nT*nT tile per deferred (intra) block.
 */
#ifndef M355_K_COMMON_H
#define M355_K_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "de265_mi355x.h"
#include <k_asm.h> /* resolved through the include path (csrc/ for the product build) */

struct DevRef {            /* one reference frame (indexed by m355_pb.ref_slot) */
  const void* plane[3];
  int stride[3];
  int valid;
  int pad;
};

/* k_intra's work item: everything a workgroup needs to know about its CTB in ONE 32-byte record (one scalar load behind the
 * ticket instead of a chain of dependent lookups: work list -> CTB record -> neighbours' slices / tiles / scan positions) */
struct DevIntraWork {
  uint32_t ctb;                     /* raster address */
  uint32_t ib_start, ib_count;      /* the CTB's intra blocks (sorted by dependency level) */
  uint16_t nb_same;                 /* bit k: 3x3 neighbour k (row-major, 4 = the CTB itself) lies in the picture, in the same slice
                                       (SliceAddrRS) and in the same tile */
  uint16_t nb_earlier;              /* bit k: neighbour k precedes the CTB in decode (tile-scan) order */
  uint8_t waves_code;               /* widest dependency level of the CTB (runtime_upload.hip intra_schedule): 0..3 */
  uint8_t pad[3];
  uint32_t plan_base;               /* first entry of the CTB's border plans in DevPic.iplan (a multiple of 8 entries) */
  uint32_t plan_count;              /* ... and how many entries they are (<= M355_INTRA_PLAN_CAP) */
  uint32_t reserved;
};
/* Border plans (k_intra_plan -> k_intra): per intra block 4nT + 1 16-bit entries (one LDS source per border entry).
 * The blocks of one component of a CTB are disjoint (runtime_upload.hip intra_schedule rejects lists where they are not), which bounds a
 * CTB's plans by its 4x4-only case, 18 entries per 4x4 block, ... */
/* DevPic.ib_aux: one EXEC RECORD of four words per intra block (runtime_upload.hip intra_schedule), everything k_intra's block chain needs
 * that is not a sample value:
 *   word 0  bits 0-6 lx, 7-13 ly (position inside the CTB, component samples), 14-16 log2 size, 17-18 component, 19-24 mode, flags:
 *   word 1  residual buffer offset (int16 units) / pcm[] offset
 *   word 2  bits 0-7 intraPredAngle (signed), 8-10 mode class (0 planar, 1 DC, 2 pure horizontal / vertical, 3 / 4 angular with a
 *           positive / negative angle), 16-31 invAngle (signed; 0 unless the angle is negative)   (intrapred.h:313-326)
 *   word 3  bits 0-15 offset of the block's border plan inside the CTB's plans, 16-29 dependency level inside the CTB */
/* blocks of this size and larger are SHARED by their component's waves in k_intra (each gathers the border itself and predicts its rows);
   intra_schedule gives a CTB that holds one all the waves there are: 32x32 only (16x16 too: C2 0.82 -> 0.87 ms, profiles/r05_v17_*) */
#ifndef M355_INTRA_SHARE_MIN_LOG2
#define M355_INTRA_SHARE_MIN_LOG2 5
#endif
#define M355_IBX_HAS_RES 0x02000000u
#define M355_IBX_PCM     0x04000000u
#define M355_IBX_FILT    0x08000000u   /* [1 2 1] border smoothing applies (intrapred.h:195-212: mode, size and component decide) */
#define M355_IBX_STRONG  0x10000000u   /* ... and the bilinear variant is allowed if the border is flat (intrapred.h:216-234) */
#define M355_IBX_BFILT   0x20000000u   /* luma block < 32x32 whose boundary smoothing applies (DC: always; pure horizontal / vertical: unless disabled) */
#define M355_IBX_PUB_COL 0x40000000u   /* the block completes a piece of the CTB's right column that the next CTB may read */
#define M355_IBX_PUB_ROW 0x80000000u   /* ... of its bottom row */
#define M355_INTRA_PLAN_CAP(cf) ((cf) == 0 ? 4608 : ((cf) == 1 ? 6912 : ((cf) == 2 ? 9216 : 13824)))
/* ... and the plans of any 64 blocks of a CTB by 2176 / 2496 / 2848 / 3520 entries (+ 7 of alignment) */
#define M355_INTRA_PLAN_BATCH(cf) ((cf) == 0 ? 2304 : ((cf) == 1 ? 2560 : ((cf) == 2 ? 3072 : 3584)))

/* m355_pic_params as the KERNELS see it: every field a whole dword.  A DevPic travels in the kernel argument segment, and a field
 * narrower than a dword there is not a scalar load (gfx950 has none below 32 bits): hipcc fetches it with a VECTOR memory load from
 * the argument segment and waits for everything in flight where it needs it — k_deblock started with such a round trip before its
 * first real load, k_sao had one behind its sample loads (bit depth of the launch's component).  Same field names, so the kernels
 * read p.pp.<field> as before; dev_pic_params() widens. */
struct DevPicParams {
  int32_t  width, height;
  int32_t  chroma_format_idc, bit_depth_luma, bit_depth_chroma, log2_ctb_size, log2_min_tb_size, log2_min_cb_size;
  int32_t  pic_cb_qp_offset, pic_cr_qp_offset;
  uint32_t flags;
  int32_t  num_tile_cols, num_tile_rows;
  uint16_t col_bd[M355_MAX_TILE_COLS + 1];
  uint16_t row_bd[M355_MAX_TILE_ROWS + 1];
};
static inline DevPicParams dev_pic_params(const m355_pic_params& q)
{
  DevPicParams d;
  d.width = q.width; d.height = q.height;
  d.chroma_format_idc = q.chroma_format_idc; d.bit_depth_luma = q.bit_depth_luma; d.bit_depth_chroma = q.bit_depth_chroma;
  d.log2_ctb_size = q.log2_ctb_size; d.log2_min_tb_size = q.log2_min_tb_size; d.log2_min_cb_size = q.log2_min_cb_size;
  d.pic_cb_qp_offset = q.pic_cb_qp_offset; d.pic_cr_qp_offset = q.pic_cr_qp_offset;
  d.flags = q.flags; d.num_tile_cols = q.num_tile_cols; d.num_tile_rows = q.num_tile_rows;
  for (int i = 0; i <= M355_MAX_TILE_COLS; i++) d.col_bd[i] = q.col_bd[i];
  for (int i = 0; i <= M355_MAX_TILE_ROWS; i++) d.row_bd[i] = q.row_bd[i];
  return d;
}

struct DevPic {
  DevPicParams pp;
  int sw, sh;                       /* SubWidthC, SubHeightC */
  int ctbW, ctbH, nCtb, w4, h4, wcb, hcb;
  int pw[3], ph[3];                 /* plane dimensions */
  void* plane[3];                   /* reconstruction / deblocking target */
  int stride[3];
  void* out_plane[3];               /* SAO output (the DPB frame) */
  int out_stride[3];
  const DevRef* refs;               /* M355_MAX_REF_FRAMES entries in device memory */
  const uint32_t* inter_tabs;       /* tap tables of k_inter_jobs' lean filters for this picture's plane type and bit depths (k_inter.hip m355_inter_tables) */
  /* work lists (device copies) */
  const m355_slice* slices;
  const m355_ctb* ctbs;
  const m355_cu* cus;
  const m355_tu* tus;
  const m355_pb* pbs;
  const m355_wt* wts;
  const m355_rb* rb_bin[4];         /* residual blocks of 4x4, 8x8, 16x16, 32x32 (rb_count[] entries each) */
  const m355_ib* ibs;               /* device copy: each CTB's blocks sorted by dependency level (runtime_upload.hip intra_schedule) */
  const uint32_t* ib_aux;           /* per ibs[i]: the block's exec record, 4 words (M355_IBX_*) */
  uint16_t* iplan;                  /* border plans of all intra blocks (k_intra_plan writes, k_intra reads), lane scratch */
  int intra_dense;                  /* k_intra variant: 1 = intra picture (12-wave workgroups, residuals in LDS), 0 = a handful of blocks per CTB */
  const uint32_t* coeffs;
  const uint16_t* pcm;
  const uint8_t* scaling;
  int n_cus, n_tus, n_pbs, n_ibs;
  int rb_count[4];
  /* derived tables / metadata planes */
  const uint32_t* ctb_ts;           /* CtbAddrRStoTS */
  const uint32_t* ts2rs;            /* CtbAddrTStoRS */
  const uint16_t* tile_id;          /* TileIdRS */
  uint32_t* cb_cu;                  /* per min CB: CU index + 1 */
  uint8_t* cuf;                     /* per CU: bit0 filterLeftCbEdge, bit1 filterTopCbEdge, bit2 deblock on */
  uint8_t* edge_tu;                 /* per 4x4: bit0 TU edge V, bit1 TU edge H, bit4 cbf_luma */
  uint8_t* edge_pb;                 /* per 4x4: bit2 PB edge V, bit3 PB edge H */
  uint32_t* pb_of;                  /* per 4x4: PB index + 1 */
  int16_t* resbuf;
  /* a dependent chain's picture (runtime_decode.hip): k_residual runs in the picture's FRONT part, beside the reference's last stages, and leaves the residuals of
     the blocks it would add to the prediction as int16 tiles — block i of size bin s at res_tiles + res_tile_base[s] + i * nT^2 —; k_residual_add adds them behind k_inter */
  int16_t* res_tiles;
  uint32_t res_tile_base[4];
  int res_front;                    /* 1: this launch of k_residual stores tiles instead of adding */
  uint16_t* sao_nb;                 /* [component][CTB]: bit (dy+1)*3+(dx+1) set = SAO edge neighbours in that CTB are not usable; bit 15 = the CTB's slice has SAO on for the component (k_meta_sao) */
  uint32_t* jobs;                   /* inter jobs: pb index | strip << 25 | row block << 29 (k_meta_pb) */
  uint32_t* job_base;               /* [256-PB chunk][4]: the chunk's jobs per range (uni, bi, weighted, edge): k_job_count leaves the counts here, every
                                       workgroup of k_meta_pb sums the ones in front of its chunk (lane scratch) */
  uint32_t* job_tot;                /* range ENDS: [0] one-list jobs, [1] + bi-predicted, [2] + explicitly weighted = where the EDGE range starts, [3] all jobs (k_meta_pb's workgroup 0; clamped to jobs_cap) */
  uint32_t jobs_cap;                /* entries of jobs[]: >= what any list of disjoint prediction blocks produces (runtime.hip prepare) */
  int fill_pb_of_in_meta;           /* 1: k_meta_pb fills pb_of (inter stage off); 0: k_inter_jobs does, two units per job */
  /* intra wavefront state */
  unsigned long long* edge;         /* k_intra's halo exchange: 8-byte granules (epoch << 32 | two samples), per component the right
                                       columns of all CTB columns [ctbX][row >> 1], then the bottom rows of all CTB rows [ctbY][col >> 1] */
  uint32_t edge_col_ofs[3], edge_row_ofs[3];   /* first granule of each component's column / row arrays */
  uint32_t* ticket;                 /* work counter */
  uint32_t* timeout;                /* [0] set when a spin bound is exceeded; [1] epoch of the last decode of this lane whose lists
                                       k_validate rejected (every kernel of THAT decode returns at once: M355_GATE); [2..3] one 64-bit
                                       word, ~epoch << 32 | lowest rejected (list << 28 | record) of that decode (atomicMin: a later
                                       rejected decode replaces an earlier one, nothing is ever reset); copied to the host's status
                                       ring at the end of every device-validated decode (runtime.hip) */
  int device_validate;              /* lists recorded in place: checked on the device (k_validate) instead of on the host */
  int n_wts;
  uint32_t n_coeffs, n_pcm, res_len, ref_valid;   /* list lengths the records index into; bit s of ref_valid = ref_frames[s] is a frame */
  uint32_t epoch;                   /* value meaning "done" for this submission */
  const DevIntraWork* intra_work;   /* one descriptor per CTB that holds intra blocks (runtime_upload.hip).  Intra pictures: the CTBs that wait for no
                                       neighbour (longest first), then the dependent ones in wavefront order, all claimed through k_intra's ticket.
                                       Inter pictures: first the n_intra_ticket CTBs of dependency chains, longest remaining chain first (ticket
                                       order: a producer precedes its readers), then the CTBs without dependencies (by workgroup index) */
  int n_intra_work, n_intra_ticket;
  int intra_grid;                   /* intra pictures: workgroups of k_intra's launch (persistent: each takes CTB after CTB); 0 = one per CTB */
  int intra_keeper;                 /* intra pictures: launch k_intra with its halo keeper wave (one picture at a time; k_intra.hip) */
  int test_halo_late;               /* test hook (M355_TEST_HALO_LATE=1, tests/test_intra_halo_late.py): k_intra's prologue takes no neighbour's
                                       sample from its granule — every one is fetched behind the prologue, by the halo keeper or the block's own poll */
  const uint8_t* ctb_dep;           /* per CTB: bit n = reads intra output of neighbour n (0 L, 1 TL, 2 T, 3 TR; orders the work list);
                                       bit 4 = a neighbour reads ours; bits 5-6 = the CTB's widest level (0: 1 luma block, 1: 2, 2: 3-4, 3: more) -> waves in k_intra */
  /* tile sharding (k_shard.hip): NULL = this context owns the whole picture */
  const uint8_t* ctb_owner;         /* per CTB (raster): 1 = a tile of this rank */
  int halo_cu_base, halo_pb_base;   /* first entry of the foreign border records appended to cus[] / pbs[] */
  int n_pb_records;                 /* n_pbs + the foreign border records: the highest PB index + 1 that pb_of may legitimately hold */
};

/* Canonical (rank-independent) layout of the tile-boundary exchange buffers (de265_mi355x.h, "Tile
 * sharding"): per component the column strips of all interior vertical boundaries, then the row strips
 * of all interior horizontal boundaries.  Column strip element (b, y, k): sample (xb_b - hw + k, y);
 * row strip element (b, k, x): sample (x, yb_b - hh + k). */
struct HaloLayout {
  int n_vb, n_hb;                   /* interior vertical / horizontal tile boundaries */
  int hw[3], hh[3];                 /* half strip width / height per component (0: component absent) */
  int xb[3][M355_MAX_TILE_COLS];    /* boundary positions in component samples */
  int yb[3][M355_MAX_TILE_ROWS];
  int col_ofs[4], row_ofs[4];       /* first sample of each component's segment (entry 3 = total) */
  int n_units;                      /* border 4x4 units: 2*n_vb*h4 + 2*n_hb*w4 */
};

/* Does any reference window of this PB leave the picture (so its jobs need the coordinate-clamped loads of motion.cc:141-159)?
 * Luma span of a 4-column job: [x-3, x+7] (+1 pad sample), chroma span of its 2 columns: [xc-1, xc+3]; a job is 8 rows high whatever
 * the PB's height, so the vertical span is that of the height rounded up to 8: rows [y-3, y+h8+3] / [yc-1, yc+h8/2+1].  The jobs of
 * every other PB are filtered without a clamp (k_inter.hip, lean filters). */
__host__ __device__ inline bool m355_pb_is_edge(const m355_pb& pb, int width, int height, int chroma_format_idc)
{
  const int h8 = (pb.h + 7) & ~7;
  for (int l = 0; l < 2; l++) {
    if (!(pb.flags & (M355_PBF_MC_L0 << l)) || (pb.flags & (M355_PBF_FILL_L0 << l))) continue;
    const int xl = pb.x + (pb.mv[l][0] >> 2), yl = pb.y + (pb.mv[l][1] >> 2);
    if (xl - 3 < 0 || xl + pb.w + 3 > width - 1) return true;
    if (yl - 3 < 0 || yl + h8 + 3 > height - 1) return true;
    if (chroma_format_idc == 1) {
      const int xc = (pb.x >> 1) + (pb.mv[l][0] >> 3), yc = (pb.y >> 1) + (pb.mv[l][1] >> 3);
      if (xc - 1 < 0 || xc + (pb.w >> 1) + 1 > (width >> 1) - 1) return true;
      if (yc - 1 < 0 || yc + (h8 >> 1) + 1 > (height >> 1) - 1) return true;
    }
  }
  return false;
}

/* Which border entries does an intra block's prediction READ?  (k_intra.hip: the planner points every other entry at the constant
 * cell, so that no block waits for a neighbour CTB's sample it will not use; runtime_upload.hip intra_schedule: which blocks of other
 * CTBs a block waits for.)  Entries i = 1 .. *top_e of the row above and -1 .. -*left_e of the column on the left (entry 0, the
 * corner, is always kept), the +-1 reach of the [1 2 1] smoothing (intrapred.h:185-258) included:
 *   planar (intrapred.h:261-285): nT + 1 on both sides | DC (:288-310, 378-392): nT | 11..25, negative angles (:330-376): the
 *   projection (x * invAngle + 128) >> 8 stays inside nT on the other side | 10 / 26: nT on their own side, the other one only for
 *   the boundary filter (:378-433) | 27..34: the row above up to nT + 1 + ((nT * intraPredAngle) >> 5) (index x + iIdx + 2 of the
 *   last sample's second tap), nothing of the column | 2..9: the mirror image.
 * 32x32 luma under strong smoothing reads both ends of both sides (:196-215): everything. */
__host__ __device__ inline void m355_intra_used_entries(int mode, int log2, int cidx, int chroma_format_idc, uint32_t pic_flags, uint32_t ib_flags, int* top_e, int* left_e)
{
  const int nT = 1 << log2;
  *top_e = 2 * nT; *left_e = 2 * nT;
  if (cidx == 0 && log2 == 5 && (pic_flags & M355_PF_STRONG_INTRA_SMOOTHING)) return;
  bool filt = false;
  if (!(pic_flags & M355_PF_INTRA_SMOOTHING_DISABLED) && (cidx == 0 || chroma_format_idc == 3) && mode != 1 && log2 != 2) {
    const int d26 = mode > 26 ? mode - 26 : 26 - mode, d10 = mode > 10 ? mode - 10 : 10 - mode, minDist = d26 < d10 ? d26 : d10;
    filt = log2 == 3 ? minDist > 7 : (log2 == 4 ? minDist > 1 : (log2 == 5 ? minDist > 0 : false));
  }
  const bool bf = cidx == 0 && log2 < 5 && (mode == 1 || !(ib_flags & M355_IBF_DISABLE_BOUNDARY_FILTER));
  const int mag[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
  int te, le;
  if (mode == 0) { te = nT + 1; le = nT + 1; }
  else if (mode == 1) { te = nT; le = nT; }
  else if (mode > 10 && mode < 26) { te = nT; le = nT; }
  else if (mode == 26) { te = nT; le = bf ? nT : 0; }
  else if (mode == 10) { le = nT; te = bf ? nT : 0; }
  else if (mode > 26 && mode <= 34) { te = nT + 1 + ((nT * mag[mode - 26]) >> 5); le = 0; }
  else if (mode >= 2 && mode < 10) { le = nT + 1 + ((nT * mag[10 - mode]) >> 5); te = 0; }
  else return;                                             /* (not a mode: rejected elsewhere) */
  if (filt) { if (te) te++; if (le) le++; }
  *top_e = te < 2 * nT ? te : 2 * nT; *left_e = le < 2 * nT ? le : 2 * nT;
}

/* rectangles of one k_tiles_copy launch (finished tiles <-> all-gather buffer); wb / xb in bytes, ofs = byte offset in the buffer */
struct TileCopyRect { uint32_t plane, xb, y, wb, h, pad; uint64_t ofs; };
/* which inter kernel a picture takes: the job kernels (k_inter_jobs: one lane per 4 x 8 block, lean filters exact for bit depths <= 12, rows fetched
   with 12- / 16-sample vector loads) for 4:2:0 / monochrome pictures at least one such vector wide; else k_inter_generic (one wavefront per PB) */
static inline bool m355_inter_uses_jobs(const DevPic& p)
{
  const int bdmax = p.pp.bit_depth_luma > p.pp.bit_depth_chroma ? p.pp.bit_depth_luma : p.pp.bit_depth_chroma;
  return p.pp.chroma_format_idc <= 1 && bdmax <= 12 && p.pw[0] >= 16 && (p.pp.chroma_format_idc == 0 || p.pw[1] >= 8);
}

#define M355_TILE_COPY_RECTS 48
struct TileCopyArgs { char* plane[3]; size_t pitch[3]; TileCopyRect r[M355_TILE_COPY_RECTS]; };
void m355_launch_tiles_copy(const TileCopyArgs& a, int n, void* xbuf, bool to_slot, hipStream_t st);

/* first statement of every kernel of a decode: a picture whose lists k_validate rejected is never acted upon */
/* Element `c` (0..2, per lane) of a three-entry table of the kernel arguments (plane pointers, pitches, ...): all three entries are
 * read as scalars and the lane selects — indexing the argument segment with a per-lane value is a VECTOR memory load from it, i.e. one
 * more dependent round trip between a record and the loads its component decides (k_residual: record -> plane pointer -> row).  The
 * pins keep hipcc from folding the selection back into an address. */
template <class T> __device__ __forceinline__ T d_sel3(int c, T a0, T a1, T a2)
{
  M355_PIN_V(a0); M355_PIN_V(a1); M355_PIN_V(a2);
  return c == 0 ? a0 : (c == 1 ? a1 : a2);
}
#define M355_SEL3(arr, c) d_sel3((c), (arr)[0], (arr)[1], (arr)[2])

#define M355_GATE(p) do { if ((p).timeout[1] == (p).epoch) return; } while (0)

enum { E_TU_V = 1, E_TU_H = 2, E_PB_V = 4, E_PB_H = 8, E_NONZERO = 16 };

/* Batch launches (m355_decode_batch): one launch of a stage over SEVERAL pictures — a grid plane (blockIdx.z) per picture, the
 * picture records in device memory, `on` = the pictures that run this stage.  Every such stage is a body function over
 * `const DevPic&` with two kernels around it: k_x(DevPic) for one picture (record in the kernel arguments) and k_x_batch(DevBatch). */
struct DevBatch { const DevPic* pics; uint32_t on; };
#define M355_BATCH_PIC_AT(b, k) if (!(((b).on >> (k)) & 1u)) return; const DevPic& p = (b).pics[(k)]
#define M355_BATCH_PIC(b) M355_BATCH_PIC_AT(b, blockIdx.z)
/* what the host needs for a batched launch: the pictures' own records (grid sizes), their device copies, how many, who takes part */
struct HostBatch { const DevPic* host; const DevPic* dev; int n; uint32_t on; };

/* ---- launchers (each in its stage's .hip); all asynchronous on `st` ---- */
void m355_launch_validate(const DevPic& p, hipStream_t st);   /* device-side validation of the work lists (k_meta.hip) */
void m355_launch_meta(const DevPic& p, hipStream_t st);
void m355_launch_clear_gated(const DevPic& p, void* ptr, size_t bytes, hipStream_t st);   /* zero fill behind the decode's gate (bytes: a multiple of 16) */
void m355_launch_meta_jobs(const DevPic& p, hipStream_t st);     /* job list for k_inter (= the two below) */
void m355_launch_job_count(const DevPic& p, bool clear_planes, hipStream_t st);   /* ... its first launch, optionally with the zero fill of the metadata planes */
void m355_launch_job_list(const DevPic& p, hipStream_t st);      /* ... the rest */
void m355_launch_meta_planes(const DevPic& p, hipStream_t st, bool cleared, bool with_tu = true);   /* planes for intra / deblock / SAO (cleared: k_job_count filled them; !with_tu: the transform edges come with m355_launch_tu_plan) */
void m355_launch_meta_planes_jobs(const DevPic& p, hipStream_t st);   /* CU plane + PB edges, SAO masks and the job list as roles of ONE launch (behind m355_launch_job_count(clear)) */
void m355_launch_tu_plan(const DevPic& p, hipStream_t st);         /* transform edges + border plans in ONE launch */
void m355_launch_inter(const DevPic& p, bool hbd, hipStream_t st);
#define M355_INTER_TAB_WORDS 300
void m355_inter_tables(bool bytes, int bd_luma, int bd_chroma, uint32_t* out);   /* host: the tables behind DevPic.inter_tabs (M355_INTER_TAB_WORDS words) */
void m355_launch_residual(const DevPic& p, bool hbd, bool big, hipStream_t st);   /* big: 32x32 + 16x16 blocks, else 8x8 + 4x4 */
void m355_launch_residual_both(const DevPic& p, bool hbd, hipStream_t st);      /* both size classes as roles of one launch (one-stream lanes) */
void m355_launch_residual_add(const DevPic& p, bool hbd, hipStream_t st);       /* the tiles a res_front launch left, added to the prediction */
void m355_launch_residual_tu_plan(const DevPic& p, bool hbd, hipStream_t st);   /* ... with the transform edges and the border plans (m355_launch_tu_plan's work) */
void m355_launch_intra_plan(const DevPic& p, hipStream_t st);   /* border plans of the intra blocks (k_intra.hip): before m355_launch_intra */
void m355_launch_intra(const DevPic& p, bool hbd, hipStream_t st, bool ticket_zero = false);
void m355_launch_meta_planes_batch(const HostBatch& b, hipStream_t st);              /* the batch forms: pictures of one sample type and chroma format */
void m355_launch_residual_batch(const HostBatch& b, bool hbd, bool big, hipStream_t st);
void m355_launch_intra_plan_batch(const HostBatch& b, hipStream_t st);
void m355_launch_deblock_batch(const HostBatch& b, bool hbd, hipStream_t st);
void m355_launch_sao_batch(const HostBatch& b, bool hbd, hipStream_t st);
void m355_launch_intra_batch(const DevPic& first, bool hbd, const DevPic* dev_pics, int n, int max_work, uint32_t* ticket, int grid, hipStream_t st);   /* intra pictures of one geometry in ONE launch */
void m355_launch_deblock(const DevPic& p, bool hbd, hipStream_t st);
void m355_launch_deblock_pass(const DevPic& p, bool hbd, bool vertical, hipStream_t st);   /* one direction (tile sharding) */
void m355_launch_sao(const DevPic& p, bool hbd, hipStream_t st);
/* tile sharding: `which` bit 0 = column strips, bit 1 = row strips; meta = border-unit records (16 B each) */
void m355_launch_halo_pack(const DevPic& p, const HaloLayout& h, bool hbd, int which, void* samples, uint32_t* meta, hipStream_t st);
/* buf[i] += sum over the n received copies scratch[k * pitch_words + i] (halo exchange of m355_decode_sharded over RCCL) */
void m355_launch_halo_add(uint32_t* buf, const uint32_t* scratch, uint32_t pitch_words, int n, uint32_t words, hipStream_t st);
void m355_launch_halo_unpack(const DevPic& p, const HaloLayout& h, bool hbd, int which, const void* samples, const uint32_t* meta, hipStream_t st);

/* ---- device helpers ---- */
/* lo <= hi at every call site: min(max()) lets hipcc emit v_med3_i32 / v_max+v_min instead of compare+select chains */
__device__ __forceinline__ int d_clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
__device__ __forceinline__ int d_clip_bd(int v, int bd) { return d_clip3(0, (1 << bd) - 1, v); }
__device__ __forceinline__ int d_abs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int d_sign(int v) { return (v > 0) - (v < 0); }

/* lanes of one wave exchanging data through LDS: order the accesses (hardware executes a wave's DS
 * operations in order; the fences keep the compiler from moving them) */
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int d_ctb_of(const DevPic& p, int xl, int yl)
{
  return (yl >> p.pp.log2_ctb_size) * p.ctbW + (xl >> p.pp.log2_ctb_size);
}
__device__ __forceinline__ const m355_slice& d_slice_at(const DevPic& p, int xl, int yl)
{
  return p.slices[p.ctbs[d_ctb_of(p, xl, yl)].slice_idx];
}
__device__ __forceinline__ uint32_t d_cu_index_at(const DevPic& p, int xl, int yl)
{
  return p.cb_cu[(yl >> p.pp.log2_min_cb_size) * p.wcb + (xl >> p.pp.log2_min_cb_size)];
}

#endif
