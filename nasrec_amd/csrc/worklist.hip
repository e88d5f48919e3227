// NASREC_OP_WORKLIST (include/nasrec_hip.h): one launch = the operators of one LEVEL of a batch-256 step (nasrec_amd/schedule.py),
// side by side on disjoint workgroup ranges.  Every item runs the body of its stand-alone kernel — the run-time-binding GEMM tile
// (gemm_rt.h), the split-K second pass, the Transformer forward / backward, the FM / DotProduct cores, segmented copies, the gating
// backward, fixed-order row reductions — so results are bit-identical to the separate launches; what disappears is a kernel
// boundary (~5 us of cold-L2 start at batch 256) per operator that has an independent neighbour.
//
// One LDS buffer is shared by all bodies (the largest wins: 35 KB for a 64x64x64 GEMM tile, 52 KB when a Transformer backward is
// in the level), so that a CU still holds 3-4 workgroups of different items next to each other.  Descriptors travel in the kernel
// arguments (a 3.6 KB blob; GEMM descriptors truncated behind their last segment): no dependent global read before the operands.
#include <stdlib.h>

#include "worklist_body.h"

// host side: geometry of every item, then ONE launch
static int wl_gemm_geometry(const nasrec_gemm_desc_t* g, nasrec_wl_item_t& it, int blob_left) {
  if (g->nseg < 1 || g->nseg > NASREC_MAX_SEGS) return nasrec_set_error(-2, "worklist: gemm nseg=%d", g->nseg);
  const int need = (int)offsetof(nasrec_gemm_desc_t, seg) + g->nseg * (int)sizeof(nasrec_gemm_seg_t);
  if (need > blob_left) return nasrec_set_error(-2, "worklist: truncated gemm descriptor (%d bytes) runs past the blob", need);
  const int nprob = g->zmode ? g->nseg : 1;
  const int S = g->splitk > 1 ? g->splitk : 1;
  if (g->splitk == NASREC_SPLITK_BALANCED) return nasrec_set_error(-2, "worklist: the balanced schedule is a throughput-regime launch");
  if ((it.part == 0) != (S == 1)) return nasrec_set_error(-2, "worklist: gemm part %d with splitk %d", it.part, g->splitk);
  if (S > 1 && !g->workspace) return nasrec_set_error(-3, "worklist: splitk=%d needs a workspace", S);
  int Mmax = 0, Nmax = 0, Kmax = 0;
  long wgs = 0;
  bool aux = false, uniform = true;
  for (int q = 0; q < nprob; ++q) {
    Mmax = g->seg[q].M > Mmax ? g->seg[q].M : Mmax;
    Nmax = g->seg[q].N > Nmax ? g->seg[q].N : Nmax;
    wgs += (long)((g->seg[q].M + 63) / 64) * ((g->seg[q].N + 63) / 64) * S;
  }
  for (int q = 0; q < g->nseg; ++q) {
    const nasrec_gemm_seg_t& s = g->seg[q];
    if (s.A && s.K > Kmax) Kmax = s.K;
    aux = aux || s.Aaux || s.Baux;
    if (!g->zmode && (s.ones_col != g->seg[0].ones_col || s.Mvalid != g->seg[0].Mvalid)) uniform = false;
    if (!g->zmode && s.A && ((s.Aaux != nullptr) != (g->seg[0].Aaux != nullptr) || (s.Baux != nullptr) != (g->seg[0].Baux != nullptr))) uniform = false;
  }
  if (!uniform) return nasrec_set_error(-2, "worklist: k-segments that disagree on row predicates / mask operands take the plain template");
  {
    const bool kca = g->amode == NASREC_AM_KC || g->amode == NASREC_AM_TOKK, kcb = g->bmode == NASREC_AM_KC || g->bmode == NASREC_AM_TOKK;
    if (!kca && kcb) return nasrec_set_error(-2, "worklist: operand binding a=%d b=%d has no body", g->amode, g->bmode);
  }
  if (Mmax <= 0 || Nmax <= 0) {
    it.nblk = 0;
    return 0;
  }
  if (it.part == 2) {
    const int per = (int)(((long)Mmax * Nmax + 255) / 256);
    it.geom[0] = per;
    it.nblk = per * nprob;
    return 0;
  }
  // small dense Linear forward (x W^T, binding KC / KC / plain), unsplit, no mask operands / row predicates / ones column, at most
  // WL_DENSE_STEPS 16-deep k-steps over all segments (a dead segment costs a step): a wavefront per 16 x 16 output tile (wl_dense_small)
  static const bool dense_body = getenv("NASREC_WL_DENSE_BODY") == nullptr || atoi(getenv("NASREC_WL_DENSE_BODY")) != 0;  // A/B knob
  if (dense_body && it.part == 0 && !g->zmode && !aux && g->amode == NASREC_AM_KC && g->bmode == NASREC_AM_KC && g->cmode == NASREC_CM_PLAIN) {
    bool plain = true;
    int steps = 0;
    for (int q = 0; q < g->nseg; ++q) {
      const nasrec_gemm_seg_t& s = g->seg[q];
      plain = plain && !s.ones_col && !(s.Mvalid > 0 && s.Mvalid < Mmax);
      // (the body's buffer resources and offsets are 32-bit byte counts: operand extents stay below 2^29 floats, as gemm_kslice / gemm_skinny require)
      plain = plain && (long)Mmax * s.lda + s.K < (1L << 29) && (long)Nmax * s.ldb + s.K < (1L << 29);
      steps += (s.A && s.K > 0) ? (s.K + 15) >> 4 : 1;
    }
    if (plain && steps <= 2 * WL_DENSE_STEPS && (Mmax + 15) / 16 < 0x10000) {
      it.geom[0] = (Nmax + 15) / 16;
      it.geom[1] = (Mmax + 15) / 16 | (((steps + WL_DENSE_STEPS - 1) / WL_DENSE_STEPS) << 16);  // (never 0: geom[1] == 0 is the dx body)
      it.geom[2] = WL_TOKS | (3 << 2);
      it.nblk = (it.geom[0] * ((Mmax + 15) / 16) + 3) / 4;
      return 0;
    }
  }
  // small dense input gradients (dy W per problem, binding KC / RC / plain, zmode), unsplit, no mask operands, K <= 16 WL_DENSE_STEPS:
  // a wavefront per 16 x 16 tile of one problem (wl_dense_small_dx; geom[1] = 0 tells the two bodies apart)
  if (dense_body && it.part == 0 && g->zmode && !aux && g->amode == NASREC_AM_KC && g->bmode == NASREC_AM_RC && g->cmode == NASREC_CM_PLAIN) {
    bool plain = true;
    int TU = 0;
    for (int q = 0; q < g->nseg; ++q) {
      const nasrec_gemm_seg_t& s = g->seg[q];
      plain = plain && !s.ones_col && !(s.Mvalid > 0 && s.Mvalid < s.M) && s.K <= 16 * WL_DENSE_STEPS && s.M > 0 && s.N > 0;
      plain = plain && (long)s.M * s.lda + s.K < (1L << 29) && (long)s.K * s.ldb + s.N < (1L << 29);  // (32-bit extents / offsets in the body)
      TU += ((s.M + 15) / 16) * ((s.N + 15) / 16);
    }
    if (plain && TU > 0) {
      it.geom[0] = TU;
      it.geom[1] = 0;
      it.geom[2] = WL_TOKS | (3 << 2);
      it.nblk = (TU + 3) / 4;
      return 0;
    }
  }
  // token-axis Linear forward (W x, binding KC / TOKR / TOKJ), unsplit, no mask operands / row predicates: a wavefront per (sample,
  // 16 rows of W), operands straight from memory into MFMA registers (wl_token_fwd)
  static const bool tok_body = getenv("NASREC_WL_TOKEN_BODY") == nullptr || atoi(getenv("NASREC_WL_TOKEN_BODY")) != 0;  // A/B knob
  if (tok_body && it.part == 0 && !g->zmode && !aux && g->amode == NASREC_AM_KC && g->bmode == NASREC_AM_TOKR && g->cmode == NASREC_CM_TOKJ &&
      Mmax <= 64 && (Nmax & 15) == 0) {
    bool plain = true;
    for (int q = 0; q < g->nseg; ++q) plain = plain && !g->seg[q].ones_col && !(g->seg[q].Mvalid > 0 && g->seg[q].Mvalid < Mmax);
    if (plain) {
      const int MT = (Mmax + 15) / 16;
      it.geom[0] = MT;
      it.geom[1] = 0;
      it.geom[2] = WL_TOKS | (1 << 2);
      it.nblk = ((Nmax >> 4) * MT + 3) / 4;
      return 0;
    }
  }
  // token-axis Linear input gradients (W^T dy per input segment, binding RC / TOKR / TOKJ, zmode), unsplit, no mask operands, at most
  // 64 output rows of the Linear: a wavefront per (sample, segment, 16 token rows) (wl_token_dx)
  if (tok_body && it.part == 0 && g->zmode && !aux && g->amode == NASREC_AM_RC && g->bmode == NASREC_AM_TOKR && g->cmode == NASREC_CM_TOKJ && (Nmax & 15) == 0) {
    bool plain = true;
    int TU = 0;
    for (int q = 0; q < g->nseg; ++q) {
      const nasrec_gemm_seg_t& s = g->seg[q];
      plain = plain && !s.ones_col && !(s.Mvalid > 0 && s.Mvalid < s.M) && s.N == Nmax && s.K <= 16 * WL_TOKDX_STEPS && s.M > 0;
      TU += (s.M + 15) / 16;
    }
    if (plain) {
      it.geom[0] = TU;
      it.geom[1] = 0;
      it.geom[2] = WL_TOKS | (2 << 2);
      it.nblk = ((Nmax >> 4) * TU + 3) / 4;
      return 0;
    }
  }
  // the tile choice of launch_gemm_t (gemm.hip) on 256-thread workgroups
  int tile, tbm, tbn;
  if (wgs >= GEMM_SKINNY_BELOW) {
    tile = WL_T32x32, tbm = 32, tbn = 32;
  } else {
    // skinny launches: the strip shape that pads the problems least (token-axis products have M = 8 .. 64 rows of weights against
    // N = 4096 token columns: 64 x 16 strips compute up to 4x the rows that exist); ties go to launch_gemm_t's rule.  The order in
    // which an output element accumulates over k does not depend on the tile shape, so results stay bit-identical.
    long pad64x16 = 0, pad16x64 = 0;
    for (int q = 0; q < nprob; ++q) {
      pad64x16 += (long)((g->seg[q].M + 63) / 64) * 64 * ((g->seg[q].N + 15) / 16) * 16;
      pad16x64 += (long)((g->seg[q].M + 15) / 16) * 16 * ((g->seg[q].N + 63) / 64) * 64;
    }
    static const bool by_padding = getenv("NASREC_WL_TILE_BY_PADDING") == nullptr || atoi(getenv("NASREC_WL_TILE_BY_PADDING")) != 0;  // A/B knob
    const bool wide = by_padding && pad64x16 != pad16x64 ? pad64x16 < pad16x64 : Nmax >= Mmax;
    if (wide) {
      tile = WL_T64x16, tbm = 64, tbn = 16;
    } else {
      tile = WL_T16x64, tbm = 16, tbn = 64;
    }
  }
  it.geom[0] = (Nmax + tbn - 1) / tbn;
  it.geom[1] = (Mmax + tbm - 1) / tbm;
  const bool kca_ = g->amode == NASREC_AM_KC || g->amode == NASREC_AM_TOKK, kcb_ = g->bmode == NASREC_AM_KC || g->bmode == NASREC_AM_TOKK;
  it.geom[2] = tile | ((kca_ && kcb_ ? 0 : (kca_ ? 1 : 2)) << 2) | ((aux ? 1 : 0) << 4);
  it.nblk = it.geom[0] * it.geom[1] * nprob * S;
  return 0;
}

int launch_worklist(hipStream_t st, const nasrec_worklist_desc_t* w) {
  if (w->n < 1 || w->n > NASREC_WL_MAX_ITEMS) return nasrec_set_error(-2, "worklist: n=%d outside [1,%d]", w->n, NASREC_WL_MAX_ITEMS);
  nasrec_worklist_desc_t wl = *w;
  static const bool force_big = getenv("NASREC_WL_BIG") != nullptr && atoi(getenv("NASREC_WL_BIG")) != 0;  // A/B knob
  bool big = force_big;
  int first = 0;
  for (int k = 0; k < wl.n; ++k) {
    nasrec_wl_item_t& it = wl.item[k];
    if (it.off < 0 || (it.off & 15) || it.off >= NASREC_WL_BLOB_BYTES) return nasrec_set_error(-2, "worklist: item %d offset %d", k, it.off);
    const char* blob = wl.blob + it.off;
    const int left = NASREC_WL_BLOB_BYTES - it.off;
    it.first = first;
    it.nblk = 0;
    it.geom[0] = it.geom[1] = it.geom[2] = 0;
#define WL_NEED(T) if ((int)sizeof(T) > left) return nasrec_set_error(-2, "worklist: item %d runs past the blob", k)
    switch (it.kind) {
      case NASREC_OP_GEMM: {
        const int rc = wl_gemm_geometry(reinterpret_cast<const nasrec_gemm_desc_t*>(blob), it, left);
        if (rc) return rc;
        break;
      }
      case NASREC_OP_MHA_FWD:
      case NASREC_OP_MHA_BWD: {
        WL_NEED(nasrec_mha_desc_t);
        const nasrec_mha_desc_t* d = reinterpret_cast<const nasrec_mha_desc_t*>(blob);
        if (d->N < 1 || d->N > MHA_N) return nasrec_set_error(-2, "worklist: mha N=%d out of range [1,%d]", d->N, MHA_N);
        if (it.kind == NASREC_OP_MHA_BWD) {
          if (!d->saved) return nasrec_set_error(-2, "worklist: mha backward needs the state saved by the forward launch");
          big = true;
        }
        it.nblk = d->B;
        break;
      }
      case NASREC_OP_FM_FWD:
      case NASREC_OP_FM_BWD:
        WL_NEED(nasrec_fm_desc_t);
        it.nblk = (reinterpret_cast<const nasrec_fm_desc_t*>(blob)->B + 3) / 4;
        break;
      case NASREC_OP_DOT_TRI_FWD:
      case NASREC_OP_DOT_TRI_BWD: {
        WL_NEED(nasrec_dot_tri_desc_t);
        const nasrec_dot_tri_desc_t* d = reinterpret_cast<const nasrec_dot_tri_desc_t*>(blob);
        const int P4 = (d->k1 * (d->k1 - 1) / 2 + 3) & ~3;
        if (d->k1 < 2 || 4 * (d->k1 * TRI_LD + P4) > WL_LDS_FLOATS)
          return nasrec_set_error(-2, "worklist: dot_tri k1=%d does not fit the shared LDS buffer", d->k1);
        it.nblk = (d->B + 3) / 4;
        break;
      }
      case NASREC_OP_COPY_SEGS: {
        WL_NEED(nasrec_copy_segs_desc_t);
        const nasrec_copy_segs_desc_t* d = reinterpret_cast<const nasrec_copy_segs_desc_t*>(blob);
        int W = 0;
        for (int q = 0; q < d->nseg; ++q) W = max(W, d->off[q] + d->width[q]);
        it.geom[0] = W;
        it.nblk = (int)(((long)d->B * W + 255) / 256);
        break;
      }
      case NASREC_OP_FINAL_FWD:
        WL_NEED(nasrec_final_desc_t);
        it.nblk = reinterpret_cast<const nasrec_final_desc_t*>(blob)->B;  // (a workgroup per sample)
        break;
      case NASREC_OP_FINAL_FUSED: {
        WL_NEED(nasrec_final_desc_t);
        const int rc = final_fused_check(reinterpret_cast<const nasrec_final_desc_t*>(blob));
        if (rc) return rc;
        it.nblk = reinterpret_cast<const nasrec_final_desc_t*>(blob)->B;  // (a workgroup per sample)
        break;
      }
      case NASREC_OP_FINAL_BWD: {
        WL_NEED(nasrec_final_desc_t);
        const nasrec_final_desc_t* d = reinterpret_cast<const nasrec_final_desc_t*>(blob);
        if (d->y != nullptr && d->logits == nullptr) return nasrec_set_error(-2, "worklist: final_bwd with the fused loss needs desc.logits");
        int K, nA, nB;
        final_bwd_geometry(*d, K, nA, nB);
        it.geom[0] = K, it.geom[1] = nA, it.geom[2] = nB;
        it.nblk = nA + nB + (d->y != nullptr ? 1 : 0);
        break;
      }
      case NASREC_OP_GATE_BWD: {
        WL_NEED(nasrec_gate_bwd_desc_t);
        const nasrec_gate_bwd_desc_t* d = reinterpret_cast<const nasrec_gate_bwd_desc_t*>(blob);
        it.nblk = (int)(((long)d->B * d->D + 255) / 256);
        break;
      }
      case NASREC_OP_DEDUP_IDS: {  // the id-only half of the optimizer's row dedup (B <= 256: the mask form, a workgroup per field)
        WL_NEED(nasrec_dedup_ids_desc_t);
        const nasrec_dedup_ids_desc_t* d = reinterpret_cast<const nasrec_dedup_ids_desc_t*>(blob);
        if (d->B < 1 || d->B > 256 || d->cap != 256 || d->Fs < 1 || d->Fs > NASREC_MAX_TABLES || !d->idx || !d->leader || !d->order || !d->lists || !d->counts)
          return nasrec_set_error(-2, "worklist: dedup_ids item needs B <= 256, cap 256 and every output (B=%d cap=%d)", d->B, d->cap);
        it.nblk = d->Fs;
        break;
      }
      case NASREC_OP_REDUCE_ROWS: {
        WL_NEED(nasrec_wl_reduce_t);
        const nasrec_wl_reduce_t* d = reinterpret_cast<const nasrec_wl_reduce_t*>(blob);
        if (d->ndst < 1 || d->ndst > NASREC_WL_REDUCE_DST) return nasrec_set_error(-2, "worklist: reduce_rows ndst=%d", d->ndst);
        it.nblk = (d->C + 15) / 16;
        break;
      }
      default:
        return nasrec_set_error(-2, "worklist: item %d has kind %d, which has no body in the worklist kernel", k, it.kind);
    }
#undef WL_NEED
    first += it.nblk;
  }
  wl.total_blocks = first;
  if (first < 1) return 0;
  // the item table packed into the twelve leading scalar arguments (worklist_body.h: they are preloaded into SGPRs)
  static_assert(NASREC_WL_MAX_ITEMS == 12 && NASREC_WL_BLOB_BYTES <= 256 * 16 && offsetof(nasrec_worklist_desc_t, blob) % 16 == 0, "packed item table");
  unsigned pf[6], pm[6];
  bool packed = first < 0xffff;
  for (int q = 0; q < 6; ++q) {
    unsigned f2[2], m2[2];
    for (int h = 0; h < 2; ++h) {
      const int k = 2 * q + h;
      if (k < wl.n) {
        f2[h] = (unsigned)wl.item[k].first;
        m2[h] = (unsigned)wl.item[k].kind | ((unsigned)wl.item[k].part << 6) | ((unsigned)(wl.item[k].off >> 4) << 8);
        packed = packed && wl.item[k].kind < 64 && wl.item[k].part < 4;
      } else {
        f2[h] = 0xffffu;
        m2[h] = 0;
      }
    }
    pf[q] = f2[0] | (f2[1] << 16);
    pm[q] = m2[0] | (m2[1] << 16);
  }
  if (!packed) pf[0] = WL_PACKED_NONE;
  if (big) {
    hipLaunchKernelGGL(worklist_kernel<true>, dim3((unsigned)first), dim3(256), sizeof(float) * WL_LDS_BIG_FLOATS, st, pf[0], pf[1], pf[2], pf[3], pf[4],
                       pf[5], pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], wl);
  } else {
    hipLaunchKernelGGL(worklist_kernel<false>, dim3((unsigned)first), dim3(256), sizeof(float) * WL_LDS_FLOATS, st, pf[0], pf[1], pf[2], pf[3], pf[4],
                       pf[5], pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], wl);
  }
  return nasrec_check_launch("worklist");
}
