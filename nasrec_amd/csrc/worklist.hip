// NASREC_OP_WORKLIST (include/nasrec_hip.h): one launch = the operators of one LEVEL of a batch-256 step (nasrec_amd/schedule.py),
// side by side on disjoint workgroup ranges.  Every item runs the body of its stand-alone kernel — the run-time-binding GEMM tile
// (gemm_rt.h), the split-K second pass, the Transformer forward / backward, the FM / DotProduct cores, segmented copies, the gating
// backward, fixed-order row reductions — so results are bit-identical to the separate launches; what disappears is a kernel
// boundary (~5 us of cold-L2 start at batch 256) per operator that has an independent neighbour.
//
// One LDS buffer is shared by all bodies (the largest wins: 31 KB, 39.5 KB when a Transformer backward is in the level), so that a CU
// holds four workgroups of different items next to each other.  Descriptors travel in the kernel arguments (a 3.6 KB blob; GEMM
// descriptors truncated behind their last segment) or, for plans whose pointers never change, sit in device memory (ABI 17:
// nasrec_worklist_prepare below — geometry once per plan, warm lines instead of a fresh copy per launch).
#include <stdlib.h>
#include <string.h>

#include <vector>

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
  bool aux_b = false;  // (a mask on dy — the fused ReLU backward — is applied as the operand is loaded; one on the weights has no body here)
  for (int q = 0; q < g->nseg; ++q) aux_b = aux_b || g->seg[q].Baux;
  if (dense_body && it.part == 0 && g->zmode && !aux_b && g->amode == NASREC_AM_KC && g->bmode == NASREC_AM_RC && g->cmode == NASREC_CM_PLAIN) {
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
  // token-axis Linear input gradients (W^T dy per input segment, binding RC / TOKR / TOKJ, zmode), unsplit, no mask on the weights (one on
  // dy — the fused ReLU backward — is applied as the operand is loaded), at most 64 output rows of the Linear: a wavefront per (sample,
  // segment, 16 token rows) (wl_token_dx)
  bool aux_a = false;
  for (int q = 0; q < g->nseg; ++q) aux_a = aux_a || g->seg[q].Aaux;
  if (tok_body && it.part == 0 && g->zmode && !aux_a && g->amode == NASREC_AM_RC && g->bmode == NASREC_AM_TOKR && g->cmode == NASREC_CM_TOKJ && (Nmax & 15) == 0) {
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
      it.nblk = std::min(((Nmax >> 4) * TU + 3) / 4, WL_TOK_MAX_WG);
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

// geometry of ONE item (nblk, geom) from its descriptor at `blob` (`left` bytes up to the end of the blob); big: a Transformer backward
static int wl_item_geometry(nasrec_wl_item_t& it, const char* blob, int left, int k, bool& big) {
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
  return 0;
}

// geometry of every item, the workgroup ranges and the packed item table of a launch (wl is completed in place)
static int wl_finalize(nasrec_worklist_desc_t& wl, bool& big, unsigned (&pf)[6], unsigned (&pm)[6]) {
  if (wl.n < 1 || wl.n > NASREC_WL_MAX_ITEMS) return nasrec_set_error(-2, "worklist: n=%d outside [1,%d]", wl.n, NASREC_WL_MAX_ITEMS);
  static const bool force_big = getenv("NASREC_WL_BIG") != nullptr && atoi(getenv("NASREC_WL_BIG")) != 0;  // A/B knob
  big = force_big;
  int first = 0;
  for (int k = 0; k < wl.n; ++k) {
    nasrec_wl_item_t& it = wl.item[k];
    if (it.off < 0 || (it.off & 15) || it.off >= NASREC_WL_BLOB_BYTES) return nasrec_set_error(-2, "worklist: item %d offset %d", k, it.off);
    const char* blob = wl.blob + it.off;
    const int left = NASREC_WL_BLOB_BYTES - it.off;
    it.first = first;
    {
      const int rc = wl_item_geometry(it, blob, left, k, big);
      if (rc) return rc;
    }
    first += it.nblk;
  }
  wl.total_blocks = first;
  // the item table packed into the twelve leading scalar arguments (worklist_body.h: they are preloaded into SGPRs)
  static_assert(NASREC_WL_MAX_ITEMS == 12 && NASREC_WL_BLOB_BYTES <= 256 * 16 && offsetof(nasrec_worklist_desc_t, blob) % 16 == 0, "packed item table");
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
  return 0;
}

int launch_worklist(hipStream_t st, const nasrec_worklist_desc_t* w) {
  nasrec_worklist_desc_t wl = *w;
  bool big;
  unsigned pf[6], pm[6];
  const int rc = wl_finalize(wl, big, pf, pm);
  if (rc) return rc;
  const int first = wl.total_blocks;
  if (first < 1) return 0;
  if (big) {
    hipLaunchKernelGGL(worklist_kernel<true>, dim3((unsigned)first), dim3(256), sizeof(float) * WL_LDS_BIG_FLOATS, st, pf[0], pf[1], pf[2], pf[3], pf[4],
                       pf[5], pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], wl);
  } else {
    hipLaunchKernelGGL(worklist_kernel<false>, dim3((unsigned)first), dim3(256), sizeof(float) * WL_LDS_FLOATS, st, pf[0], pf[1], pf[2], pf[3], pf[4],
                       pf[5], pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], wl);
  }
  return nasrec_check_launch("worklist");
}

// ABI 17: the same launch with the completed descriptor resident in device memory (include/nasrec_hip.h nasrec_worklist_dev_desc_t)
extern "C" int nasrec_worklist_prepare(const nasrec_worklist_desc_t* w, void* dev_buf, nasrec_worklist_dev_desc_t* out) {
  if (!w || w->kind != NASREC_OP_WORKLIST || !dev_buf || !out) return nasrec_set_error(-1, "worklist_prepare: wrong descriptor / null buffer");
  nasrec_worklist_desc_t wl = *w;
  bool big;
  const int rc = wl_finalize(wl, big, out->pf, out->pm);
  if (rc) return rc;
  if (hipMemcpy(dev_buf, &wl, sizeof(wl), hipMemcpyHostToDevice) != hipSuccess) return nasrec_set_error(-3, "worklist_prepare: copy to the device buffer failed");
  out->kind = NASREC_OP_WORKLIST_DEV;
  out->total_blocks = wl.total_blocks;
  out->big = big ? 1 : 0;
  out->_pad = 0;
  out->dev = reinterpret_cast<const nasrec_worklist_desc_t*>(dev_buf);
  return 0;
}

int launch_worklist_dev(hipStream_t st, const nasrec_worklist_dev_desc_t* d) {
  if (!d->dev) return nasrec_set_error(-2, "worklist (device-resident): not prepared");
  if (d->total_blocks < 1) return 0;
  const unsigned* pf = d->pf;
  const unsigned* pm = d->pm;
  if (d->big) {
    hipLaunchKernelGGL(worklist_dev_kernel<true>, dim3((unsigned)d->total_blocks), dim3(256), sizeof(float) * WL_LDS_BIG_FLOATS, st, pf[0], pf[1], pf[2], pf[3],
                       pf[4], pf[5], pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], d->dev);
  } else {
    hipLaunchKernelGGL(worklist_dev_kernel<false>, dim3((unsigned)d->total_blocks), dim3(256), sizeof(float) * WL_LDS_FLOATS, st, pf[0], pf[1], pf[2], pf[3],
                       pf[4], pf[5], pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], d->dev);
  }
  return nasrec_check_launch("worklist (device-resident)");
}

// ======================================================================================================================================
// NASREC_OP_PERSIST (include/nasrec_hip.h): the items of many levels in ONE launch, dependencies resolved in the kernel.
// ======================================================================================================================================
#define PS_SPIN_BUDGET_TICKS 1000000ull  // 10 ms of the 100 MHz wall clock: a wait that long is a bug (or a lost workgroup), not a slow producer

template <bool BIG>
__device__ __forceinline__ void ps_run_item(int it_kind, int it_part, int g0, int g1, int g2, unsigned long long blob, int vb, int nblk, int nwg) {
  float* lds = wl_lds;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  switch (it_kind) {
    case NASREC_OP_GEMM: {
      if (it_part == 2) {  // (units of one element per thread: a workgroup walks vb, vb + nwg, ...)
        for (int u = vb; u < nblk; u += nwg) wl_gemm_second_pass(blob, u, g0);
        break;
      }
      const int cfg = g2;  // tile | binding pair << 2 | mask operand << 4
      if ((cfg & 3) == WL_TOKS && ((cfg >> 2) & 3) == 3 && g1 == 0) wl_dense_small_dx(blob, vb, g0);
      else if ((cfg & 3) == WL_TOKS && ((cfg >> 2) & 3) == 3) wl_dense_small(blob, vb, g0, g1);
      else if ((cfg & 3) == WL_TOKS && ((cfg >> 2) & 3) == 1) wl_token_fwd(blob, vb, g0);
      else if ((cfg & 3) == WL_TOKS) wl_token_dx(blob, vb, g0);
      else if ((cfg >> 4) & 1) {
        for (int u = vb; u < nblk; u += nwg) {
          wl_gemm_bind<true>((cfg >> 2) & 3, cfg & 3, blob, u, g0, g1);
          if (u + nwg < nblk) __syncthreads();
        }
      } else {
        for (int u = vb; u < nblk; u += nwg) {  // (tiles of a product: the bulk items walk several per workgroup)
          wl_gemm_bind<false>((cfg >> 2) & 3, cfg & 3, blob, u, g0, g1);
          if (u + nwg < nblk) __syncthreads();
        }
      }
      break;
    }
    case NASREC_OP_MHA_FWD:
      wl_mha_fwd(blob, vb);
      break;
    case NASREC_OP_MHA_BWD:
      if (BIG) mha_bwd_tok(wl_ref<nasrec_mha_desc_t>(blob), __builtin_amdgcn_readfirstlane(vb), wl_lds);
      break;
    case NASREC_OP_FM_FWD: {
      const nasrec_fm_desc_t& d = wl_ref<nasrec_fm_desc_t>(blob);
      const int b = vb * 4 + wave;
      if (b < d.B) fm_fwd_sample(d, b, lane);
      break;
    }
    case NASREC_OP_FM_BWD: {
      const nasrec_fm_desc_t& d = wl_ref<nasrec_fm_desc_t>(blob);
      const int b = vb * 4 + wave;
      if (b < d.B) fm_bwd_sample(d, b, lane);
      break;
    }
    case NASREC_OP_DOT_TRI_FWD: {
      const nasrec_dot_tri_desc_t& d = wl_ref<nasrec_dot_tri_desc_t>(blob);
      const int b = vb * 4 + wave;
      if (b < d.B) dot_tri_fwd_sample(d, b, lane, lds + wave * (d.k1 * TRI_LD));
      break;
    }
    case NASREC_OP_DOT_TRI_BWD: {
      const nasrec_dot_tri_desc_t& d = wl_ref<nasrec_dot_tri_desc_t>(blob);
      const int b = vb * 4 + wave;
      const int P4 = (d.k1 * (d.k1 - 1) / 2 + 3) & ~3;
      if (b < d.B) dot_tri_bwd_sample(d, b, lane, lds + wave * (d.k1 * TRI_LD), lds + 4 * (d.k1 * TRI_LD) + wave * P4);
      break;
    }
    case NASREC_OP_COPY_SEGS: {
      const nasrec_copy_segs_desc_t& d = wl_ref<nasrec_copy_segs_desc_t>(blob);
      const int W = g0;
      const long t = (long)vb * 256 + tid;
      if (t < (long)d.B * W) copy_segs_element(d, (int)(t / W), (int)(t % W));
      break;
    }
    case NASREC_OP_GATE_BWD:
      gate_bwd_element(wl_ref<nasrec_gate_bwd_desc_t>(blob), (long)vb * 256 + tid);
      break;
    case NASREC_OP_REDUCE_ROWS:
      reduce_rows_block(wl_ref<nasrec_wl_reduce_t>(blob), vb, tid, lds);
      break;
    case NASREC_OP_DEDUP_IDS: {
      const nasrec_dedup_ids_desc_t& d = wl_ref<nasrec_dedup_ids_desc_t>(blob);
      dedup_ids_small_body(d, d.idx, d.B, d.Fs, __builtin_amdgcn_readfirstlane(vb), reinterpret_cast<int*>(lds), reinterpret_cast<int*>(lds) + 256);
      break;
    }
    case NASREC_OP_FINAL_FWD:
      final_fwd_block(wl_ref<nasrec_final_desc_t>(blob), __builtin_amdgcn_readfirstlane(vb), lds);
      break;
    case NASREC_OP_FINAL_FUSED:
      final_fused_block(wl_ref<nasrec_final_desc_t>(blob), __builtin_amdgcn_readfirstlane(vb), lds);
      break;
    case NASREC_OP_FINAL_BWD:
      for (int u = vb; u < nblk; u += nwg) {
        final_bwd_block(wl_ref<nasrec_final_desc_t>(blob), g0, g1, g2, __builtin_amdgcn_readfirstlane(u), lds);
        if (u + nwg < nblk) __syncthreads();  // (the next unit reuses the LDS buffer)
      }
      break;
    default:
      break;
  }
}

template <bool BIG>
__global__ __launch_bounds__(256, BIG ? 3 : 4) void persist_kernel(const nasrec_persist_item_t* __restrict__ items, const char* __restrict__ blob0,
                                                                    const uint16_t* __restrict__ chunk_item, unsigned long long* counters,
                                                                    uint32_t* flags, uint32_t* err, int n, int shard_above, unsigned long long* trace) {
  const int bid = blockIdx.x, tid = threadIdx.x;
  if (trace && tid == 0) trace[4 * (size_t)bid] = wall_clock64();
  // the workgroup's item: chunk index, then a short scan (the tables are written once per plan: scalar loads, warm after the first few workgroups)
  int k = __builtin_amdgcn_readfirstlane((int)chunk_item[bid >> 4]);
  {
    const nasrec_persist_item_t* __restrict__ tab = items;
#pragma unroll 1
    while (k + 1 < n && bid >= wl_ref<nasrec_persist_item_t>((unsigned long long)(tab + k + 1)).first) ++k;
  }
  const nasrec_persist_item_t& it = wl_ref<nasrec_persist_item_t>((unsigned long long)(items + k));
  const int it_first = it.first, it_nblk = it.nblk, it_kind = it.kind, it_part = it.part, it_off = it.off, ndeps = it.ndeps, nsucc = it._pad[0];
  const int it_nwg = it._pad[1], desc_bytes = it._pad[2];
  const int g0 = it.geom[0], g1 = it.geom[1], g2 = it.geom[2];
  const int wg = bid - it_first;  // this workgroup's index in its item: it runs the units wg, wg + nwg, ... < nblk
  const bool tracked = ndeps > 0 || nsucc > 0;
  const unsigned long long blob = (unsigned long long)blob0 + (unsigned)it_off;
  // arrival bookkeeping of this workgroup: shard of its item's counters, the shard's size
  const int NS = it_nwg > shard_above ? NASREC_PS_SHARDS : 1;  // (one arrival counter up to shard_above workgroups: ~12 ns per add on one address)
  const int shard = wg % NS;
  const unsigned long long shard_n = (unsigned long long)((it_nwg - shard + NS - 1) / NS);
  unsigned long long* const cnt = counters + (size_t)k * (NASREC_PS_SHARDS + 1) * NASREC_PS_COUNTER_STRIDE;
  if (ndeps > 0) {
    if (tid < 64) {
      // the step's epoch: this item's shard counter only ever holds whole epochs plus the arrivals of THIS epoch, and this workgroup's own
      // arrival is still to come — floor(counter / shard size) is the number of completed steps
      const unsigned long long c = __hip_atomic_load(cnt + shard * NASREC_PS_COUNTER_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int dep = tid < ndeps ? items[k].deps[tid] : -1;  // lane i polls dependency i
      const uint32_t* f = flags + ((size_t)(dep < 0 ? 0 : dep) * NASREC_PS_REPL + (bid % NASREC_PS_REPL)) * NASREC_PS_FLAG_STRIDE;
      uint32_t seen = dep < 0 ? 0u : __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (in flight together with the counter)
      wl_warm_blob(blob, desc_bytes);  // the body's descriptor through the scalar cache while the polls are in flight
      const uint32_t want = (uint32_t)(c / shard_n) + 1u;
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        const bool ok = dep < 0 || seen == want;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > PS_SPIN_BUDGET_TICKS) {
          if (!ok) {
            atomicExch(err, 1u);
            err[1] = (uint32_t)k;
            err[2] = (uint32_t)dep;
          }
          break;
        }
        if (dep >= 0) seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // what the dependencies stored is in (uncached) memory; this CU's L1 may still hold older lines of the same addresses
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  if (trace && tid == 0) trace[4 * (size_t)bid + 1] = wall_clock64();
  // (no loop around the switch: with the bodies inside a loop the register allocator keeps their hoisted descriptor reads live across
  // ALL of them — 168 registers, 300 spilled, a step three times slower.  Only the two kinds whose units are one element per thread walk
  // several units per workgroup, in loops of their own inside ps_run_item.)
  ps_run_item<BIG>(it_kind, it_part, g0, g1, g2, blob, wg, it_nblk, it_nwg);
  if (trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) trace[4 * (size_t)bid + 2] = wall_clock64();
  }
  if (tracked) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have left
    __syncthreads();
    if (tid < 64) {
      unsigned long long old = 0;
      if (tid == 0) old = __hip_atomic_fetch_add(cnt + shard * NASREC_PS_COUNTER_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(old >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)old);
      bool last = (old + 1) % shard_n == 0;
      const uint32_t e1 = (uint32_t)(old / shard_n) + 1u;
      if (last && NS > 1) {
        unsigned long long o2 = 0;
        if (tid == 0) o2 = __hip_atomic_fetch_add(cnt + NASREC_PS_SHARDS * NASREC_PS_COUNTER_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o2 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(o2 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)o2);
        last = (o2 + 1) % (unsigned long long)NS == 0;
      }
      if (last && nsucc > 0 && tid < NASREC_PS_REPL)
        __hip_atomic_store(flags + ((size_t)k * NASREC_PS_REPL + tid) * NASREC_PS_FLAG_STRIDE, e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (trace && tid == 0) trace[4 * (size_t)bid + 3] = wall_clock64();
}

int nasrec_persist_prepare(nasrec_persist_desc_t* d) {
  if (!d || d->kind != NASREC_OP_PERSIST) return nasrec_set_error(-1, "persist: wrong descriptor kind");
  if (d->n < 1 || d->n > 65535) return nasrec_set_error(-2, "persist: n=%d items", d->n);
  if (!d->host_items || !d->host_blob) return nasrec_set_error(-2, "persist: a host table pointer is null");
  const bool dry = !d->items;  // geometry only (no device): the workgroup ranges are written back into host_items (CPU tests, tools)
  if (!dry && (!d->blob || !d->chunk_item || !d->counters || !d->flags || !d->err)) return nasrec_set_error(-2, "persist: a device table pointer is null");
  std::vector<nasrec_persist_item_t> tab(d->host_items, d->host_items + d->n);
  bool big = false;
  int first = 0;
  for (int k = 0; k < d->n; ++k) {
    nasrec_persist_item_t& p = tab[k];
    if (p.off < 0 || (p.off & 15) || p.off >= d->blob_bytes) return nasrec_set_error(-2, "persist: item %d offset %d", k, p.off);
    if (p.ndeps < 0 || p.ndeps > NASREC_PS_MAX_DEPS) return nasrec_set_error(-2, "persist: item %d has %d dependencies", k, p.ndeps);
    for (int q = 0; q < p.ndeps; ++q)
      if (p.deps[q] < 0 || p.deps[q] >= k) return nasrec_set_error(-2, "persist: item %d depends on item %d (not an earlier one)", k, p.deps[q]);
    nasrec_wl_item_t w;
    w.kind = p.kind, w.part = p.part, w.off = p.off;
    const int rc = wl_item_geometry(w, d->host_blob + p.off, d->blob_bytes - p.off, k, big);
    if (rc) return rc;
    if (w.nblk < 1) return nasrec_set_error(-2, "persist: item %d (kind %d) has no workgroups; the caller drops empty operators", k, p.kind);
    // units per workgroup: an item of thousands of one-element-per-thread units (split-K second passes, the final-logit backward) runs
    // as at most max_wg workgroups that walk their units — the wait / acquire / arrival of the protocol is paid per WORKGROUP
    static const int max_wg = getenv("NASREC_PS_MAX_WG") ? atoi(getenv("NASREC_PS_MAX_WG")) : 512;
    // (ps_run_item: the kinds with a unit loop — second passes, the final-logit backward, the tile bodies of the products)
    const bool tiles = p.kind == NASREC_OP_GEMM && p.part != 2 && (w.geom[2] & 3) != WL_TOKS;
    const bool walks = (p.kind == NASREC_OP_GEMM && p.part == 2) || p.kind == NASREC_OP_FINAL_BWD || tiles;
    // the caller's hint (item._pad[1] > 0): an operator nobody waits for soon runs on few workgroups, each walking many units, so that it
    // cannot fill the chip's workgroup slots in front of the operators the step's critical chain is waiting for
    const int hint = d->host_items[k]._pad[1];
    const int cap = hint > 0 ? hint : (tiles ? 1 << 30 : max_wg);
    const int upw = walks ? (w.nblk + cap - 1) / (cap > 0 ? cap : 1) : 1;
    const int nwg = (w.nblk + upw - 1) / upw;
    p.first = first, p.nblk = w.nblk;
    p.geom[0] = w.geom[0], p.geom[1] = w.geom[1], p.geom[2] = w.geom[2];
    p._pad[1] = nwg;
    p._pad[2] = ((k + 1 < d->n ? d->host_items[k + 1].off : d->blob_bytes) - p.off + 63) & ~63;  // bytes of the descriptor (rounded up to the warm-up's step)
    first += nwg;
  }
  // successors: an item nobody waits for publishes nothing
  for (int k = 0; k < d->n; ++k) tab[k]._pad[0] = 0;
  for (int k = 0; k < d->n; ++k)
    for (int q = 0; q < tab[k].ndeps; ++q) tab[tab[k].deps[q]]._pad[0] += 1;
  const int chunks = (first + 15) / 16;
  if (!dry && chunks > d->chunk_cap) return nasrec_set_error(-2, "persist: %d workgroups need %d chunk entries, the table holds %d", first, chunks, d->chunk_cap);
  std::vector<uint16_t> ci(chunks);
  for (int c = 0, k = 0; c < chunks; ++c) {
    while (k + 1 < d->n && 16 * c >= tab[k + 1].first) ++k;
    ci[c] = (uint16_t)k;
  }
  d->total_blocks = first;
  d->big = big ? 1 : 0;
  if (dry) {
    memcpy(const_cast<nasrec_persist_item_t*>(d->host_items), tab.data(), sizeof(nasrec_persist_item_t) * d->n);
    return 0;
  }
  hipError_t rc = hipMemcpy(d->items, tab.data(), sizeof(nasrec_persist_item_t) * d->n, hipMemcpyHostToDevice);
  if (rc == hipSuccess) rc = hipMemcpy(d->blob, d->host_blob, d->blob_bytes, hipMemcpyHostToDevice);
  if (rc == hipSuccess) rc = hipMemcpy(d->chunk_item, ci.data(), sizeof(uint16_t) * chunks, hipMemcpyHostToDevice);
  if (rc == hipSuccess) rc = hipMemset(d->counters, 0, sizeof(unsigned long long) * (size_t)d->n * (NASREC_PS_SHARDS + 1) * NASREC_PS_COUNTER_STRIDE);
  if (rc == hipSuccess) rc = hipMemset(d->flags, 0, sizeof(uint32_t) * (size_t)d->n * NASREC_PS_REPL * NASREC_PS_FLAG_STRIDE);
  if (rc == hipSuccess) rc = hipMemset(d->err, 0, sizeof(uint32_t) * 4);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "persist: table upload: %s", hipGetErrorString(rc));
  return 0;
}

int launch_persist(hipStream_t st, const nasrec_persist_desc_t* d) {
  if (d->total_blocks < 1) return nasrec_set_error(-2, "persist: the descriptor has not been through nasrec_persist_prepare");
  static const int shard_above = getenv("NASREC_PS_SHARD_ABOVE") ? atoi(getenv("NASREC_PS_SHARD_ABOVE")) : 64;  // A/B knob
  if (d->big) {
    hipLaunchKernelGGL(persist_kernel<true>, dim3((unsigned)d->total_blocks), dim3(256), sizeof(float) * WL_LDS_BIG_FLOATS, st, d->items, d->blob, d->chunk_item,
                       d->counters, d->flags, d->err, d->n, shard_above, d->trace);
  } else {
    hipLaunchKernelGGL(persist_kernel<false>, dim3((unsigned)d->total_blocks), dim3(256), sizeof(float) * WL_LDS_FLOATS, st, d->items, d->blob, d->chunk_item,
                       d->counters, d->flags, d->err, d->n, shard_above, d->trace);
  }
  return nasrec_check_launch("persist");
}
