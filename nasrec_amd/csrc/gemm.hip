// fp32 MFMA GEMM family for gfx950: C(i,j) = epilogue(sum_k A(i,k) * B(j,k)).
//
// One kernel template, six operand bindings (include/nasrec_hip.h "addressing modes"): it serves the dense
// nn.Linear forward (modules.py:171 …), its two autograd products, and the token-axis Linear over
// [B,N,16] tensors (modules.py:222-234, 358-361, 648-650) with its two autograd products — without ever
// materialising a transpose or a concatenation (segments = K-ranges or independent z-problems).
//
// Tiling: block tile 64x64x32 on 4 or 8 wavefronts (see NT below), each wave 32x32 (or 32x16) =
// v_mfma_f32_16x16x4_f32 tiles (exact fp32 FMA chain, so results match an fp32 dot product bit-for-bit in
// k order within a lane group).  Global -> registers -> LDS staging with one tile of register prefetch.
// The k index inside a 16-deep tile is permuted (lane group g takes k = 4g..4g+3 as one ds_read_b128)
// identically for A and B, which only reorders the fp32 summation.
#include "gemm_tile.h"
#include "gemm_ring.h"

template <int AM, int BMODE, int CM, int NT, int TK, int TBM, int TBN>
__global__ __launch_bounds__(NT) void gemm_kernel(const nasrec_gemm_desc_t d, int Mmax, int Nmax) {
  gemm_tile<AM, BMODE, CM, NT, TK, TBM, TBN>(d, Mmax, Nmax, blockIdx.x, blockIdx.y, blockIdx.z);
}

#ifndef GEMM_RING
#define GEMM_RING 4  // staged k-tiles in flight per workgroup in the latency-regime configurations (gemm_ring.h)
#endif
template <int AM, int BMODE, int CM, int NT, int TK, int TBM, int TBN, bool AUX>
__global__ __launch_bounds__(NT) void gemm_ring_kernel(const nasrec_gemm_desc_t d, int Mmax, int Nmax) {
  gemm_tile_ring<AM, BMODE, CM, NT, TK, TBM, TBN, AUX, AUX ? 2 : GEMM_RING>(d, Mmax, Nmax, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <int CM>
__global__ __launch_bounds__(256) void gemm_splitk_epilogue(const nasrec_gemm_desc_t d, int Mmax, int Nmax) {
  splitk_second_pass<CM>(d, Mmax, Nmax, blockIdx.z, (long)blockIdx.x * 256 + threadIdx.x);
}

// NASREC_OP_SPLITK_EPILOGUES: the second passes of up to three launches (defer_second_pass = 1, CM_PLAIN) in one
struct EpiGeom {
  int first[NASREC_EPILOGUES_MAX + 1];  // first workgroup of each descriptor's share
  int per[NASREC_EPILOGUES_MAX];        // workgroups per problem
  int Mmax[NASREC_EPILOGUES_MAX], Nmax[NASREC_EPILOGUES_MAX];
};

__global__ __launch_bounds__(256) void gemm_splitk_epilogues_kernel(const nasrec_splitk_epilogues_desc_t b, const EpiGeom g) {
  int k = 0;
  while (k + 1 < b.n && (int)blockIdx.x >= g.first[k + 1]) ++k;
  const int local = (int)blockIdx.x - g.first[k];
  const int z = local / g.per[k];
  splitk_second_pass<NASREC_CM_PLAIN>(b.g[k], g.Mmax[k], g.Nmax[k], z, (long)(local - z * g.per[k]) * 256 + threadIdx.x);
}

int launch_splitk_epilogues(hipStream_t st, const nasrec_splitk_epilogues_desc_t* b) {
  if (b->n < 1 || b->n > NASREC_EPILOGUES_MAX) return nasrec_set_error(-2, "splitk_epilogues: n=%d outside [1,%d]", b->n, NASREC_EPILOGUES_MAX);
  EpiGeom g;
  g.first[0] = 0;
  for (int k = 0; k < b->n; ++k) {
    const nasrec_gemm_desc_t& d = b->g[k];
    if (d.cmode != NASREC_CM_PLAIN || d.splitk < 2 || !d.workspace)
      return nasrec_set_error(-2, "splitk_epilogues: descriptor %d is not a split CM_PLAIN launch", k);
    const int nprob = d.zmode ? d.nseg : 1;
    int Mm = 0, Nm = 0;
    for (int q = 0; q < nprob; ++q) {
      if (d.seg[q].M > Mm) Mm = d.seg[q].M;
      if (d.seg[q].N > Nm) Nm = d.seg[q].N;
    }
    g.Mmax[k] = Mm;
    g.Nmax[k] = Nm;
    g.per[k] = (int)(((long)Mm * Nm + 255) / 256);
    g.first[k + 1] = g.first[k] + g.per[k] * nprob;
  }
  if (g.first[b->n] < 1) return 0;
  hipLaunchKernelGGL(gemm_splitk_epilogues_kernel, dim3((unsigned)g.first[b->n]), dim3(256), 0, st, *b, g);
  return nasrec_check_launch("splitk_epilogues");
}

template <int AM, int BMODE, int CM, int NT, int TK, int TBM, int TBN>
static void launch_cfg(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax, int zdim) {
  dim3 grid((Nmax + TBN - 1) / TBN, (Mmax + TBM - 1) / TBM, zdim);
  hipLaunchKernelGGL((gemm_kernel<AM, BMODE, CM, NT, TK, TBM, TBN>), grid, dim3(NT), 0, st, *d, Mmax, Nmax);
}

// latency-regime configurations run the ring variant unless the launch's k-segments disagree on the row predicates
template <int AM, int BMODE, int CM, int NT, int TK, int TBM, int TBN>
static void launch_ring(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax, int zdim) {
  bool aux = false, uniform = true;
  for (int q = 0; q < d->nseg; ++q) {
    aux = aux || d->seg[q].Aaux || d->seg[q].Baux;
    if (!d->zmode && (d->seg[q].ones_col != d->seg[0].ones_col || d->seg[q].Mvalid != d->seg[0].Mvalid)) uniform = false;
    // (the ring keeps one "masked operand" flag per launch segment in flight, not per staged slot: k-segments that disagree on
    // having a ReLU-mask operand — contributions of different Linears concatenated along K — take the plain template)
    if (!d->zmode && d->seg[q].A && ((d->seg[q].Aaux != nullptr) != (d->seg[0].Aaux != nullptr) || (d->seg[q].Baux != nullptr) != (d->seg[0].Baux != nullptr)))
      uniform = false;
  }
  if (!uniform || !GEMM_RING) {
    launch_cfg<AM, BMODE, CM, NT, TK, TBM, TBN>(st, d, Mmax, Nmax, zdim);
    return;
  }
  dim3 grid((Nmax + TBN - 1) / TBN, (Mmax + TBM - 1) / TBM, zdim);
  if (aux)
    hipLaunchKernelGGL((gemm_ring_kernel<AM, BMODE, CM, NT, TK, TBM, TBN, true>), grid, dim3(NT), 0, st, *d, Mmax, Nmax);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<AM, BMODE, CM, NT, TK, TBM, TBN, false>), grid, dim3(NT), 0, st, *d, Mmax, Nmax);
}

template <int AM, int BMODE, int CM>
static int launch_gemm_t(hipStream_t st, const nasrec_gemm_desc_t* d) {
  int Mmax = 0, Nmax = 0, Kmax = 0;
  const int nprob = d->zmode ? d->nseg : 1;
  for (int q = 0; q < nprob; ++q) {
    if (d->seg[q].M > Mmax) Mmax = d->seg[q].M;
    if (d->seg[q].N > Nmax) Nmax = d->seg[q].N;
  }
  for (int q = 0; q < d->nseg; ++q)
    if (d->seg[q].A && d->seg[q].K > Kmax) Kmax = d->seg[q].K;
  if (Mmax <= 0 || Nmax <= 0) return 0;
  const int S = d->splitk > 1 ? d->splitk : 1;
  if (S > 1 && d->workspace == nullptr) return nasrec_set_error(-3, "gemm: splitk=%d needs a workspace", S);
  const int zdim = nprob * S;
  // LIVE workgroups of a 64x64 tiling (a zmode grid is padded to Mmax x Nmax: the surplus workgroups exit at once)
  long wgs = 0;
  for (int q = 0; q < nprob; ++q) wgs += (long)((d->seg[q].M + 63) / 64) * ((d->seg[q].N + 63) / 64) * S;
  const bool deep = Kmax > 32;
  if (AM == NASREC_AM_KC && BMODE == NASREC_AM_KC && CM == NASREC_CM_PLAIN && gemm_kslice_eligible(d)) {
    return launch_gemm_kslice(st, d);  // batch-256 regime, one large forward product: K split inside the workgroup, single pass
  } else if (AM == NASREC_AM_KC && (BMODE == NASREC_AM_KC || BMODE == NASREC_AM_RC) && CM == NASREC_CM_PLAIN && gemm_skinny_n_eligible(d)) {
    return launch_gemm_skinny_n(st, d);  // large batch, N <= 16: a streaming read of x, K split inside the workgroup, single pass
  } else if (AM == NASREC_AM_KC && (BMODE == NASREC_AM_KC || BMODE == NASREC_AM_RC) && CM == NASREC_CM_PLAIN && gemm_tinyk_eligible(d)) {
    return launch_gemm_tinyk(st, d);  // large batch, K <= 16: a streaming write of y, weights in registers, no LDS
  } else if (AM == NASREC_AM_TOKK && token_dw_eligible(d)) {
    launch_token_dw(st, d, Mmax, Nmax);  // large batch: a wavefront per sample, operands straight to MFMA registers (token_linear.hip)
  } else if (CM == NASREC_CM_PLAIN && gemm_fast_eligible(d, Mmax, Nmax)) {
    // throughput regime: 128x128x32 tiles, 16-byte staging, LDS double buffer (gemm_fast.hip)
    const int rc = launch_gemm_fast(st, d, Mmax, Nmax, zdim);
    if (rc) return rc;
  } else if (d->splitk == NASREC_SPLITK_BALANCED) {
    return nasrec_set_error(-2, "gemm: the balanced schedule exists for throughput-regime launches only (plan.py decides both)");
  } else if (wgs >= 1024) {
    launch_cfg<AM, BMODE, CM, GEMM_BIG_NT, GEMM_BIG_TK, 64, 64>(st, d, Mmax, Nmax, zdim);
  } else if (wgs >= GEMM_SKINNY_BELOW) {
    if (GEMM_ZBATCH_TILE32 && d->zmode && nprob > 1 && deep) {
      // a batch of independent products of very different sizes (the parked weight gradients): 32x32 tiles give 4x the
      // workgroups, which balances the batch over the CUs (dense batch 24.7 -> 17.9 us, token batch 22.8 -> 16.1 us);
      // on a single large product the same tiles are 1-3 us slower than 64x64
      launch_ring<AM, BMODE, CM, 256, GEMM_TK_DEEP, 32, 32>(st, d, Mmax, Nmax, zdim);
    } else if (deep) {
      launch_ring<AM, BMODE, CM, GEMM_MID_NT, GEMM_TK_DEEP, 64, 64>(st, d, Mmax, Nmax, zdim);
    } else {
      launch_ring<AM, BMODE, CM, GEMM_MID_NT, 32, 64, 64>(st, d, Mmax, Nmax, zdim);
    }
  } else if (Nmax >= Mmax) {
    if (deep) launch_ring<AM, BMODE, CM, 256, GEMM_TK_DEEP, 64, 16>(st, d, Mmax, Nmax, zdim);
    else launch_ring<AM, BMODE, CM, 256, 32, 64, 16>(st, d, Mmax, Nmax, zdim);
  } else {
    if (deep) launch_ring<AM, BMODE, CM, 256, GEMM_TK_DEEP, 16, 64>(st, d, Mmax, Nmax, zdim);
    else launch_ring<AM, BMODE, CM, 256, 32, 16, 64>(st, d, Mmax, Nmax, zdim);
  }
  if (S > 1 && !d->defer_second_pass) {
    long elems = (long)Mmax * Nmax;
    dim3 g2((unsigned)((elems + 255) / 256), 1, nprob);
    hipLaunchKernelGGL((gemm_splitk_epilogue<CM>), g2, dim3(256), 0, st, *d, Mmax, Nmax);
  }
  return nasrec_check_launch("gemm");
}

int launch_gemm(hipStream_t st, const nasrec_gemm_desc_t* d) {
  if (d->nseg < 1 || d->nseg > NASREC_MAX_SEGS) return nasrec_set_error(-2, "gemm: nseg=%d out of range", d->nseg);
  const int key = d->amode * 100 + d->bmode * 10 + d->cmode;
  switch (key) {
    case NASREC_AM_KC * 100 + NASREC_AM_KC * 10 + NASREC_CM_PLAIN:  // y = x Wᵀ
      return launch_gemm_t<NASREC_AM_KC, NASREC_AM_KC, NASREC_CM_PLAIN>(st, d);
    case NASREC_AM_KC * 100 + NASREC_AM_RC * 10 + NASREC_CM_PLAIN:  // dx = dy W
      return launch_gemm_t<NASREC_AM_KC, NASREC_AM_RC, NASREC_CM_PLAIN>(st, d);
    case NASREC_AM_RC * 100 + NASREC_AM_RC * 10 + NASREC_CM_PLAIN:  // dW = dyᵀ x
      return launch_gemm_t<NASREC_AM_RC, NASREC_AM_RC, NASREC_CM_PLAIN>(st, d);
    case NASREC_AM_KC * 100 + NASREC_AM_TOKR * 10 + NASREC_CM_TOKJ:  // token-axis y = W x
      if (token_linear_eligible(d)) return launch_token_linear(st, d);  // large batch: weights in LDS, a wavefront per sample
      return launch_gemm_t<NASREC_AM_KC, NASREC_AM_TOKR, NASREC_CM_TOKJ>(st, d);
    case NASREC_AM_RC * 100 + NASREC_AM_TOKR * 10 + NASREC_CM_TOKJ:  // token-axis dx = Wᵀ dy
      if (token_linear_eligible(d)) return launch_token_linear(st, d);
      return launch_gemm_t<NASREC_AM_RC, NASREC_AM_TOKR, NASREC_CM_TOKJ>(st, d);
    case NASREC_AM_TOKK * 100 + NASREC_AM_TOKK * 10 + NASREC_CM_PLAIN:  // token-axis dW = dy xᵀ
      return launch_gemm_t<NASREC_AM_TOKK, NASREC_AM_TOKK, NASREC_CM_PLAIN>(st, d);
    default:
      return nasrec_set_error(-2, "gemm: unsupported operand binding a=%d b=%d c=%d", d->amode, d->bmode, d->cmode);
  }
}
