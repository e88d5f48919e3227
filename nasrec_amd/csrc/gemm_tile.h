// Body of the fp32 MFMA GEMM family (see gemm.hip for the launch side): shared by gemm_kernel and by the per-sample
// chain kernel (chain.hip).
#pragma once
#include "common.h"

#ifndef GEMM_TK_DEEP
#define GEMM_TK_DEEP 64
#endif
#ifndef GEMM_XCD_REMAP
#define GEMM_XCD_REMAP 0  // measured: 3.4x less fabric traffic on the big products, no time gain (Infinity Cache serves the re-reads), +6% step time from the index arithmetic
#endif
#ifndef GEMM_ZBATCH_TILE32
#define GEMM_ZBATCH_TILE32 1
#endif
#ifndef GEMM_BIG_NT
#define GEMM_BIG_NT 256
#endif
#ifndef GEMM_BIG_TK
#define GEMM_BIG_TK 32
#endif
#ifndef GEMM_MID_NT
#define GEMM_MID_NT 1024  // threads of the 64x64 latency-regime configuration (A/B on the bench step: 256 -> 0.677, 512 -> 0.646, 1024 -> 0.640 ms)
#endif
#ifndef GEMM_SKINNY_BELOW
#define GEMM_SKINNY_BELOW 128  // launches with fewer 64x64 workgroups than this use the skinny tiles
#endif
#ifndef NASREC_GEMM_FAST_MIN_TILES
#define NASREC_GEMM_FAST_MIN_TILES 120  // launches with at least this many 128x128 tiles (x split-K) take gemm_fast.hip
#endif
bool gemm_fast_eligible(const nasrec_gemm_desc_t* d, int Mmax, int Nmax);
int launch_gemm_fast(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax, int zdim);
bool gemm_kslice_eligible(const nasrec_gemm_desc_t* d);  // gemm_kslice.hip: large forward product at batch ~256, K split inside the workgroup
int launch_gemm_kslice(hipStream_t st, const nasrec_gemm_desc_t* d);
bool gemm_skinny_n_eligible(const nasrec_gemm_desc_t* d);  // gemm_skinny.hip: forward Linear with N <= 16 at large batch, x streamed once
int launch_gemm_skinny_n(hipStream_t st, const nasrec_gemm_desc_t* d);
bool gemm_tinyk_eligible(const nasrec_gemm_desc_t* d);  // gemm_skinny.hip: K <= 16 at large batch, the output streamed once
int launch_gemm_tinyk(hipStream_t st, const nasrec_gemm_desc_t* d);
bool token_linear_eligible(const nasrec_gemm_desc_t* d);  // token_linear.hip: token-axis Linear at large batch
int launch_token_linear(hipStream_t st, const nasrec_gemm_desc_t* d);
bool token_dw_eligible(const nasrec_gemm_desc_t* d);      // token-axis weight gradient at large batch (main pass; slabs -> gemm_splitk_epilogue)
int launch_token_dw(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax);
// Tile configurations (template parameters NT threads, TK staged k depth, TBM x TBN block tile):
//   256 thr, 32, 64x64  4 waves 2x2, 32x32 each — large products, throughput regime (>= 4 workgroups per CU);
//   1024 thr, TK, 64x64  16 waves 4x4, one 16x16 MFMA tile each — four waves per SIMD, one wave's waits hide under
//     the others' MFMAs (GEMM_MID_NT; 512 threads = 8 waves of 32x16 measured 1 % slower, 256 threads 5 % slower);
//   256 thr, TK, 32x32  4 waves, one MFMA tile each — zmode batches of unequal products in the latency regime;
//   256 thr, TK, 64x16 / 16x64  4 waves, one 16x16 MFMA tile each — "skinny" products of the batch-256 step (a
//     [B,n]x[n,16] Linear, a token-axis Linear over B*16 columns): a 64x64 tiling would leave them on 4..64 of the 256
//     CUs, and because every kernel starts on a cold L2 a CU only sustains ~13 KB/us of staging loads (outstanding
//     misses x ~1 us latency) — so the operand traffic has to be spread over as many CUs as the problem allows.
// TK = 32 when no k-segment is deeper than one 32-wide tile, else GEMM_TK_DEEP (fewer round trips and barriers; 64
// measured best, 128 loses to its partial tiles).

// thread -> (row, k) mapping of the staging loads of an R x TK operand tile: lanes run along the contiguous axis
template <int MODE, int NT, int TK, int R>
__device__ __forceinline__ void stage_coords(int tid, int it, int& rr, int& kk) {
  if (MODE == NASREC_AM_KC || MODE == NASREC_AM_TOKK) {
    kk = tid & (TK - 1);
    rr = tid / TK + (NT / TK) * it;
  } else {
    rr = tid & (R - 1);
    kk = tid / R + (NT / R) * it;
  }
}

__device__ __forceinline__ float mul_lookup(const nasrec_gemm_desc_t& d, int i, int j) {
  for (int q = 0; q < d.mul_nseg; ++q) {
    int jj = j - d.mul_off[q];
    if (jj >= 0 && jj < d.mul_width[q]) return d.mul_ptr[q] ? d.mul_ptr[q][(long)i * d.mul_ld[q] + jj] : 0.f;
  }
  return 0.f;
}

// leading dimension of operand A (which = 0) or B (1) of segment sq; only needed on the partial-k-tile path
__device__ __forceinline__ int seg_ld(const nasrec_gemm_desc_t& d, int sq, int which) {
  return which ? d.seg[sq].ldb : d.seg[sq].lda;
}

// One element through the epilogue.  Reads first (residual, bias, gating operand, accumulation target), one explicit wait, then arithmetic
// and stores: see epilogue_store_col4 below for why (the in-order vmcnt queue, and the descriptor in global memory).
template <int CM>
__device__ __forceinline__ void epilogue_store(const nasrec_gemm_desc_t& d, const nasrec_gemm_seg_t& sg, int i, int j,
                                               float v) {
  if (sg.ones_col && j == sg.N - 1) {  // virtual column: row sums of A (bias gradient)
    (sg.rowsum ? sg.rowsum : d.rowsum_out)[i] = v;
    return;
  }
  const long o = c_offset<CM>(i, j, sg.ldc);
  const float* const pre = d.pre_add;
  const float* const bias = d.bias;
  float* const zp = d.save_z;
  float* const ap = d.save_act;
  float* const Cp = sg.C;
  const int act = d.act, dims = d.dims_in_use, nmul = d.mul_nseg;
  const bool mrow = d.mask_on_rows != 0;
  const bool acc_c = (d.zmode ? sg.accumulate : d.beta) != 0;
  // Each operand that is present is loaded behind its own uniform branch; the asm below pins all four values as live at one point, so
  // the compiler can neither sink a load to its use nor merge `if (p) load` with the later `if (p) v += ...` into one block with a wait in
  // it (it did: four serial round trips).  (Reading an absent operand at a dummy address instead — C[o] — made the one-pass K = 1565
  // product's tail wait for a cold line of its own output: 13.4 -> 13.7 us.)
  const float* mp = nullptr;
  long mo = 0;
  for (int q = 0; q < nmul; ++q) {
    const int jj = j - d.mul_off[q];
    if (jj >= 0 && jj < d.mul_width[q]) {
      mp = d.mul_ptr[q];
      mo = (long)i * d.mul_ld[q] + jj;
      break;
    }
  }
  float pv = 0.f, bv = 0.f, mraw = 0.f, cv = 0.f;
  if (pre) pv = pre[o];
  if (bias) bv = bias[d.bias_on_rows ? i : j];
  if (mp) mraw = mp[mo];
  if (acc_c) cv = Cp[o];
  asm volatile("" : "+v"(pv), "+v"(bv), "+v"(mraw), "+v"(cv));
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
  const float mv = mp ? mraw : 0.f;  // (mul_lookup: a column outside every segment, or a null segment, multiplies by 0)
  if (pre) v += pv;
  if (bias) v += bv;
  if (zp) zp[o] = v;
  v = act_apply(v, act);
  if (ap) ap[o] = v;
  if (nmul > 0) v *= mv;
  if (dims >= 0 && (mrow ? i : j) >= dims) v = 0.f;
  if (acc_c) v += cv;
  Cp[o] = v;
}

// The four elements a lane holds of one 16 x 16 MFMA D tile — rows i0 .. i0 + 3 of column j — through the same epilogue, element for
// element the same arithmetic in the same order as epilogue_store (bit-identical), but laid out for the memory pipeline:
//   * the descriptor's fields are read ONCE (it lives in global memory: after every store the compiler has to assume it changed and
//     re-reads each field — a scalar load and a wait per field and element);
//   * everything the four elements READ (residual, bias, gating operand, accumulation target) is in flight together, ONE explicit wait,
//     then arithmetic and stores only: vmcnt counts loads and stores in one in-order queue, so a load issued behind a store makes the
//     wave wait for that store's acknowledgement — element by element that was a store round trip per element; and because the stores
//     sit behind uniform branches the compiler cannot count them, hence the spelled-out wait.
template <int CM>
__device__ __forceinline__ void epilogue_store_col4(const nasrec_gemm_desc_t& d, const nasrec_gemm_seg_t& sg, int i0, int j, int M, const float (&v4)[4]) {
  if (sg.ones_col && j == sg.N - 1) {  // virtual column: row sums of A (bias gradient)
    float* rs = sg.rowsum ? sg.rowsum : d.rowsum_out;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (i0 + r < M) rs[i0 + r] = v4[r];
    return;
  }
  const float* const pre = d.pre_add;
  const float* const bias = d.bias;
  float* const zp = d.save_z;
  float* const ap = d.save_act;
  float* const Cp = sg.C;
  const int act = d.act, dims = d.dims_in_use, ldc = sg.ldc, nmul = d.mul_nseg;
  const bool brow = d.bias_on_rows != 0, mrow = d.mask_on_rows != 0;
  const bool acc_c = (d.zmode ? sg.accumulate : d.beta) != 0;
  const float* mp = nullptr;  // the gating operand's column (mul_lookup's segment search, once)
  long mld = 0;
  for (int q = 0; q < nmul; ++q) {
    const int jj = j - d.mul_off[q];
    if (jj >= 0 && jj < d.mul_width[q]) {
      mp = d.mul_ptr[q] ? d.mul_ptr[q] + jj : nullptr;
      mld = d.mul_ld[q];
      break;
    }
  }
  long o[4];
  int ic[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ic[r] = min(i0 + r, M - 1);  // (clamped: rows >= M are read, never stored)
    o[r] = c_offset<CM>(ic[r], j, ldc);
  }
  float pv[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
  if (pre) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pv[r] = pre[o[r]];
  }
  if (bias) {
    if (brow) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bias[ic[r]];
    } else {
      const float bj = bias[j];
      bv[0] = bv[1] = bv[2] = bv[3] = bj;
    }
  }
  if (mp) {
#pragma unroll
    for (int r = 0; r < 4; ++r) mv[r] = mp[(long)ic[r] * mld];
  }
  if (acc_c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) cv[r] = Cp[o[r]];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (i0 + r >= M) continue;
    float v = v4[r];
    if (pre) v += pv[r];
    if (bias) v += bv[r];
    if (zp) zp[o[r]] = v;
    v = act_apply(v, act);
    if (ap) ap[o[r]] = v;
    if (nmul > 0) v *= mv[r];
    if (dims >= 0 && (mrow ? i0 + r : j) >= dims) v = 0.f;
    if (acc_c) v += cv[r];
    Cp[o[r]] = v;
  }
}

// second pass of split-K: fixed-order sum of the partial slabs, then the same epilogue (element e of problem z)
template <int CM>
__device__ __forceinline__ void splitk_second_pass(const nasrec_gemm_desc_t& d, int Mmax, int Nmax, int z, long e) {
  const int S = d.splitk;
  const nasrec_gemm_seg_t& sg = d.seg[d.zmode ? z : 0];
  const int M = sg.M, N = sg.N;
  if (e >= (long)M * N) return;
  int i, j;
  if (CM == NASREC_CM_TOKJ) {  // token-axis outputs: consecutive threads walk e (16 contiguous floats), then i
    int jb = (int)(e / ((long)M * 16));
    int rem = (int)(e % ((long)M * 16));
    i = rem >> 4;
    j = jb * 16 + (rem & 15);
  } else {
    i = (int)(e / N);
    j = (int)(e % N);
  }
  const float* slab = d.workspace + ((long)z * S) * Mmax * Nmax + (long)i * N + j;
  const long stride = (long)Mmax * Nmax;
  // four independent partial sums keep several slab loads in flight (every kernel starts on a cold L2); the
  // association order is still a fixed function of S, so results stay reproducible
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  int q = 0;
  for (; q + 16 <= S; q += 16) {  // sixteen slabs in flight (same chains: 32 slabs were eight dependent trips of four loads)
    float a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = slab[(long)(q + u) * stride];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      v0 += a[u];
      v1 += a[u + 1];
      v2 += a[u + 2];
      v3 += a[u + 3];
    }
  }
  for (; q + 4 <= S; q += 4) {
    const float a0 = slab[(long)q * stride], a1 = slab[(long)(q + 1) * stride];
    const float a2 = slab[(long)(q + 2) * stride], a3 = slab[(long)(q + 3) * stride];
    v0 += a0;
    v1 += a1;
    v2 += a2;
    v3 += a3;
  }
  for (; q < S; ++q) v0 += slab[(long)q * stride];
  epilogue_store<CM>(d, sg, i, j, (v0 + v1) + (v2 + v3));
}

// One workgroup's share of a GEMM launch: block tile (bx, by) of problem / k-split bz.  Called by gemm_kernel with the
// hardware block index, and by the per-sample chain kernel (chain.hip) with bx = the sample.
template <int AM, int BMODE, int CM, int NT, int TK, int TBM, int TBN>
__device__ __forceinline__ void gemm_tile(const nasrec_gemm_desc_t& d, int Mmax, int Nmax, const int bx_, const int by_, const int bz_) {
  constexpr int LDS_LD = TK + 4;  // +4 pad: rows stay 16-byte aligned for ds_read_b128 and spread over banks
  constexpr int NITA = TBM * TK / NT, NITB = TBN * TK / NT;  // staging loads per thread
  constexpr int NITX = NITA > NITB ? NITA : NITB;
  constexpr int PER_WAVE = (TBM / 16) * (TBN / 16) / (NT / 64);  // 16x16 MFMA tiles per wave
  constexpr int WTM = (PER_WAVE >= 2 && TBM >= 32) ? 32 : 16;    // wave tile
  constexpr int WTN = 16 * PER_WAVE / (WTM / 16);
  constexpr int FA = WTM / 16, FB = WTN / 16;
  static_assert(NITA >= 1 && NITB >= 1 && PER_WAVE >= 1 && (TBM / WTM) * (TBN / WTN) == NT / 64, "tile configuration");
  __shared__ __attribute__((aligned(16))) float As[TBM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[TBN * LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / (TBN / WTN), wn = wave % (TBN / WTN);
  const int fr = lane & 15, fg = lane >> 4;
  const int S = d.splitk > 1 ? d.splitk : 1;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  Give
  // every XCD one CONTIGUOUS run of the (n fastest, then m, then problem/split) tile order, so the workgroups that share
  // an A row-panel or a k-split fetch it into one L2 instead of eight.
  int bx = bx_, by = by_, bz = bz_;
#if GEMM_XCD_REMAP
  {
    const int gx = gridDim.x, gy = gridDim.y;  // (only meaningful when called with the hardware block index)
    const int total = gx * gy * (int)gridDim.z;
    const int lin = bx + gx * (by + gy * bz);
    const int xcd = lin & 7, q = lin >> 3;
    const int chunk = total >> 3, rem = total & 7;
    const int lp = xcd * chunk + (xcd < rem ? xcd : rem) + q;
    bx = lp % gx;
    by = (lp / gx) % gy;
    bz = lp / (gx * gy);
  }
#endif
  const int z = d.zmode ? bz / S : 0;
  const int ks = bz % S;
  const nasrec_gemm_seg_t& s0 = d.seg[z];
  const int M = s0.M, N = s0.N;
  const int m0 = by * TBM, n0 = bx * TBN;
  if (m0 >= M || n0 >= N) return;

  // live k-tiles of this problem and the range owned by this split
  int T = 0;
  if (d.zmode) {
    T = s0.A ? (s0.K + TK - 1) / TK : 0;
  } else {
    for (int q = 0; q < d.nseg; ++q)
      if (d.seg[q].A) T += (d.seg[q].K + TK - 1) / TK;
  }
  const int t0 = (int)((long)T * ks / S), t1 = (int)((long)T * (ks + 1) / S);
  int s = z, kt = t0;
  if (!d.zmode) {
    s = 0;
    int skip = t0;
    while (s < d.nseg) {
      int nt = d.seg[s].A ? (d.seg[s].K + TK - 1) / TK : 0;
      if (skip < nt) break;
      skip -= nt;
      ++s;
    }
    kt = skip;
  }

  f32x4 acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Staging loads.  The per-iteration instruction budget decides this kernel at small batch (one wave per SIMD: every
  // VALU instruction costs >= 4 cycles), so everything loop-invariant is hoisted to segment entry:
  //   * per slot: a 32-bit BYTE offset of (row, kk) relative to the segment base (rows outside the operand are
  //     redirected to row 0 and zeroed at commit), the LDS slot, the row-valid predicate;
  //   * per k-tile only the UNIFORM base pointer advances (scalar adds), so a load is `global_load_dword v, voff, s[base]`
  //     with no vector address arithmetic at all;
  //   * k bounds matter only in the last tile of a segment -> uniform branch to a checked path;
  //   * the ReLU-mask operand and the virtual ones-column sit behind uniform branches.
  const char* cA = nullptr;
  const char* cB = nullptr;
  const char* cAaux = nullptr;
  const char* cBaux = nullptr;
  int cK = 0, cOnes = 0, cur = -1;
  long stepA = 0, stepB = 0;          // bytes per k-tile
  unsigned voffA[NITX], voffB[NITX];  // byte offset of (row, kk) at k-tile 0
  bool rvA[NITX], rvB[NITX], oneB[NITX];
  bool edgeA = false, edgeB = false;  // any row of this tile outside the operand?
  auto load_seg = [&](int sq) {
    const nasrec_gemm_seg_t& sg = d.seg[sq];
    cA = reinterpret_cast<const char*>(sg.A);
    cB = reinterpret_cast<const char*>(sg.B);
    cAaux = reinterpret_cast<const char*>(sg.Aaux);
    cBaux = reinterpret_cast<const char*>(sg.Baux);
    cK = sg.K;
    cOnes = sg.ones_col;
    const int lda = sg.lda, ldb = sg.ldb;
    const int Ra = (sg.Mvalid > 0 && sg.Mvalid < M) ? sg.Mvalid : M;
    const int Rb = cOnes ? N - 1 : N;
    // every addressing mode is linear in k across k-tiles (TK is a multiple of 16)
    stepA = 4 * operand_offset<AM>(0, TK, lda);
    stepB = 4 * operand_offset<BMODE>(0, TK, ldb);
    edgeA = (m0 + TBM > Ra);
    edgeB = (n0 + TBN > Rb);
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      int rr, kk;
      if (it < NITA) {
        stage_coords<AM, NT, TK, TBM>(tid, it, rr, kk);
        rvA[it] = (m0 + rr) < Ra;
        voffA[it] = 4u * (unsigned)operand_offset<AM>(rvA[it] ? m0 + rr : 0, kk, lda);
      }
      if (it < NITB) {
        stage_coords<BMODE, NT, TK, TBN>(tid, it, rr, kk);
        rvB[it] = (n0 + rr) < Rb;
        oneB[it] = cOnes && (n0 + rr == N - 1);
        voffB[it] = 4u * (unsigned)operand_offset<BMODE>(rvB[it] ? n0 + rr : 0, kk, ldb);
      }
    }
    cur = sq;
  };
  // fetch() only ISSUES loads; commit() applies predicates and parks the tile in LDS one iteration later.
  f32x4 ra[(NITX + 3) / 4], rb[(NITX + 3) / 4], xa[(NITX + 3) / 4], xb[(NITX + 3) / 4];  // staged values, 4 slots per vector register group
  bool hasAaux = false, hasBaux = false, ktail = false;
  int tailK = 0;  // valid k in a tail tile
  auto fetch = [&](int ktq) {
    const int k0 = ktq * TK;
    const char* pa = cA + (long)ktq * stepA;
    const char* pb = cB + (long)ktq * stepB;
    ktail = (k0 + TK > cK);
    tailK = cK - k0;
    hasAaux = cAaux != nullptr;
    hasBaux = cBaux != nullptr;
    if (!ktail) {
#pragma unroll
      for (int it = 0; it < NITX; ++it) {
        if (it < NITA) ra[it >> 2][it & 3] = *reinterpret_cast<const float*>(pa + voffA[it]);
        if (it < NITB) rb[it >> 2][it & 3] = *reinterpret_cast<const float*>(pb + voffB[it]);
      }
      if (hasAaux) {
        const char* xp = cAaux + (long)ktq * stepA;
#pragma unroll
        for (int it = 0; it < NITA; ++it) xa[it >> 2][it & 3] = *reinterpret_cast<const float*>(xp + voffA[it]);
      }
      if (hasBaux) {
        const char* xp = cBaux + (long)ktq * stepB;
#pragma unroll
        for (int it = 0; it < NITB; ++it) xb[it >> 2][it & 3] = *reinterpret_cast<const float*>(xp + voffB[it]);
      }
    } else {
      // last (partial) k-tile of the segment: k beyond K is redirected to kk = 0 of the slot's row and zeroed at commit
#pragma unroll
      for (int it = 0; it < NITX; ++it) {
        int rr, kk;
        if (it < NITA) {
          stage_coords<AM, NT, TK, TBM>(tid, it, rr, kk);
          const unsigned oa = (kk < tailK) ? voffA[it] : voffA[it] - 4u * (unsigned)operand_offset<AM>(0, kk, seg_ld(d, cur, 0));
          ra[it >> 2][it & 3] = *reinterpret_cast<const float*>(pa + oa);
          if (hasAaux) xa[it >> 2][it & 3] = *reinterpret_cast<const float*>(cAaux + (long)ktq * stepA + oa);
        }
        if (it < NITB) {
          stage_coords<BMODE, NT, TK, TBN>(tid, it, rr, kk);
          const unsigned ob = (kk < tailK) ? voffB[it] : voffB[it] - 4u * (unsigned)operand_offset<BMODE>(0, kk, seg_ld(d, cur, 1));
          rb[it >> 2][it & 3] = *reinterpret_cast<const float*>(pb + ob);
          if (hasBaux) xb[it >> 2][it & 3] = *reinterpret_cast<const float*>(cBaux + (long)ktq * stepB + ob);
        }
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      int rr, kk;
      if (it < NITA) {
        float a = ra[it >> 2][it & 3];
        if (hasAaux) a = (xa[it >> 2][it & 3] > 0.f) ? a : 0.f;
        if (edgeA) a = rvA[it] ? a : 0.f;
        stage_coords<AM, NT, TK, TBM>(tid, it, rr, kk);
        if (ktail) a = (kk < tailK) ? a : 0.f;
        As[rr * LDS_LD + kk] = a;
      }
      if (it < NITB) {
        float b = rb[it >> 2][it & 3];
        if (hasBaux) b = (xb[it >> 2][it & 3] > 0.f) ? b : 0.f;
        if (edgeB) b = rvB[it] ? b : 0.f;
        stage_coords<BMODE, NT, TK, TBN>(tid, it, rr, kk);
        if (ktail) b = (kk < tailK) ? b : 0.f;
        if (cOnes && oneB[it]) b = (!ktail || kk < tailK) ? 1.f : 0.f;
        Bs[rr * LDS_LD + kk] = b;
      }
    }
  };

  if (t0 < t1) {
    load_seg(s);
    fetch(kt);
  }
  for (int t = t0; t < t1; ++t) {
    __syncthreads();
    commit();
    __syncthreads();
    // advance to the next live tile and prefetch it while the MFMAs run
    ++kt;
    if (!d.zmode) {
      while (s < d.nseg && (!d.seg[s].A || kt * TK >= d.seg[s].K)) {
        ++s;
        kt = 0;
      }
    }
    if (t + 1 < t1) {
      if (s != cur) load_seg(s);
      fetch(kt);
    }

#pragma unroll
    for (int kb = 0; kb < TK / 16; ++kb) {
      f32x4 af[FA], bf[FB];
#pragma unroll
      for (int a = 0; a < FA; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[(wm * WTM + a * 16 + fr) * LDS_LD + kb * 16 + 4 * fg]);
#pragma unroll
      for (int b = 0; b < FB; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WTN + b * 16 + fr) * LDS_LD + kb * 16 + 4 * fg]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < FA; ++a)
#pragma unroll
          for (int b = 0; b < FB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
    }
  }

  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  if (S > 1) {
    // split-K: park the partial tile in this split's slab; gemm_splitk_epilogue sums the slabs in fixed order
    // (deterministic) with one thread per output element.  An in-kernel "last arriver reduces" variant (agent-scope
    // release/acquire, or sc1 write-through slabs) was measured 1.3-2x slower at these sizes: the serial tail of one
    // workgroup reading S slabs costs more than the extra ~5 us launch of a fully parallel second pass.
    float* slab = d.workspace + ((long)(z * S + ks)) * Mmax * Nmax;
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int i = m0 + wm * WTM + a * 16 + 4 * fg + r, j = n0 + wn * WTN + b * 16 + fr;
          if (i < M && j < N) slab[(long)i * N + j] = acc[a][b][r];
        }
    return;
  }
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      const int i0 = m0 + wm * WTM + a * 16 + 4 * fg, j = n0 + wn * WTN + b * 16 + fr;
      const float v4[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (i0 < M && j < N) epilogue_store_col4<CM>(d, s0, i0, j, M, v4);
    }
}

