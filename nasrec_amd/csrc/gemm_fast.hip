// Throughput-regime fp32 MFMA GEMM for gfx950: the large-batch (supernet) products of the nn.Linear family
// (modules.py:171,489,515,584; supernet.py:1140 forward, and their two autograd products), C(i,j) = sum_k A(i,k) B(j,k).
//
// Same descriptor, bindings KC/KC (y = x W^T), KC/RC (dx = dy W), RC/RC (dW = dy^T x) and epilogue as gemm_kernel
// (gemm_tile.h); that kernel stays the general fallback (token-axis bindings, ReLU-mask operands, small launches).
// What differs is the tiling and the staging, chosen for launches with hundreds of full tiles:
//   * block tile 128 x 128 x 32, 4 wavefronts as 2 x 2, wave tile 64 x 64 = 2 x 2 v_mfma_f32_32x32x2_f32 tiles
//     (64 accumulator registers, 16 MFMAs = 1024 matrix-pipe cycles per 8 k): half the LDS fragment traffic per flop of the
//     64 x 64 block tile, and half the L2 -> LDS traffic;
//   * staging loads are 16-byte BUFFER loads (dword-aligned addresses are enough: weight rows are [N, 13 + 1024 i] floats):
//     resource descriptor = the operand's extent, per-lane byte offset fixed per segment, the k-tile advance in the scalar
//     offset — no per-tile address arithmetic, and the hardware bounds check replaces every edge predicate: rows beyond the
//     operand read other valid bytes of it or 0, and their results are never stored;
//   * LDS double buffering with ONE barrier per k-tile; k-contiguous operands sit in LDS as [row][32 + 4] (ds_read_b128
//     fragments, conflict-free with the 36-float pitch), row-contiguous operands as [k][128] (ds_read_b32 fragments, 32
//     consecutive rows per lane group) — nothing is transposed; both use the same k order inside an 8-deep chunk (lane group
//     g, MFMA j -> k = 8q + 4g + j), which only permutes the fp32 summation identically for A and B;
//   * a wave issues in order and a 32x32x2 MFMA holds the matrix pipe for 64 cycles, so what is issued right behind an MFMA
//     is free and what is issued in a block is not (in-kernel stamps, 1 workgroup per CU: 762 cycles to push 4 waves' staging
//     loads through the CU's address path, 586 to park them in LDS, 445 for descriptor reads, against 4096 matrix cycles per
//     k-tile; 84 VALU instructions of address / select arithmetic cost another ~900).  Hence ONE memory instruction behind
//     one MFMA, pinned with scheduling fences, and no vector ALU work in the loop at all;
//   * the last partial k-tile of a segment takes a synchronous checked path; the virtual ones-column (bias gradient) is a
//     template variant, so ordinary tiles pay nothing for it;
//   * XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2; every XCD gets a contiguous
//     run of the (n fastest) tile order, i.e. a few A row-panels and all B panels, instead of every A panel.
//   * balanced schedule (desc.splitk == NASREC_SPLITK_BALANCED): a CU holds two workgroups, so a launch runs in "rounds" of
//     512 tiles and one whose tile count is not a multiple of 512 idles a large part of the chip in its last round (8 x 1024^2
//     weight gradients = 512 tiles: 142 TFLOP/s; 528 tiles: 94; 320 tiles: 85).  Here the first 512 workgroups share the
//     k-iterations of the 512 + (tiles mod 512) "odd" tiles EQUALLY (a contiguous run of iterations each, i.e. the tail of one
//     tile, whole tiles, the head of another), the remaining tiles — a multiple of 512 — follow one per workgroup.  A run that
//     covers a whole tile finishes it; partial runs dump their accumulators to the workspace and gemm_fast_fixup_kernel sums
//     the pieces of each split tile in ascending k order (fixed association: results do not depend on timing) and applies the
//     epilogue.  No inter-workgroup signalling.
// fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain (no reduced-precision path exists on gfx950).
#include <stdlib.h>
#include "gemm_tile.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define FT_BM 128
#define FT_BN 128
#define FT_BK 32
#define FT_KC_LD 36                        // [row][32 + 4]
#define FT_TILE_FLOATS (FT_BM * FT_KC_LD)  // 4608 >= 32 * 128 (the [k][row] form)
#define FT_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__device__ __forceinline__ void ft_slot(int tid, int it, int& row, int& k) {
  const int idx = tid + 256 * it;
  if (MODE == NASREC_AM_KC) {  // 8 float4 along k per row
    row = idx >> 3;
    k = (idx & 7) << 2;
  } else {                     // 32 float4 along rows per k
    k = idx >> 5;
    row = (idx & 31) << 2;
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ft_rsrc(const float* base, long extent_floats) {
  const long bytes = extent_floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes > 0x7fffffffL ? 0x7fffffff : (int)bytes, 0x00020000);
}

#define FT_SK_WGS 512            // workgroups that share the odd tiles' iterations (2 per CU)
#define FT_SK_PIECES 3           // a share of < 2 tiles touches at most 3 tiles
#define FT_PIECE_FLOATS (FT_BM * FT_BN)

// Order of the tiles of one problem: groups of FT_GROUP_M tile rows, inside a group column by column.  The ~64 workgroups an
// XCD runs at a time then cover 8 rows x 8 columns (8 A panels + 8 B panels in its 4 MB L2) instead of 1.5 rows x 41 columns,
// and an XCD's contiguous run re-reads the B panels once per group instead of once per row: fabric reads of the
// 4096 x 5133 x 1024 product 634 MB -> (see profiles/) against 122 MB of operands.
#define FT_GROUP_M 8
__device__ __forceinline__ void ft_grouped(int t, int tm, int tn, int& by, int& bx) {
  const int per_group = FT_GROUP_M * tn;
  const int grp = t / per_group, first = grp * FT_GROUP_M;
  const int rows = tm - first < FT_GROUP_M ? tm - first : FT_GROUP_M;
  const int r = t - grp * per_group;
  bx = r / rows;
  by = first + (r - bx * rows);
}

// tile index (live tiles, problem-major, then k-split, grouped (m, n) order) -> problem z, k-split ks, tile row / column; false: no such tile
__device__ __forceinline__ bool ft_decode(const nasrec_gemm_desc_t& d, int lin, int S, int tiles_m, int tiles_n, int& z, int& ks, int& by,
                                          int& bx) {
  z = 0;
  if (d.zmode) {
    // a batch of independent problems: the grid holds exactly their LIVE tiles, so the eight contiguous runs the XCDs get
    // carry equal work whatever the mix of problem sizes (a grid padded to Mmax x Nmax handed six XCDs 72 tiles each — more
    // than their 64 workgroup slots — and two XCDs 8)
    int rem = lin, tn = 1, per = 1;
    for (;; ++z) {
      if (z >= d.nseg) return false;
      tn = (d.seg[z].N + FT_BN - 1) / FT_BN;
      per = ((d.seg[z].M + FT_BM - 1) / FT_BM) * tn;
      if (rem < per * S) break;
      rem -= per * S;
    }
    ks = rem / per;
    ft_grouped(rem - ks * per, per / tn, tn, by, bx);
  } else {
    const int per_z = tiles_m * tiles_n;
    ks = lin / per_z;
    ft_grouped(lin - ks * per_z, tiles_m, tiles_n, by, bx);
  }
  return true;
}

// XCD-aware order: ids are dealt round-robin to the 8 XCDs; give every XCD one contiguous run of [0, total) (bijective)
__device__ __forceinline__ int ft_xcd_run(int id, int total) {
  const int xcd = id & 7, q = id >> 3;
  const int chunk = total >> 3, rem = total & 7;
  return xcd * chunk + (xcd < rem ? xcd : rem) + q;
}

// first global k-iteration of share w of W iterations split over FT_SK_WGS workgroups
__device__ __host__ __forceinline__ long ft_share_begin(long W, int w) { return W * w / FT_SK_WGS; }

// epilogue of one finished 128 x 128 tile held in the D layout of v_mfma_f32_32x32x2_f32:
// col = lane & 31, row = 8 * (reg >> 2) + 4 * (lane >> 5) + (reg & 3)
__device__ __forceinline__ void ft_epilogue(const nasrec_gemm_desc_t& d, const nasrec_gemm_seg_t& s0, int m0, int n0, int wm, int wn, int fr,
                                            int fg, const f32x16 (&acc)[2][2], bool acc_in_tile) {
  const int M = s0.M, N = s0.N;
  const int Mv = (s0.Mvalid > 0 && s0.Mvalid < M) ? s0.Mvalid : M;
  // plain product (the common case of the large launches: LayerNorm / the split-K pass own the epilogue): straight stores
  // (acc_in_tile: the accumulators were started from the output tile — the accumulation is already in them)
  const bool plain = acc_in_tile || (!d.bias && !d.pre_add && !d.save_z && !d.save_act && d.act == NASREC_ACT_NONE && d.mul_nseg == 0 &&
                                     d.dims_in_use < 0 && !(d.zmode ? s0.accumulate : d.beta) && !s0.ones_col);
  if (plain) {
    float* Cp = s0.C;
    const int ldc = s0.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3), j = n0 + wn * 64 + b * 32 + fr;
          if (i < M && j < N) Cp[(long)i * ldc + j] = i < Mv ? acc[a][b][r] : 0.f;
        }
    return;
  }
  // general epilogue == epilogue_store<NASREC_CM_PLAIN> element by element (gemm_tile.h), with everything that depends on the
  // column alone looked up once per lane and column: a lane owns 2 columns x 32 rows, and the gating product's segment search
  // (mul_lookup: a scalar loop over up to 8 k-segments) used to run for each of its 64 elements
  const bool acc_c = d.zmode ? s0.accumulate != 0 : d.beta != 0;
  const bool has_pre = d.pre_add != nullptr;
  float* rs = s0.rowsum ? s0.rowsum : d.rowsum_out;
  // per column of the lane: the gating operand's pointer / stride, the bias, the dead-column flag
  const float* mp[2] = {nullptr, nullptr};
  int mld[2] = {0, 0};
  float bias_c[2];
  bool dead_c[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = min(n0 + wn * 64 + b * 32 + fr, N - 1);
    if (d.mul_nseg > 0) {
      for (int q = 0; q < d.mul_nseg; ++q) {
        const int jj = j - d.mul_off[q];
        if (jj >= 0 && jj < d.mul_width[q]) {
          mp[b] = d.mul_ptr[q] ? d.mul_ptr[q] + jj : nullptr;
          mld[b] = d.mul_ld[q];
          break;
        }
      }
    }
    bias_c[b] = d.bias ? d.bias[j] : 0.f;  // (a bias over rows belongs to the token-axis layout: gemm_fast_eligible)
    dead_c[b] = d.dims_in_use >= 0 && !d.mask_on_rows && j >= d.dims_in_use;
  }
  // one element from accumulator to memory: the same operations in the same order whichever way its operands were fetched.  NO load
  // in here: vmcnt counts loads and stores in one in-order queue, so a load between two stores makes the wave wait for every earlier
  // store to be acknowledged — once per element
  float* const Cp = s0.C;
  float* const zp = d.save_z;
  float* const ap = d.save_act;
  const long ldc = s0.ldc;
  const int act = d.act;
  const bool has_bias = d.bias != nullptr, has_mulv = d.mul_nseg > 0;
  const int dead_rows = (d.dims_in_use >= 0 && d.mask_on_rows) ? d.dims_in_use : 0x7fffffff;
  auto finish = [&](int i, int j, int b, float v, float prevv, float mulv, float cvv) {
    const long o = (long)i * ldc + j;
    if (has_pre) v += prevv;
    if (has_bias) v += bias_c[b];
    if (zp) zp[o] = v;
    v = act_apply(v, act);
    if (ap) ap[o] = v;
    if (has_mulv) v *= mulv;
    if (dead_c[b] || i >= dead_rows) v = 0.f;
    if (acc_c) v += cvv;
    Cp[o] = v;
  };
  const int nread = (d.mul_nseg > 0 ? 1 : 0) + (has_pre ? 1 : 0) + (acc_c ? 1 : 0);
  if (nread <= 1 && !s0.ones_col) {
    // ONE array is read (the accumulation target of a dx product, the gating operand, or the residual): all 64 of the lane's values are
    // in flight before its first store.  Sixteen at a time — load, wait, store, and vmcnt counts loads and stores in one queue, so the next
    // sixteen loads wait for the previous stores to be acknowledged — a 4096 x 1024 x 128 dx product took 39.7 us with accumulation against
    // 18.2 us without; a tile alone on its CU (256-tile launches) has nothing to hide four such round trips behind.
    float rd[2][2][16] = {};
    bool live_c[2] = {false, false};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (nread == 0) break;  // (bias / activation / saved planes only: nothing to read)
      const int j = min(n0 + wn * 64 + b * 32 + fr, N - 1);
      const float* base = acc_c ? s0.C + j : has_pre ? d.pre_add + j : mp[b];
      const bool live = base != nullptr;  // (a column outside every gating segment: the loads go to C — unconditional, no branch — and count as 0)
      live_c[b] = live;
      const float* bp = live ? base : s0.C + j;
      const long ld = (acc_c || has_pre || !live) ? s0.ldc : mld[b];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = min(m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3), M - 1);  // (clamped: rows >= M are never stored)
          rd[a][b][r] = bp[(long)i * ld];  // (straight into its register: a select here and the compiler loads one element at a time)
        }
    }
    // ONE wait for the whole batch, spelled out: the stores below sit behind uniform branches (saved planes present or not), the
    // compiler cannot count them, and without this it waits with vmcnt(0) — every earlier store acknowledged — at the first use of each
    // of the 64 loaded registers
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = n0 + wn * 64 + b * 32 + fr;
      if (j >= N) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
          if (i >= M) continue;
          const float x = live_c[b] ? rd[a][b][r] : 0.f;
          finish(i, j, b, i < Mv ? acc[a][b][r] : 0.f, x, x, x);
        }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = n0 + wn * 64 + b * 32 + fr;
    if (j >= N) continue;
    const bool ones_j = s0.ones_col && j == N - 1;
    // Everything the 16 rows of a fragment READ (the gating operand, the residual, the accumulation target) is loaded before the first of
    // their stores (round 4): element by element — load, use, store, and the next load may not pass that store, the arrays could
    // alias — a lane paid a dependent memory round trip per element, 64 per tile (the gated 4096 x 5133 x 1024 product: 96 TFLOP/s
    // against 131 for the plain product of the same shape).  Same arithmetic per element: same bits.
    const bool has_mul = d.mul_nseg > 0 && mp[b] != nullptr;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float mulv[16] = {}, prev[16] = {}, cv[16] = {};
      long off[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) off[r] = min(m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3), M - 1);  // (clamped: rows >= M are never stored)
      // (one uniform branch per ARRAY, sixteen loads inside: a branch per element makes the compiler wait for element r before it issues r + 1)
      if (has_mul) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mulv[r] = mp[b][off[r] * mld[b]];
      }
      if (has_pre) {
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[r] = d.pre_add[off[r] * ldc + j];
      }
      if (acc_c) {  // (the ones-column's lanes read a real element too and drop it)
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[r] = Cp[off[r] * ldc + j];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one wait per batch of sixteen (see above)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
        if (i >= M) continue;
        const float v = i < Mv ? acc[a][b][r] : 0.f;
        if (ones_j) {
          rs[i] = v;
          continue;
        }
        finish(i, j, b, v, prev[r], mulv[r], cv[r]);
      }
    }
  }
}

// sk_tiles > 0: balanced schedule — workgroups [0, FT_SK_WGS) share the sk_T k-iterations of each of the first sk_tiles tiles
template <int AM, int BMODE, bool ONES>
__global__ __launch_bounds__(256, 2) void gemm_fast_kernel(const nasrec_gemm_desc_t d, int Mmax, int Nmax, int tiles_m, int tiles_n,
                                                           int sk_tiles, int sk_T) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][FT_TILE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fg = lane >> 5;
  const int S = d.splitk > 1 ? d.splitk : 1;

  // ---- which piece(s) of work ---------------------------------------------------------------------------------------
  // plain schedule: one (tile, k-split) per workgroup.  Balanced schedule: workgroup w < FT_SK_WGS runs the global k-iterations
  // [W w / 512, W (w + 1) / 512) of the first sk_tiles tiles (W = sk_tiles * sk_T), the others one whole tile each.
  long g = 0, g_end = 1;
  int lin_dp = -1, piece = 0, w_sk = 0;
  if (sk_tiles > 0) {
    if ((int)blockIdx.x < FT_SK_WGS) {
      w_sk = ft_xcd_run(blockIdx.x, FT_SK_WGS);
      const long W = (long)sk_tiles * sk_T;
      g = ft_share_begin(W, w_sk);
      g_end = ft_share_begin(W, w_sk + 1);
    } else {
      lin_dp = sk_tiles + ft_xcd_run(blockIdx.x - FT_SK_WGS, gridDim.x - FT_SK_WGS);
    }
  } else {
    lin_dp = ft_xcd_run(blockIdx.x, gridDim.x);
  }
  for (; g < g_end; ++piece) {
  int lin, pa = 0, pb = 0;  // balanced piece: k-tiles [pa, pb) of tile lin
  if (lin_dp >= 0) {
    lin = lin_dp;
    g = g_end;
  } else {
    lin = (int)(g / sk_T);
    pa = (int)(g - (long)lin * sk_T);
    const long left = g_end - g;
    pb = left < sk_T - pa ? pa + (int)left : sk_T;
    g += pb - pa;
  }
  int z, ks, by, bx;
  if (!ft_decode(d, lin, S, tiles_m, tiles_n, z, ks, by, bx)) continue;
  const nasrec_gemm_seg_t& s0 = d.seg[z];
  const int M = s0.M, N = s0.N;
  const int m0 = by * FT_BM, n0 = bx * FT_BN;
  if (m0 >= M || n0 >= N) continue;

  // ---- k range of this split / piece ------------------------------------------------------------------------------------
  int T = 0;
  if (d.zmode) {
    T = s0.A ? (s0.K + FT_BK - 1) / FT_BK : 0;
  } else {
    for (int q = 0; q < d.nseg; ++q)
      if (d.seg[q].A) T += (d.seg[q].K + FT_BK - 1) / FT_BK;
  }
  const bool sk_piece = lin_dp < 0;
  const int t0 = sk_piece ? pa : (int)((long)T * ks / S), t1 = sk_piece ? pb : (int)((long)T * (ks + 1) / S);
  int s = z, kt = t0;
  if (!d.zmode) {
    s = 0;
    int skip = t0;
    while (s < d.nseg) {
      const int nt = d.seg[s].A ? (d.seg[s].K + FT_BK - 1) / FT_BK : 0;
      if (skip < nt) break;
      skip -= nt;
      ++s;
    }
    kt = skip;
  }

  // A product that only ACCUMULATES into its output (a dx joining a gradient that already holds a contribution: no bias, activation,
  // gating, mask, saved plane) starts its accumulators from the output tile instead of zero: the tile's 64 loads per lane go out at the
  // head of the kernel beside the first operand tile's, and the epilogue is the plain store — read after the k-loop they were a round
  // trip nothing hid on a tile that is alone on its CU (4096 x 1024 x 128 with accumulation: 27.3 us against 18.2 without).  The sum is
  // C + (k-tiles in order) instead of (k-tiles in order) + C: the same fp32 terms, one rounding order for every launch geometry that
  // takes this path (whole tiles of unsplit launches).
  const bool acc_init = S == 1 && lin_dp >= 0 && (d.zmode ? s0.accumulate != 0 : d.beta != 0) && !d.bias && !d.pre_add && !d.save_z &&
                        !d.save_act && d.act == NASREC_ACT_NONE && d.mul_nseg == 0 && d.dims_in_use < 0 && !s0.ones_col &&
                        !(s0.Mvalid > 0 && s0.Mvalid < M);
  f32x16 acc[2][2];
  if (acc_init) {
    const float* Cp = s0.C;
    const long ldc = s0.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // (clamped: what lies outside the product is read and never stored)
          const int i = min(m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3), M - 1), j = min(n0 + wn * 64 + b * 32 + fr, N - 1);
          acc[a][b][r] = Cp[(long)i * ldc + j];
        }
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }

  // ---- staging state (per segment) ------------------------------------------------------------------------------------
  const bool has_ones = ONES && s0.ones_col != 0;  // (ONES = some problem of the launch has the virtual column; this one: has_ones)
  const int Rb = has_ones ? N - 1 : N;             // real rows of B
  const float* pA = nullptr;
  const float* pB = nullptr;
  __amdgpu_buffer_rsrc_t rsA = ft_rsrc(nullptr, 0), rsB = ft_rsrc(nullptr, 0);
  int cK = 0, lda = 0, ldb = 0;
  int stepA = 0, stepB = 0;  // bytes per k-tile
  int voffA[4], voffB[4];    // byte offset of the slot's (row, k) at k-tile 0 (rows beyond the operand: bounds-checked garbage / 0)
  auto load_seg = [&](int sq) {
    const nasrec_gemm_seg_t& sg = d.seg[sq];
    pA = sg.A;
    pB = sg.B;
    cK = sg.K;
    lda = sg.lda;
    ldb = sg.ldb;
    // extents: last element any in-range (row, k) can touch
    rsA = ft_rsrc(pA, AM == NASREC_AM_KC ? (long)(M - 1) * lda + cK : (long)(cK - 1) * lda + M);
    rsB = ft_rsrc(pB, BMODE == NASREC_AM_KC ? (long)(Rb - 1) * ldb + cK : (long)(cK - 1) * ldb + Rb);
    stepA = 4 * FT_BK * (AM == NASREC_AM_KC ? 1 : lda);
    stepB = 4 * FT_BK * (BMODE == NASREC_AM_KC ? 1 : ldb);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int row, k;
      ft_slot<AM>(tid, it, row, k);
      // rows beyond M are redirected to row M - 1 for k-contiguous operands (keeps the offset inside 31 bits for any shape);
      // row-contiguous slots simply run on into the next k-row
      voffA[it] = 4 * (int)(AM == NASREC_AM_KC ? (long)min(m0 + row, M - 1) * lda + k : (long)k * lda + (m0 + row));
      ft_slot<BMODE>(tid, it, row, k);
      voffB[it] = 4 * (int)(BMODE == NASREC_AM_KC ? (long)min(n0 + row, Rb - 1) * ldb + k : (long)k * ldb + (n0 + row));
    }
  };
  // per-thread constants of the ones-column variant: which elements of a B slot belong to column N - 1
  int oneB[4] = {0, 0, 0, 0};
  if (has_ones) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int row, k;
      ft_slot<BMODE>(tid, it, row, k);
      if (BMODE == NASREC_AM_KC) {
        oneB[it] = (n0 + row == N - 1) ? 15 : 0;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) oneB[it] |= (n0 + row + e == N - 1) ? (1 << e) : 0;
      }
    }
  }

  f32x4 ra[4], rb[4];
  auto parkA = [&](int buf, int it, const f32x4& va) {
    float* As = smem[buf][0];
    int row, k;
    ft_slot<AM>(tid, it, row, k);
    if (AM == NASREC_AM_KC)
      *reinterpret_cast<f32x4*>(&As[row * FT_KC_LD + k]) = va;
    else
      *reinterpret_cast<f32x4*>(&As[k * FT_BM + row]) = va;
  };
  auto parkB = [&](int buf, int it, f32x4 vb, int kmask) {  // kmask: bit e set = element e lies inside the segment's K
    float* Bs = smem[buf][1];
    int row, k;
    ft_slot<BMODE>(tid, it, row, k);
    if (ONES) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((oneB[it] >> e) & 1) vb[e] = ((kmask >> e) & 1) ? 1.f : 0.f;
    }
    if (BMODE == NASREC_AM_KC)
      *reinterpret_cast<f32x4*>(&Bs[row * FT_KC_LD + k]) = vb;
    else
      *reinterpret_cast<f32x4*>(&Bs[k * FT_BN + row]) = vb;
  };
  // A partial tile goes global -> LDS in one synchronous, fully checked step (it never shares registers with the asynchronous
  // path: the compiler would merge the two with register copies right behind the vector loads, i.e. wait for every load at once)
  auto commit_tail = [&](int buf, int ktq) {
    const int k0 = ktq * FT_BK;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      f32x4 va, vb;
      int row, k, kmask = 0;
      float xa[4], xb[4];
      // the slot's eight dwords in flight together, THEN the selects (the asm pins all eight as live here: left to itself the compiler
      // — short of registers around this loop — emitted load, wait, select eight times over: 32 dependent round trips per partial
      // k-tile, and a K = 16 product is nothing but its partial tile)
      ft_slot<AM>(tid, it, row, k);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k0 + k + (AM == NASREC_AM_KC ? e : 0);
        const int rr = min(m0 + row + (AM == NASREC_AM_KC ? 0 : e), M - 1);
        xa[e] = pA[operand_offset<AM>(rr, min(kk, cK - 1), lda)];
      }
      ft_slot<BMODE>(tid, it, row, k);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k0 + k + (BMODE == NASREC_AM_KC ? e : 0);
        const int rr = min(n0 + row + (BMODE == NASREC_AM_KC ? 0 : e), Rb - 1);
        xb[e] = pB[operand_offset<BMODE>(rr, min(kk, cK - 1), ldb)];
      }
      asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]));
      ft_slot<AM>(tid, it, row, k);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k0 + k + (AM == NASREC_AM_KC ? e : 0);
        va[e] = kk < cK ? xa[e] : 0.f;
      }
      ft_slot<BMODE>(tid, it, row, k);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k0 + k + (BMODE == NASREC_AM_KC ? e : 0);
        vb[e] = kk < cK ? xb[e] : 0.f;
        if (kk < cK) kmask |= 1 << e;
      }
      parkA(buf, it, va);
      parkB(buf, it, vb, kmask);
    }
  };
  // one operand fragment (4 k-values of chunk q for this lane's row) of the k-tile parked in LDS buffer `buf`; f[j] feeds MFMA j
  auto read_fragA = [&](int buf, int q, int a, float (&f)[4]) {
    const float* As = smem[buf][0];
    const int row = wm * 64 + a * 32 + fr;
    if (AM == NASREC_AM_KC) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&As[row * FT_KC_LD + 8 * q + 4 * fg]);
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = As[(8 * q + 4 * fg + j) * FT_BM + row];
    }
  };
  auto read_fragB = [&](int buf, int q, int b, float (&f)[4]) {
    const float* Bs = smem[buf][1];
    const int row = wn * 64 + b * 32 + fr;
    if (BMODE == NASREC_AM_KC) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[row * FT_KC_LD + 8 * q + 4 * fg]);
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = Bs[(8 * q + 4 * fg + j) * FT_BN + row];
    }
  };

  // ---- main loop: one barrier per k-tile ------------------------------------------------------------------------------
  if (t0 < t1) {
    load_seg(s);
    if ((kt + 1) * FT_BK <= cK) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        ra[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[it], kt * stepA, 0));
        rb[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[it], kt * stepB, 0));
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        parkA(0, it, ra[it]);
        parkB(0, it, rb[it], 15);
      }
    } else {
      commit_tail(0, kt);
    }
    __syncthreads();
  }
#ifdef FT_STAMPS  // diagnostic build only: cycle stamps of one workgroup / wave 0 into d.counters (s_memtime, shader clock)
  long long* stamps = reinterpret_cast<long long*>(d.counters);
  const bool stamp = stamps != nullptr && blockIdx.x == 8 && tid == 0;
  int sn = 0;
#define FT_STAMP() do { if (stamp && sn < 512) stamps[sn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FT_STAMP() do {} while (0)
#endif
  const __amdgpu_buffer_rsrc_t rs_null = ft_rsrc(nullptr, 0);  // num_records 0: every load returns 0 without touching memory
  int seg_tiles = (cK + FT_BK - 1) / FT_BK;  // k-tiles of the current segment (registers: no descriptor reads per tile)
  float f0a[2][4], f0b[2][4], f1a[2][4], f1b[2][4];
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    FT_STAMP();
    ++kt;
    const bool more = t + 1 < t1;
    if (more && kt >= seg_tiles) {  // segment exhausted (rare): next live segment
      if (!d.zmode) {
        do {
          ++s;
        } while (s < d.nseg && (!d.seg[s].A || d.seg[s].K <= 0));
        kt = 0;
      }
      load_seg(s);
      seg_tiles = (cK + FT_BK - 1) / FT_BK;
    }
    const bool next_full = more && (kt + 1) * FT_BK <= cK;
    // without a full next tile the loads still issue (a branch inside the MFMA sequence makes the compiler wait for every
    // earlier load at the block boundary) — against the null resource, and what they park is never read
    const __amdgpu_buffer_rsrc_t curA = next_full ? rsA : rs_null;
    const __amdgpu_buffer_rsrc_t curB = next_full ? rsB : rs_null;
    const int soffA = next_full ? kt * stepA : 0, soffB = next_full ? kt * stepB : 0;
    // fragments of chunk 0 (the only LDS latency a k-tile exposes)
    read_fragA(buf, 0, 0, f0a[0]);
    read_fragA(buf, 0, 1, f0a[1]);
    read_fragB(buf, 0, 0, f0b[0]);
    read_fragB(buf, 0, 1, f0b[1]);
    FT_FENCE();
    FT_STAMP();
#pragma clang loop unroll(full)
    for (int q = 0; q < 4; ++q)
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; ++r) {
        const int i = 16 * q + r, j = (r >> 2) & 3, a = (r >> 1) & 1, b = r & 1;
        if (q & 1)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1a[a][j], f1b[b][j], acc[a][b], 0, 0, 0);
        else
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0a[a][j], f0b[b][j], acc[a][b], 0, 0, 0);
        FT_FENCE();
        if (q < 3 && r >= 1 && r <= 4) {  // fragments of chunk q + 1 into the other register set
          if (q & 1) {
            if (r == 1) read_fragA(buf, q + 1, 0, f0a[0]);
            if (r == 2) read_fragA(buf, q + 1, 1, f0a[1]);
            if (r == 3) read_fragB(buf, q + 1, 0, f0b[0]);
            if (r == 4) read_fragB(buf, q + 1, 1, f0b[1]);
          } else {
            if (r == 1) read_fragA(buf, q + 1, 0, f1a[0]);
            if (r == 2) read_fragA(buf, q + 1, 1, f1a[1]);
            if (r == 3) read_fragB(buf, q + 1, 0, f1b[0]);
            if (r == 4) read_fragB(buf, q + 1, 1, f1b[1]);
          }
        }
#ifndef FT_NO_LOADS
        if (i >= 6 && i < 38 && ((i - 6) & 3) == 0) {  // staging load l of tile t+1 behind MFMAs 6, 10, ... 34
          const int l = (i - 6) >> 2;
          if (l < 4)
            ra[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(curA, voffA[l], soffA, 0));
          else
            rb[l - 4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(curB, voffB[l - 4], soffB, 0));
        }
#endif
#ifndef FT_NO_PARKS
        if (i >= 41 && i <= 55 && ((i - 41) & 1) == 0) {  // park slot w in the other LDS buffer behind MFMAs 41, 43, ... 55
          const int w = (i - 41) >> 1;
          if (w < 4)
            parkA(buf ^ 1, w, ra[w]);
          else
            parkB(buf ^ 1, w - 4, rb[w - 4], 15);
        }
#endif
        FT_FENCE();
      }
    FT_STAMP();
    if (more && !next_full) commit_tail(buf ^ 1, kt);
    FT_STAMP();
    __syncthreads();
  }
  FT_STAMP();

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  if (sk_piece && !(t0 == 0 && t1 == T)) {
    // a partial run of a tile's k-iterations: raw accumulators to this workgroup's piece slot (gemm_fast_fixup_kernel sums)
    float* slot = d.workspace + ((long)w_sk * FT_SK_PIECES + piece) * FT_PIECE_FLOATS;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[((a * 2 + b) * 16 + r) * 256 + tid] = acc[a][b][r];
    continue;  // (the main loop ends on a barrier: the next piece may restage the LDS buffers)
  }
  if (S > 1) {
    const int Mv = (s0.Mvalid > 0 && s0.Mvalid < M) ? s0.Mvalid : M;
    float* slab = d.workspace + ((long)(z * S + ks)) * Mmax * Nmax;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + wm * 64 + a * 32 + 8 * (r >> 2) + 4 * fg + (r & 3), j = n0 + wn * 64 + b * 32 + fr;
          if (i < M && j < N) slab[(long)i * N + j] = i < Mv ? acc[a][b][r] : 0.f;
        }
    continue;
  }
  ft_epilogue(d, s0, m0, n0, wm, wn, fr, fg, acc, acc_init);
  }
}

// second pass of the balanced schedule: one workgroup per odd tile; pieces in ascending k order, then the epilogue
__global__ __launch_bounds__(256) void gemm_fast_fixup_kernel(const nasrec_gemm_desc_t d, int tiles_m, int tiles_n, int sk_tiles, int sk_T) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fg = lane >> 5;
  const int t = blockIdx.x;
  const long W = (long)sk_tiles * sk_T, lo = (long)t * sk_T, hi = lo + sk_T;
  int w = (int)(lo * FT_SK_WGS / W);
  while (w > 0 && ft_share_begin(W, w) > lo) --w;
  while (ft_share_begin(W, w + 1) <= lo) ++w;
  if (ft_share_begin(W, w) <= lo && ft_share_begin(W, w + 1) >= hi) return;  // one workgroup ran the whole tile and finished it
  int z, ks, by, bx;
  if (!ft_decode(d, t, 1, tiles_m, tiles_n, z, ks, by, bx)) return;
  const nasrec_gemm_seg_t& s0 = d.seg[z];
  const int m0 = by * FT_BM, n0 = bx * FT_BN;
  if (m0 >= s0.M || n0 >= s0.N) return;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  for (; w < FT_SK_WGS && ft_share_begin(W, w) < hi; ++w) {
    const int piece = t - (int)(ft_share_begin(W, w) / sk_T);  // tiles this workgroup touched before tile t
    const float* slot = d.workspace + ((long)w * FT_SK_PIECES + piece) * FT_PIECE_FLOATS;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += slot[((a * 2 + b) * 16 + r) * 256 + tid];
  }
  ft_epilogue(d, s0, m0, n0, wm, wn, fr, fg, acc, false);
}

template <int AM, int BMODE>
static int launch_fast_t(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax, int zdim, bool ones) {
  const int tm = (Mmax + FT_BM - 1) / FT_BM, tn = (Nmax + FT_BN - 1) / FT_BN;
  long blocks = (long)tm * tn * zdim;
  if (d->zmode) {  // live tiles only (zdim = problems x split-K)
    const int S = d->splitk > 1 ? d->splitk : 1;
    blocks = 0;
    for (int q = 0; q < d->nseg; ++q)
      blocks += (long)((d->seg[q].M + FT_BM - 1) / FT_BM) * ((d->seg[q].N + FT_BN - 1) / FT_BN) * S;
  }
  int sk_tiles = 0, sk_T = 0;
  if (d->splitk == NASREC_SPLITK_BALANCED) {
    // every tile of the launch must have the same number of k-iterations
    int T = 0;
    if (d->zmode) {
      for (int q = 0; q < d->nseg; ++q) {
        const int tq = d->seg[q].A ? (d->seg[q].K + FT_BK - 1) / FT_BK : 0;
        if (q > 0 && tq != T) return nasrec_set_error(-2, "gemm: balanced schedule needs equal K over the batch (problem %d)", q);
        T = tq;
      }
    } else {
      for (int q = 0; q < d->nseg; ++q)
        if (d->seg[q].A) T += (d->seg[q].K + FT_BK - 1) / FT_BK;
    }
    if (T < 1) return nasrec_set_error(-2, "gemm: balanced schedule on an empty product");
    if (!d->workspace) return nasrec_set_error(-3, "gemm: balanced schedule needs a workspace of NASREC_SK_WORKSPACE_FLOATS floats");
    const long tiles = blocks;
    if (tiles % FT_SK_WGS != 0) {
      sk_tiles = (int)(tiles >= FT_SK_WGS ? FT_SK_WGS + tiles % FT_SK_WGS : tiles);
      sk_T = T;
      blocks = FT_SK_WGS + (tiles - sk_tiles);
    }
  }
  const dim3 grid((unsigned)blocks);
  if (ones)
    hipLaunchKernelGGL((gemm_fast_kernel<AM, BMODE, true>), grid, dim3(256), 0, st, *d, Mmax, Nmax, tm, tn, sk_tiles, sk_T);
  else
    hipLaunchKernelGGL((gemm_fast_kernel<AM, BMODE, false>), grid, dim3(256), 0, st, *d, Mmax, Nmax, tm, tn, sk_tiles, sk_T);
  if (sk_tiles > 0) hipLaunchKernelGGL(gemm_fast_fixup_kernel, dim3((unsigned)sk_tiles), dim3(256), 0, st, *d, tm, tn, sk_tiles, sk_T);
  return 0;
}

// Does this launch belong to the throughput regime?  (plan.py mirrors the rule when it sizes split-K: `_fast_gemm_splitk`.)
bool gemm_fast_eligible(const nasrec_gemm_desc_t* d, int Mmax, int Nmax) {
  if (d->cmode != NASREC_CM_PLAIN) return false;
  if (d->bias && d->bias_on_rows) return false;  // (token-axis layouts only; the epilogue here keeps loads out of its store sequence)
  if ((d->amode != NASREC_AM_KC && d->amode != NASREC_AM_RC) || (d->bmode != NASREC_AM_KC && d->bmode != NASREC_AM_RC)) return false;
  if (d->amode == NASREC_AM_RC && d->bmode == NASREC_AM_KC) return false;  // no call site
  const int nprob = d->zmode ? d->nseg : 1;
  const int S = d->splitk > 1 ? d->splitk : 1;
  long tiles = 0, kmax = 0;
  for (int q = 0; q < d->nseg; ++q) {
    const nasrec_gemm_seg_t& s = d->seg[q];
    if (s.Aaux || s.Baux) return false;  // ReLU-mask operands: general kernel
    if (s.A && s.K > kmax) kmax = s.K;
    // byte offsets of the staging loads are 31-bit: the operands' extents in floats must stay below 2^29
    const long r = s.M > s.N ? s.M : s.N, ld = s.lda > s.ldb ? s.lda : s.ldb;
    if (s.A && (r * ld + s.K >= (1L << 29) || (long)s.K * ld + r >= (1L << 29))) return false;
  }
  long useful = 0, padded = 0;
  for (int q = 0; q < nprob; ++q) {
    const nasrec_gemm_seg_t& s = d->seg[q];
    const long t = (long)((s.M + FT_BM - 1) / FT_BM) * ((s.N + FT_BN - 1) / FT_BN);
    tiles += t * S;
    useful += (long)s.M * s.N;
    padded += t * FT_BM * FT_BN;
  }
  (void)Mmax;
  (void)Nmax;
  // skinny problems (a [1024, 13] weight gradient fills a tenth of its 128 x 128 tiles) belong to the small-tile kernel
  if (2 * useful < padded) return false;
  static const long min_k = getenv("NASREC_FAST_MIN_K") ? atol(getenv("NASREC_FAST_MIN_K")) : 1;  // A/B knob (plan.py mirrors it; 64 until round 4)
  return tiles >= NASREC_GEMM_FAST_MIN_TILES && kmax >= min_k;
}

int launch_gemm_fast(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax, int zdim) {
  bool ones = false;
  for (int q = 0; q < d->nseg; ++q) ones = ones || d->seg[q].ones_col != 0;
  if (d->amode == NASREC_AM_KC && d->bmode == NASREC_AM_KC) return launch_fast_t<NASREC_AM_KC, NASREC_AM_KC>(st, d, Mmax, Nmax, zdim, ones);
  if (d->amode == NASREC_AM_KC && d->bmode == NASREC_AM_RC) return launch_fast_t<NASREC_AM_KC, NASREC_AM_RC>(st, d, Mmax, Nmax, zdim, ones);
  return launch_fast_t<NASREC_AM_RC, NASREC_AM_RC>(st, d, Mmax, Nmax, zdim, ones);
}
