// Forward Linear with a narrow output at large batch: y[M, N] = x[M, K] W[N, K]^T (+ epilogue), N <= 16, M >= 1024, K in up to
// NASREC_MAX_SEGS k-contiguous segments (binding KC / KC / PLAIN; modules.py:171,340: the dense -> one-token projections of the
// dense nodes, x = the K-concatenated dense outputs of the selected blocks).
//
// The product streams x once (B = 8192, K = 3075: 100 MB) and is nothing else: 0.8 GFLOP, a 197 KB weight panel that lives in L2.
// The general template tiles it 64 x 64 — three quarters of every B tile is padding — stages x through LDS with one-dword loads and
// needs split-K plus a second pass to fill the chip: 43.8 us for the 100 MB (2.3 TB/s).  Here, as in the batch-256 wavefront-per-tile
// bodies (worklist_body.h), the MFMA operands ARE the memory layout:
//   * a workgroup owns 16 rows of x; its NW wavefronts split the k-steps (16 k each, all segments laid end to end) into NW
//     contiguous runs — K is split INSIDE the workgroup, no workspace, no second launch;
//   * lane (r, g) of a wavefront loads x[m0 + r][16 s + 4 g .. + 3] and W[r][16 s + 4 g .. + 3] with one 16-byte buffer load each
//     (rows past M / N clamped, k past the segment's K masked; the extent check of the buffer resource keeps the last row's
//     overrun inside the operand) — eight steps = sixteen loads in flight per lane before their 32 v_mfma_f32_16x16x4_f32
//     (MFMA j of step s sums k = 16 s + 4 g + j);
//   * the NW partial 16 x 16 tiles are added through LDS in a fixed order, wave 0 runs the common epilogue (epilogue_store_col4:
//     bias, activation, saved planes, accumulation — reads first, one wait, stores).
// The summation order is a fixed function of (segment list, NW); NW depends on M alone.
#include <stdlib.h>

#include "gemm_tile.h"

#define SKN_STEPS 8  // k-steps (of 16) fetched before their MFMAs

// BRC: the B operand is row-contiguous (B(n, k) = B[k ldb + n]: the input gradient dx = dy W of a Linear with <= 16 INPUTS, one problem
// of a zmode launch) — its fragment is four dwords a row of W apart instead of one 16-byte load.
template <int NW, bool BRC>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_n_kernel(const nasrec_gemm_desc_t d) {
  __shared__ float red[NW][256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const nasrec_gemm_seg_t& s0 = d.seg[0];
  const int M = s0.M, N = s0.N;
  const int m0 = (int)blockIdx.x * 16;
  const int row = min(m0 + r, M - 1), col = min(r, N - 1);
  // this wave's run of k-steps
  const int nsegk = d.zmode ? 1 : d.nseg;  // (a zmode launch of ONE problem: its segment is the whole K)
  int T = 0;
  for (int q = 0; q < nsegk; ++q) T += (d.seg[q].A && d.seg[q].K > 0) ? (d.seg[q].K + 15) >> 4 : 0;
  const int t0 = (int)((long)T * wave / NW), t1 = (int)((long)T * (wave + 1) / NW);
  // cursor: (segment, first k of the step)
  int fq = 0, fk = 0;
  {
    int skip = t0;
    for (; fq < nsegk; ++fq) {
      const int nt = (d.seg[fq].A && d.seg[fq].K > 0) ? (d.seg[fq].K + 15) >> 4 : 0;
      if (skip < nt) break;
      skip -= nt;
    }
    fk = 16 * skip;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = t0; t < t1; t += SKN_STEPS) {
    f32x4 fa[SKN_STEPS], fb[SKN_STEPS];
    int lim[SKN_STEPS];  // valid k of the step in the slot (<= 0: nothing)
#pragma unroll
    for (int s = 0; s < SKN_STEPS; ++s) {  // unconditional loads — past the run (or in a dead segment) against a null resource — so that the compiler can count them
      const bool in = t + s < t1 && fq < nsegk;
      const nasrec_gemm_seg_t& sg = d.seg[in ? fq : 0];
      const bool live = in && sg.A != nullptr && sg.K > 0;
      const int K = live ? sg.K : 0, kk = fk + 4 * g;
      lim[s] = K - fk;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(sg.A), 0, live ? (int)min(4L * ((long)(M - 1) * sg.lda + K), 0x7fffffffL) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(sg.B), 0, live ? (int)(4 * (BRC ? (long)(K - 1) * sg.ldb + N : (long)(N - 1) * sg.ldb + K)) : 0, 0x00020000);
      fa[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(4 * ((long)row * sg.lda + kk)), 0, 0));
      if (BRC) {
#pragma unroll
        for (int j = 0; j < 4; ++j)  // (k past K: beyond the extent, zeros)
          fb[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, 4 * ((kk + j) * sg.ldb + col), 0, 0));
      } else {
        fb[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, 4 * (col * sg.ldb + kk), 0, 0));
      }
      fk += 16;
      bool next = in && fk >= K;
      if (next) {  // next LIVE segment (uniform; scalar loads only)
        ++fq;
        while (fq < nsegk && !(d.seg[fq].A && d.seg[fq].K > 0)) ++fq;
        fk = 0;
      }
    }
#pragma unroll
    for (int s = 0; s < SKN_STEPS; ++s) {
      const int l = lim[s];
      if (l > 0) {  // (uniform)
        f32x4 a = fa[s], b = fb[s];
        if (l < 16) {  // last step of a segment: the 16-byte loads run into the row's next columns
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = 4 * g + j < l;
            a[j] = ok ? a[j] : 0.f;
            b[j] = ok ? b[j] : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
      }
    }
  }
  // ---- the NW partial tiles, fixed order: pairs, then pairs of pairs, ... (D: row = 4 g + reg, column = lane & 15) -----------------
#pragma unroll
  for (int q = 0; q < 4; ++q) red[wave][q * 64 + lane] = acc[q];
  __syncthreads();
  if (wave == 0) {
    float v4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float p[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) p[w] = red[w][q * 64 + lane];
#pragma unroll
      for (int span = 1; span < NW; span *= 2)
#pragma unroll
        for (int w = 0; w + span < NW; w += 2 * span) p[w] += p[w + span];
      v4[q] = p[0];
    }
    if (m0 + 4 * g < M && r < N) epilogue_store_col4<NASREC_CM_PLAIN>(d, s0, m0 + 4 * g, r, M, v4);
  }
}

// Which launches take this kernel (plan.py mirrors the rule — `skinny_n_eligible` — and gives them splitk = 1)
bool gemm_skinny_n_eligible(const nasrec_gemm_desc_t* d) {
  if (d->amode != NASREC_AM_KC || (d->bmode != NASREC_AM_KC && d->bmode != NASREC_AM_RC) || d->cmode != NASREC_CM_PLAIN || d->splitk > 1) return false;
  if (d->zmode && d->nseg != 1) return false;
  const nasrec_gemm_seg_t& s0 = d->seg[0];
  if (s0.N < 1 || s0.N > 16 || s0.M < 1024) return false;
  long K = 0;
  for (int q = 0; q < d->nseg; ++q) {
    const nasrec_gemm_seg_t& s = d->seg[q];
    if (s.Aaux || s.Baux || s.ones_col || (s.Mvalid > 0 && s.Mvalid < s0.M)) return false;
    if (s.A && s.K > 0) {
      K += s.K;
      // 31-bit byte offsets
      if ((long)s0.M * s.lda >= (1L << 29) || (long)(d->bmode == NASREC_AM_RC ? s.K : s0.N) * s.ldb >= (1L << 29)) return false;
    }
  }
  return K >= 256;
}

int launch_gemm_skinny_n(hipStream_t st, const nasrec_gemm_desc_t* d) {
  const int M = d->seg[0].M;
  const int tiles = (M + 15) / 16;
  const bool brc = d->bmode == NASREC_AM_RC;
  if (tiles >= 512) {
    if (brc) hipLaunchKernelGGL((gemm_skinny_n_kernel<4, true>), dim3((unsigned)tiles), dim3(256), 0, st, *d);
    else hipLaunchKernelGGL((gemm_skinny_n_kernel<4, false>), dim3((unsigned)tiles), dim3(256), 0, st, *d);
  } else {
    if (brc) hipLaunchKernelGGL((gemm_skinny_n_kernel<8, true>), dim3((unsigned)tiles), dim3(512), 0, st, *d);
    else hipLaunchKernelGGL((gemm_skinny_n_kernel<8, false>), dim3((unsigned)tiles), dim3(512), 0, st, *d);
  }
  return nasrec_check_launch("gemm_skinny_n");
}

// ======================================================================================================================================
// The opposite corner (round 6): a WIDE output from a handful of inputs at large batch — y[M, N] = x[M, K] W^T with K <= 16 (the dense
// projections of the raw dense features, modules.py:171 on supernet.py:1137-1145's 13 integer columns: 4096 x 1024 x 13), and the input
// gradients dx = dy W of Linears with <= 16 outputs (KC / RC, one problem per k-segment of a zmode launch: 3 x 4096 x 1024 x 16).  These are
// streaming WRITES (16.8 MB per problem, 0.1 GFLOP): on the 128 x 128 x 32 throughput tile they took 24 - 25 us (0.7 - 2 TB/s) — a k-tile
// that is half padding, staged through LDS behind barriers, for an output that is one pass over memory.  Here a thread owns one column j
// and eight rows: the K <= 16 weights of its column sit in registers (KC: W[j][0..K); RC: W[k][j], coalesced over the lanes), the rows'
// K inputs are wave-uniform (scalar loads), 8 K FMAs, and two epilogue_store_col4 calls write 4 + 4 rows of the column — a wavefront's
// store is 256 contiguous bytes per row.  No LDS, no barrier, no matrix pipe: 4 B of output per 2 K flops is far below any compute roof.
// ======================================================================================================================================
#define TK_MAXK 16
#define TK_GROUPS 2
#define TK_ROWS (8 * TK_GROUPS)
template <bool BRC>
__global__ __launch_bounds__(256) void gemm_tinyk_kernel(const nasrec_gemm_desc_t d) {
  const int q = d.zmode ? (int)blockIdx.z : 0;
  const nasrec_gemm_seg_t& sg = d.seg[q];
  const int M = sg.M, N = sg.N, K = sg.K;
  const int j = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int i0 = (int)blockIdx.y * TK_ROWS;
  if (i0 >= M) return;
  const int jc = min(j, N - 1);
  float w[TK_MAXK];
  if (sg.A != nullptr && K > 0) {
#pragma unroll
    for (int k = 0; k < TK_MAXK; ++k) w[k] = k < K ? (BRC ? sg.B[(long)k * sg.ldb + jc] : sg.B[(long)jc * sg.ldb + k]) : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < TK_MAXK; ++k) w[k] = 0.f;
  }
  // the common epilogue's simple case, decided once: no residual, no gating operand, no accumulation, no prefix mask, bias along the columns
  const bool simple = !d.pre_add && d.mul_nseg == 0 && !(d.zmode ? sg.accumulate : d.beta) && d.dims_in_use < 0 && !(d.bias && d.bias_on_rows);
  const float bj = (simple && d.bias) ? d.bias[jc] : 0.f;
  float* const zp = d.save_z;
  float* const ap = d.save_act;
  const int act = d.act;
  // the rows' inputs, eight rows at a time: 8 x 16 floats for the whole wavefront in TWO vector loads (lane l holds
  // x[i + l / 8][2 (l % 8) + {0, 1}]), handed to every lane through v_readlane (the operand of the FMA is then a scalar register).  Read per
  // lane and element — 128 broadcast loads per thread for 8 stores — the kernel was bound by the issue rate of its vector-memory instructions
  // (22 us for a 16.8 MB output).  TK_GROUPS groups of eight rows share the column's weights.
  const int lane = (int)threadIdx.x & 63;
  // (every group's inputs are loaded before the first store: vmcnt counts loads and stores in one in-order queue, so a load issued behind
  // a group's stores would make the wave wait for those stores to be acknowledged — a store round trip per group)
  float xa[TK_GROUPS], xb[TK_GROUPS];
#pragma unroll
  for (int gr = 0; gr < TK_GROUPS; ++gr) {
    xa[gr] = xb[gr] = 0.f;
    if (sg.A != nullptr && K > 0) {
      const int ri = min(i0 + 8 * gr + (lane >> 3), M - 1), kc = 2 * (lane & 7);
      const float* xr = sg.A + (long)ri * sg.lda;
      if (kc < K) xa[gr] = xr[kc];
      if (kc + 1 < K) xb[gr] = xr[kc + 1];
    }
  }
#pragma unroll
  for (int gr = 0; gr < TK_GROUPS; ++gr) {
    const int ib = i0 + 8 * gr;
    if (ib >= M) break;  // (uniform)
    const float x0 = xa[gr], x1 = xb[gr];
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < TK_MAXK; ++k) {
        const float xv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (k & 1) ? x1 : x0), r * 8 + (k >> 1)));
        s = fmaf(xv, w[k], s);
      }
      acc[r] = s;
    }
    if (j < N) {
      if (simple) {
        // bias / activation / saved planes only: nothing to READ per element, so the eight rows' stores leave back to back (the general
        // epilogue reads, waits for vmcnt(0) — which also waits for the previous group's stores — and only then stores: a store round trip
        // per group of four rows, 8 groups per thread: that, not bandwidth, made the first version take 21 us for 16.8 MB)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (ib + r < M) {
            const long o = (long)(ib + r) * sg.ldc + j;
            float v = acc[r] + bj;
            if (zp) zp[o] = v;
            v = act_apply(v, act);
            if (ap) ap[o] = v;
            sg.C[o] = v;
          }
        }
      } else {
        const float va[4] = {acc[0], acc[1], acc[2], acc[3]}, vb[4] = {acc[4], acc[5], acc[6], acc[7]};
        epilogue_store_col4<NASREC_CM_PLAIN>(d, sg, ib, j, M, va);
        if (ib + 4 < M) epilogue_store_col4<NASREC_CM_PLAIN>(d, sg, ib + 4, j, M, vb);
      }
    }
  }
}

bool gemm_tinyk_eligible(const nasrec_gemm_desc_t* d) {
  static const bool on = getenv("NASREC_TINYK") == nullptr || atoi(getenv("NASREC_TINYK")) != 0;  // A/B knob
  if (!on || d->amode != NASREC_AM_KC || (d->bmode != NASREC_AM_KC && d->bmode != NASREC_AM_RC) || d->cmode != NASREC_CM_PLAIN || d->splitk > 1) return false;
  // (the input-gradient form — KC / RC, a problem per segment — measured 22 - 25 us against the throughput tile's 24: the kernel has it, the rule
  // does not take it unless NASREC_TINYK=2)
  static const bool rc_too = getenv("NASREC_TINYK") != nullptr && atoi(getenv("NASREC_TINYK")) == 2;
  if (d->bmode == NASREC_AM_RC && !rc_too) return false;
  if (!d->zmode && d->nseg != 1) return false;
  for (int q = 0; q < d->nseg; ++q) {
    const nasrec_gemm_seg_t& s = d->seg[q];
    if (s.Aaux || s.Baux || s.ones_col || (s.Mvalid > 0 && s.Mvalid < s.M)) return false;
    if (s.M < 1024 || s.N < 256 || s.K < 0 || s.K > TK_MAXK) return false;
  }
  return true;
}

int launch_gemm_tinyk(hipStream_t st, const nasrec_gemm_desc_t* d) {
  int Mmax = 0, Nmax = 0;
  const int nprob = d->zmode ? d->nseg : 1;
  for (int q = 0; q < nprob; ++q) {
    Mmax = max(Mmax, d->seg[q].M);
    Nmax = max(Nmax, d->seg[q].N);
  }
  const dim3 grid((unsigned)((Nmax + 255) / 256), (unsigned)((Mmax + TK_ROWS - 1) / TK_ROWS), (unsigned)nprob);
  if (d->bmode == NASREC_AM_RC) hipLaunchKernelGGL((gemm_tinyk_kernel<true>), grid, dim3(256), 0, st, *d);
  else hipLaunchKernelGGL((gemm_tinyk_kernel<false>), grid, dim3(256), 0, st, *d);
  return nasrec_check_launch("gemm_tinyk");
}
