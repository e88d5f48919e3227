// Device side of NASREC_OP_WORKLIST (see worklist.hip): the kernel and its item bodies.
#pragma once
#include "attention_tok.h"
#include "gemm_rt.h"
#include "final_bodies.h"
#include "interact_bodies.h"
#include "dedup_bodies.h"

#define WL_LDS_BIG_FLOATS 9888  // 39.5 KB, 4 workgroups per CU: the Transformer backward (MHA_TOK_BWD_LDS_FLOATS)
#define WL_LDS_FLOATS 7840  // 31 KB (5 workgroups per CU): the DotProduct cores up to k1 = 48; a 64x16x64 GEMM tile needs 5568, the Transformer forward 3744
static_assert(MHA_TOK_BWD_LDS_FLOATS <= WL_LDS_BIG_FLOATS && MHA_TOK_FWD_LDS_FLOATS <= WL_LDS_FLOATS, "Transformer bodies fit the worklist launches");

#define WL_TK 64  // staged k depth of every GEMM item (a product with K <= 32 pads its one tile with zeros: same sums, bit for bit)
// tile configurations of GEMM items: geom[2] = tile | binding pair << 2 | mask operand << 4
#ifndef WL_AUX_RING2
#define WL_AUX_RING2 0  // (A/B knob: ring depth 2 for the mask-operand instantiations too)
#endif
enum { WL_TOKS = 0, WL_T32x32 = 1, WL_T64x16 = 2, WL_T16x64 = 3 };  // WL_TOKS: wl_token_fwd (a wavefront per sample)

// All bodies share the launch's DYNAMIC LDS buffer (sized by the launcher: 31 KB, or 39.5 KB when a Transformer backward is in the level).
extern __shared__ __attribute__((aligned(16))) float wl_lds[];

// Taking the address of the by-value kernel argument (`wl.blob + off` bound to a reference through a cast) makes clang copy all 4 KB of
// it into scratch, per thread, at kernel entry (a step 8x slower).  The kernel therefore reads the kernel-argument segment pointer
// itself, items are 64-bit ADDRESSES, and a body turns its address into a uniform pointer to constant memory (scalar loads).
// Every body is inlined: as real functions each call saved ~100 callee-saved registers in scratch.
template <typename T>
__device__ __forceinline__ const T& wl_ref(unsigned long long addr) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
  return *(const T*)(const __attribute__((address_space(4))) T*)(((unsigned long long)hi << 32) | lo);
}

template <bool KCA, bool KCB, int TBM, int TBN, bool AUX>
__device__ __forceinline__ void wl_gemm_cfg(unsigned long long blob, int vb_, int gx_, int gy_) {
  // ring depth 2: the kernel must fit 128 registers so that FOUR workgroups of different items share a CU (a level's items only run
  // side by side if they are resident together); a staged k-tile is <= 20 floats per thread on these tiles.  With mask operands a
  // staged element is two registers: those instantiations spilled 37 - 90 registers at depth 2 (a 128 x 768 x 256 weight gradient
  // behind a ReLU took 20 us as an item against 10 us with registers to spare) and stage one k-tile at a time instead.
  constexpr int RING = (AUX && !WL_AUX_RING2) ? 1 : 2;
  const nasrec_gemm_desc_t& g = wl_ref<nasrec_gemm_desc_t>(blob);
  const int vb = __builtin_amdgcn_readfirstlane(vb_), gx = __builtin_amdgcn_readfirstlane(gx_), gy = __builtin_amdgcn_readfirstlane(gy_);
  const int nprob = g.zmode ? g.nseg : 1;
  int Mmax = 0, Nmax = 0;
  for (int q = 0; q < nprob; ++q) {
    Mmax = max(Mmax, g.seg[q].M);
    Nmax = max(Nmax, g.seg[q].N);
  }
  const int bx = vb % gx, by = (vb / gx) % gy, bz = vb / (gx * gy);
  gemm_tile_rt<KCA, KCB, 256, WL_TK, TBM, TBN, AUX, RING>(g, Mmax, Nmax, bx, by, bz, wl_lds);
}

template <bool KCA, bool KCB, bool AUX>
__device__ __forceinline__ void wl_gemm_tile(int tile, unsigned long long off, int vb, int gx, int gy) {
  switch (tile) {
    case WL_T32x32: wl_gemm_cfg<KCA, KCB, 32, 32, AUX>(off, vb, gx, gy); break;
    case WL_T64x16: wl_gemm_cfg<KCA, KCB, 64, 16, AUX>(off, vb, gx, gy); break;
    default: wl_gemm_cfg<KCA, KCB, 16, 64, AUX>(off, vb, gx, gy); break;
  }
}

template <bool AUX>
__device__ __forceinline__ void wl_gemm_bind(int bind, int tile, unsigned long long off, int vb, int gx, int gy) {
  // which axis of each operand is contiguous: the six bindings of the step fall into three pairs (the launcher rejects the fourth)
  if (bind == 0) wl_gemm_tile<true, true, AUX>(tile, off, vb, gx, gy);          // x W^T, token-axis dW
  else if (bind == 1) wl_gemm_tile<true, false, AUX>(tile, off, vb, gx, gy);    // dy W, token-axis W x
  else wl_gemm_tile<false, false, AUX>(tile, off, vb, gx, gy);                  // dy^T x, token-axis W^T dy
}

// Pull `bytes` of the launch's argument blob, starting at `addr`, through the scalar cache behind ONE wait (one asm block: its
// registers are dead afterwards).  A body that walks a descriptor's k-segments one after the other reads each segment's fields when
// it gets there — and every new segment is a cold line of kernel-argument memory, 1.6 us per segment measured in wl_token_fwd.
__device__ __forceinline__ void wl_warm_blob(unsigned long long addr, int bytes) {
  int sink, off;
  asm volatile(
      "s_mov_b32 %1, 0\n"
      "1:\n"
      "s_load_dword %0, %2, %1\n"
      "s_add_u32 %1, %1, 64\n"
      "s_cmp_lt_u32 %1, %3\n"
      "s_cbranch_scc1 1b\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(sink), "=&s"(off)
      : "s"(addr), "s"(bytes)
      : "scc", "memory");
}

// Token-axis Linear forward, a wavefront per (sample, 16 rows of W): out[b][i][e] = sum_k W[i][k] x[b][k][e] (binding KC / TOKR / TOKJ,
// modules.py:222-234, 358-361, 648-650).  On the general tile these products (M = 8 .. 64 rows of W against N = 16 B token columns,
// K = 26 .. 234 tokens in up to four segments) are the slowest items of four forward levels (10 - 13 us each): a 64 x 16 or 16 x 64
// strip stages both operands through LDS with one-dword loads that touch four samples per wave-instruction.  But for one sample the
// B operand of v_mfma_f32_16x16x4_f32 IS the sample's memory: lane (e, g) of MFMA j of a 16-token step needs x[b][16 s + 4 g + j][e],
// a dword per lane with 16 consecutive e per token, and lane (i, g) needs W[i][16 s + 4 g + j], one 16-byte load for the four j.
// No LDS, no barrier; loads run KT steps ahead of the MFMAs.  The k-grouping (MFMA j of step s sums k = 16 s + 4 g + j, g = 0..3;
// steps and segments in order) is exactly the general tile's, so the result is bit-identical to it.
#define WL_TOK_DEPTH 4
__device__ __forceinline__ void wl_token_fwd(unsigned long long blob, int vb_, int MT_) {
  const nasrec_gemm_desc_t& g = wl_ref<nasrec_gemm_desc_t>(blob);
  const int vb = __builtin_amdgcn_readfirstlane(vb_), MT = __builtin_amdgcn_readfirstlane(MT_);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = lane & 15, fg = lane >> 4;
  wl_warm_blob(blob, (int)offsetof(nasrec_gemm_desc_t, seg) + g.nseg * (int)sizeof(nasrec_gemm_seg_t));
  const nasrec_gemm_seg_t& s0 = g.seg[0];
  const int M = s0.M, B = s0.N >> 4;
  const int unit = vb * 4 + wave;  // (sample, row block)
  const int b = unit / MT, mt = unit - b * MT;
  if (b >= B) return;
  const int row = min(mt * 16 + e, M - 1);  // (lane & 15 doubles as the row of the A fragment; rows beyond M are never stored)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 fa[WL_TOK_DEPTH];
  float fb[WL_TOK_DEPTH][4];
  int lim[WL_TOK_DEPTH];  // valid k of the step in the slot (<= 0: nothing)
  // Cursor of the fetches: (segment, first k of the step).  A fetch is five UNCONDITIONAL buffer loads — past the last segment (or in
  // a dead one) against a null resource, which returns zeros and moves nothing: with a branch around the loads the compiler cannot
  // count what is in flight and waits for everything before every MFMA group (measured: one full memory latency per step).
  int fq = 0, fk = 0;
  auto fetch = [&](int slot) {
    const bool in = fq < g.nseg;
    const nasrec_gemm_seg_t& sg = g.seg[in ? fq : 0];
    const bool live = in && sg.A != nullptr && sg.K > 0;
    const int K = live ? sg.K : 0, kk = fk + 4 * fg;
    lim[slot] = K - fk;
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A), 0, live ? (int)(4 * ((long)(M - 1) * sg.lda + K)) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.B + (long)b * sg.ldb), 0, live ? 64 * K : 0, 0x00020000);
    fa[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, 4 * (row * sg.lda + kk), 0, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[slot][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, 64 * (kk + j) + 4 * e, 0, 0));  // (tokens >= K: zeros)
    fk += 16;
    const bool next = fk >= K;
    fq = next ? fq + 1 : fq;
    fk = next ? 0 : fk;
  };
  auto multiply = [&](int slot) {
    const int l = lim[slot];
    if (l <= 0) return;
    f32x4 a = fa[slot];
    if (l < 16) {  // last step of a segment: the 16-byte load of W may run into the next segment's columns
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = 4 * fg + j < l ? a[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], fb[slot][j], acc, 0, 0, 0);
  };
  int steps = 0;  // (a dead segment costs one empty step)
  for (int q = 0; q < g.nseg; ++q) steps += (g.seg[q].A && g.seg[q].K > 0) ? (g.seg[q].K + 15) >> 4 : 1;
#pragma unroll
  for (int r = 0; r < WL_TOK_DEPTH; ++r) fetch(r);
  for (int t = 0; t < steps; t += WL_TOK_DEPTH) {
#pragma unroll
    for (int r = 0; r < WL_TOK_DEPTH; ++r) {
      multiply(r);
      fetch(r);
    }
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15 (e), row = 4 * (lane >> 4) + reg
  {
    const float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (mt * 16 + 4 * fg < M) epilogue_store_col4<NASREC_CM_TOKJ>(g, s0, mt * 16 + 4 * fg, b * 16 + e, M, v4);
  }
}

// Small dense Linear forward, the same scheme once more: y[i][n] = sum_k x[i][k] W[n][k] over up to eight 16-deep k-steps in all
// (binding KC / KC / plain: the 13-, 16-, 32- and 36-wide products of the dense nodes, the 16-wide FM projections; modules.py:171,340,
// 385,489,515,740).  On the general tile they are the slowest members of several forward levels (6 - 8.4 us as items): 64 x 16 strips
// staged through LDS behind a barrier, the bias read behind the MFMAs — three dependent trips to a cold L2.  Here a wavefront owns one
// 16 x 16 output tile: lane (r, g) loads x[16 mt + r][16 s + 4 g .. + 3] and W[16 nt + r][16 s + 4 g .. + 3] for EVERY step up front
// (one 16-byte load each; a null resource past the last segment), the bias line is touched at the same time, then the MFMAs run
// (MFMA j of step s sums k = 16 s + 4 g + j: the general tile's grouping, steps and segments in order — bit-identical to it).
#define WL_DENSE_STEPS 8
__device__ __forceinline__ void wl_dense_small(unsigned long long blob, int vb_, int NT_, int MTNB_) {
  const nasrec_gemm_desc_t& g = wl_ref<nasrec_gemm_desc_t>(blob);
  const int vb = __builtin_amdgcn_readfirstlane(vb_), NT = __builtin_amdgcn_readfirstlane(NT_);
  const int MT = __builtin_amdgcn_readfirstlane(MTNB_) & 0xffff, NB = __builtin_amdgcn_readfirstlane(MTNB_) >> 16;  // row tiles | batches of eight steps << 16
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = lane & 15, fg = lane >> 4;
  wl_warm_blob(blob, (int)offsetof(nasrec_gemm_desc_t, seg) + g.nseg * (int)sizeof(nasrec_gemm_seg_t));
  const nasrec_gemm_seg_t& s0 = g.seg[0];
  const int M = s0.M, N = s0.N;
  const int unit = vb * 4 + wave;  // (row tile, column tile)
  if (unit >= MT * NT) return;
  const int mt = unit / NT, nt = unit - mt * NT;
  const int row = min(mt * 16 + e, M - 1), col = min(nt * 16 + e, N - 1);  // (lane & 15: row of the A fragment, column of the B fragment)
  f32x4 fa[WL_DENSE_STEPS], fb[WL_DENSE_STEPS];
  int lim[WL_DENSE_STEPS];
  int fq = 0, fk = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int batch = 0; batch < NB; ++batch) {  // (one batch of eight steps for K <= 128; the 160-wide products take a second trip)
#pragma unroll
  for (int slot = 0; slot < WL_DENSE_STEPS; ++slot) {  // unconditional loads (see wl_token_fwd): the compiler can count them
    const bool in = fq < g.nseg;
    const nasrec_gemm_seg_t& sg = g.seg[in ? fq : 0];
    const bool live = in && sg.A != nullptr && sg.K > 0;
    const int K = live ? sg.K : 0, kk = fk + 4 * fg;
    lim[slot] = K - fk;
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A), 0, live ? (int)(4 * ((long)(M - 1) * sg.lda + K)) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.B), 0, live ? (int)(4 * ((long)(N - 1) * sg.ldb + K)) : 0, 0x00020000);
    fa[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, 4 * (row * sg.lda + kk), 0, 0));
    fb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, 4 * (col * sg.ldb + kk), 0, 0));
    fk += 16;
    const bool next = fk >= K;
    fq = next ? fq + 1 : fq;
    fk = next ? 0 : fk;
  }
  if (batch == 0 && g.bias) {  // pull the bias line in beside the operands (the epilogue reads it behind the MFMAs otherwise: one more cold trip)
    const float warm = g.bias[g.bias_on_rows ? row : col];
    asm volatile("" ::"v"(warm));
  }
#pragma unroll
  for (int slot = 0; slot < WL_DENSE_STEPS; ++slot) {
    const int l = lim[slot];
    if (l > 0) {  // (uniform)
      f32x4 a = fa[slot], b = fb[slot];
      if (l < 16) {  // last step of a segment: the 16-byte loads may run into the row's next columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = 4 * fg + j < l;
          a[j] = in ? a[j] : 0.f;
          b[j] = in ? b[j] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
  }
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  {
    const float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (mt * 16 + 4 * fg < M && nt * 16 + e < N) epilogue_store_col4<NASREC_CM_PLAIN>(g, s0, mt * 16 + 4 * fg, nt * 16 + e, M, v4);
  }
}

// ... and the small dense input gradients: dx_q[i][n] = sum_k dy_q[i][k] W_q[k][n] for every problem q of a zmode launch (binding KC / RC /
// plain; K = the Linear's output width <= 128).  A wavefront per 16 x 16 tile of one problem; lane (r, g) loads dy[16 mt + r][16 s + 4 g
// .. + 3] (one 16-byte load) and W[16 s + 4 g + j][16 nt + r], j = 0..3 (four dwords, a row of W apart), everything up front.
__device__ __forceinline__ void wl_dense_small_dx(unsigned long long blob, int vb_, int TU_) {
  const nasrec_gemm_desc_t& g = wl_ref<nasrec_gemm_desc_t>(blob);
  const int vb = __builtin_amdgcn_readfirstlane(vb_), TU = __builtin_amdgcn_readfirstlane(TU_);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = lane & 15, fg = lane >> 4;
  wl_warm_blob(blob, (int)offsetof(nasrec_gemm_desc_t, seg) + g.nseg * (int)sizeof(nasrec_gemm_seg_t));
  int t = vb * 4 + wave, q = 0;
  if (t >= TU) return;
  while (q < g.nseg - 1 && t >= ((g.seg[q].M + 15) >> 4) * ((g.seg[q].N + 15) >> 4)) t -= ((g.seg[q].M + 15) >> 4) * ((g.seg[q].N + 15) >> 4), ++q;
  const nasrec_gemm_seg_t& sg = g.seg[q];
  const int M = sg.M, N = sg.N, K = sg.K, NT = (N + 15) >> 4;
  const int mt = t / NT, nt = t - mt * NT;
  if (!sg.A || K <= 0) {  // a dead problem still owns its output: zero unless it accumulates (what the general tile does)
    if (!sg.accumulate) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = mt * 16 + 4 * fg + r, j = nt * 16 + e;
        if (i < M && j < N) epilogue_store<NASREC_CM_PLAIN>(g, sg, i, j, 0.f);
      }
    }
    return;
  }
  const int row = min(mt * 16 + e, M - 1), col = min(nt * 16 + e, N - 1);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A), 0, (int)(4 * ((long)(M - 1) * sg.lda + K)), 0x00020000);
  const float* wp = sg.B + col;  // B(n, k) = B[k ldb + n]
  f32x4 fa[WL_DENSE_STEPS];
  float fb[WL_DENSE_STEPS][4];
#pragma unroll
  for (int s = 0; s < WL_DENSE_STEPS; ++s) {
    fa[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, 4 * (row * sg.lda + 16 * s + 4 * fg), 0, 0));  // (past the extent: zeros)
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[s][j] = wp[(long)min(16 * s + 4 * fg + j, K - 1) * sg.ldb];  // (clamped, masked below)
  }
  if (sg.Aaux) {  // fused ReLU backward: A(r, k) counts as 0 where Aaux(r, k) <= 0 (same binding as A); the mask loads ride with the operands
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.Aaux), 0, (int)(4 * ((long)(M - 1) * sg.lda + K)), 0x00020000);
    f32x4 fx[WL_DENSE_STEPS];
#pragma unroll
    for (int s = 0; s < WL_DENSE_STEPS; ++s) fx[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, 4 * (row * sg.lda + 16 * s + 4 * fg), 0, 0));
#pragma unroll
    for (int s = 0; s < WL_DENSE_STEPS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[s][j] = fx[s][j] > 0.f ? fa[s][j] : 0.f;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < WL_DENSE_STEPS; ++s) {
    if (16 * s < K) {  // (uniform)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = 16 * s + 4 * fg + j < K;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? fa[s][j] : 0.f, in ? fb[s][j] : 0.f, acc, 0, 0, 0);
      }
    }
  }
  {
    const float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (mt * 16 + 4 * fg < M && nt * 16 + e < N) epilogue_store_col4<NASREC_CM_PLAIN>(g, sg, mt * 16 + 4 * fg, nt * 16 + e, M, v4);
  }
}

// Token-axis Linear input gradient, the same scheme: dx_q[b][r][e] = sum_i W[i][koff_q + r] dy[b][i][e] for every input segment q of
// the Linear (binding RC / TOKR / TOKJ, one independent problem per segment: zmode).  A wavefront per (sample, segment, 16 token
// rows); lane (r, g) of MFMA j of step s needs W[16 s + 4 g + j][r] (four dwords, a row of W apart), lane (e, g) needs
// dy[b][16 s + 4 g + j][e].  K = the Linear's output rows (16 .. 64): one to four steps, all loads issued up front.
#define WL_TOKDX_STEPS 4
// (A cap on the workgroups of this item with a wave walking several units — WL_TOK_MAX_WG = 256 — was measured: the three-segment product's
// 704 workgroups take 8.6 us capped or not, with or without its loads and stores (tools/step_table.py ITEMS=1), two-segment items got
// slower (4.9 -> 6.3 us), the step 0.2283 -> 0.2315 ms.  The loop stays, the cap does not bind.)
#define WL_TOK_MAX_WG (1 << 20)
__device__ __forceinline__ void wl_token_dx_unit(const nasrec_gemm_desc_t& g, int unit, int TU, int e, int fg);
__device__ __forceinline__ void wl_token_dx(unsigned long long blob, int vb_, int TU_) {
  const nasrec_gemm_desc_t& g = wl_ref<nasrec_gemm_desc_t>(blob);
  const int vb = __builtin_amdgcn_readfirstlane(vb_), TU = __builtin_amdgcn_readfirstlane(TU_);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  wl_warm_blob(blob, (int)offsetof(nasrec_gemm_desc_t, seg) + g.nseg * (int)sizeof(nasrec_gemm_seg_t));  // (the segment walk below reads every segment's fields)
  const int total = (g.seg[0].N >> 4) * TU, stride = 4 * min((total + 3) >> 2, WL_TOK_MAX_WG);
  for (int unit = vb * 4 + wave; unit < total; unit += stride) wl_token_dx_unit(g, unit, TU, lane & 15, lane >> 4);
}
__device__ __forceinline__ void wl_token_dx_unit(const nasrec_gemm_desc_t& g, int unit, int TU, int e, int fg) {
  const int b = unit / TU;
  int t = unit - b * TU, q = 0;
  while (q < g.nseg && t >= ((g.seg[q].M + 15) >> 4)) t -= (g.seg[q].M + 15) >> 4, ++q;  // (segment, row block) of this unit
  const nasrec_gemm_seg_t& sg = g.seg[q];
  const int M = sg.M, K = sg.K;
  if (!sg.A || K <= 0) {  // a dead problem still owns its output: zero unless it accumulates (what the general tile does)
    if (!sg.accumulate) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = t * 16 + 4 * fg + r;
        if (i < M) epilogue_store<NASREC_CM_TOKJ>(g, sg, i, b * 16 + e, 0.f);
      }
    }
    return;
  }
  const int row = min(t * 16 + e, M - 1);
  const float* wp = sg.A + row;                      // A(r, k) = A[k lda + r]
  const float* yb = sg.B + (long)b * sg.ldb + e;     // B(j, k) = B[b ldb + e + 16 k]
  float fa[WL_TOKDX_STEPS][4], fb[WL_TOKDX_STEPS][4];
#pragma unroll
  for (int s = 0; s < WL_TOKDX_STEPS; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = min(16 * s + 4 * fg + j, K - 1);  // (clamped, masked below)
      fa[s][j] = wp[(long)k * sg.lda];
      fb[s][j] = yb[k * 16];
    }
  if (sg.Baux) {  // fused ReLU backward: B(j, k) counts as 0 where Baux(j, k) <= 0 (same binding as B); the mask loads ride with the operands
    const float* xb = sg.Baux + (long)b * sg.ldb + e;
    float fx[WL_TOKDX_STEPS][4];
#pragma unroll
    for (int s = 0; s < WL_TOKDX_STEPS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) fx[s][j] = xb[min(16 * s + 4 * fg + j, K - 1) * 16];
#pragma unroll
    for (int s = 0; s < WL_TOKDX_STEPS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[s][j] = fx[s][j] > 0.f ? fb[s][j] : 0.f;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < WL_TOKDX_STEPS; ++s) {
    if (16 * s < K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = 16 * s + 4 * fg + j < K;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? fa[s][j] : 0.f, in ? fb[s][j] : 0.f, acc, 0, 0, 0);
      }
    }
  }
  {
    const float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (t * 16 + 4 * fg < M) epilogue_store_col4<NASREC_CM_TOKJ>(g, sg, t * 16 + 4 * fg, b * 16 + e, M, v4);
  }
}

__device__ __forceinline__ void wl_gemm_second_pass(unsigned long long blob, int vb_, int per_) {
  const nasrec_gemm_desc_t& g = wl_ref<nasrec_gemm_desc_t>(blob);
  const int vb = __builtin_amdgcn_readfirstlane(vb_), per = __builtin_amdgcn_readfirstlane(per_);
  const int nprob = g.zmode ? g.nseg : 1;
  int Mmax = 0, Nmax = 0;
  for (int q = 0; q < nprob; ++q) {
    Mmax = max(Mmax, g.seg[q].M);
    Nmax = max(Nmax, g.seg[q].N);
  }
  const int z = vb / per;
  const long e = (long)(vb - z * per) * 256 + threadIdx.x;
  if (g.cmode == NASREC_CM_PLAIN)
    splitk_second_pass<NASREC_CM_PLAIN>(g, Mmax, Nmax, z, e);
  else
    splitk_second_pass<NASREC_CM_TOKJ>(g, Mmax, Nmax, z, e);
}

__device__ __forceinline__ void wl_mha_fwd(unsigned long long blob, int vb) {
  mha_fwd_tok(wl_ref<nasrec_mha_desc_t>(blob), __builtin_amdgcn_readfirstlane(vb), wl_lds);
}

// BIG: the variant that also carries the Transformer backward (39.5 KB of LDS instead of 31; 127 registers and four workgroups per CU
// either way since round 6 — rounds 3 - 5: 164 registers, 52 KB, three per CU) — used for the levels that contain one
// The item table once more as twelve leading scalar kernel arguments: the build passes -amdgpu-kernarg-preload-count=12, so on gfx950
// they arrive in SGPRs with the wavefront (no memory access): f = first workgroup of item k (16 bits each, 0xffff behind the last
// item), m = kind | part << 6 | (blob offset / 16) << 8.  The search for the workgroup's item, its kind and the address of its
// descriptor then cost scalar ALU only; read from the argument segment — cold at every launch — the search was one dependent
// ~1 us miss per 64-byte line of the table before the descriptor's own (a one-item launch 5.45 us against 4.65 us for the
// stand-alone kernel).  WL_PACKED_NONE in f01: the table did not fit 16 bits per entry, use the copy in memory.
#define WL_PACKED_NONE 0xffffffffu
#define WL_HEAD_BYTES 48  // the twelve scalars in front of the descriptor in the argument segment
// `base`: address of the launch's nasrec_worklist_desc_t — in the kernel-argument segment (passed by value) or in device memory (ABI 17:
// uploaded once per plan; warm lines instead of a fresh copy per launch)
template <bool BIG>
__device__ __forceinline__ void wl_run(unsigned f01, unsigned f23, unsigned f45, unsigned f67, unsigned f89, unsigned fab, unsigned m01, unsigned m23,
                                       unsigned m45, unsigned m67, unsigned m89, unsigned mab, const unsigned long long base) {
  const nasrec_worklist_desc_t& wl = wl_ref<nasrec_worklist_desc_t>(base);
  float* lds = wl_lds;
  const int bid = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int k = 0, it_first = 0, it_kind, it_part, it_off;
  if (f01 != WL_PACKED_NONE) {
    const unsigned f[6] = {f01, f23, f45, f67, f89, fab}, m[6] = {m01, m23, m45, m67, m89, mab};
    unsigned meta = m01 & 0xffffu;
#pragma unroll
    for (int q = 1; q < NASREC_WL_MAX_ITEMS; ++q) {
      const unsigned fq = (f[q >> 1] >> (16 * (q & 1))) & 0xffffu, mq = (m[q >> 1] >> (16 * (q & 1))) & 0xffffu;
      if ((unsigned)bid >= fq) {  // (ascending; 0xffff behind the last item)
        k = q;
        it_first = (int)fq;
        meta = mq;
      }
    }
    it_kind = meta & 63;
    it_part = (meta >> 6) & 3;
    it_off = (int)(meta >> 8) * 16;
  } else {
#pragma unroll 1
    while (k + 1 < wl.n && bid >= wl.item[k + 1].first) ++k;
    it_first = wl.item[k].first;
    it_kind = wl.item[k].kind;
    it_part = wl.item[k].part;
    it_off = wl.item[k].off;
  }
  const nasrec_wl_item_t& it = wl.item[k];  // (geometry only below)
  const int vb = bid - it_first;
  const unsigned long long blob = base + offsetof(nasrec_worklist_desc_t, blob) + it_off;
  switch (it_kind) {
    case NASREC_OP_GEMM: {
      if (it_part == 2) {
        wl_gemm_second_pass(blob, vb, it.geom[0]);
        break;
      }
      const int cfg = it.geom[2];  // tile | binding pair << 2 | mask operand << 4
      if ((cfg & 3) == WL_TOKS && ((cfg >> 2) & 3) == 3 && it.geom[1] == 0) wl_dense_small_dx(blob, vb, it.geom[0]);
      else if ((cfg & 3) == WL_TOKS && ((cfg >> 2) & 3) == 3) wl_dense_small(blob, vb, it.geom[0], it.geom[1]);
      else if ((cfg & 3) == WL_TOKS && ((cfg >> 2) & 3) == 1) wl_token_fwd(blob, vb, it.geom[0]);
      else if ((cfg & 3) == WL_TOKS) wl_token_dx(blob, vb, it.geom[0]);
      else if ((cfg >> 4) & 1) wl_gemm_bind<true>((cfg >> 2) & 3, cfg & 3, blob, vb, it.geom[0], it.geom[1]);
      else wl_gemm_bind<false>((cfg >> 2) & 3, cfg & 3, blob, vb, it.geom[0], it.geom[1]);
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
      const int W = it.geom[0];
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
      final_bwd_block(wl_ref<nasrec_final_desc_t>(blob), it.geom[0], it.geom[1], it.geom[2], __builtin_amdgcn_readfirstlane(vb), lds);
      break;
    default:
      break;
  }
}

template <bool BIG>
__global__ __launch_bounds__(256, 4) void worklist_kernel(unsigned f01, unsigned f23, unsigned f45, unsigned f67, unsigned f89, unsigned fab,
                                                                    unsigned m01, unsigned m23, unsigned m45, unsigned m67, unsigned m89, unsigned mab,
                                                                    const nasrec_worklist_desc_t wl) {
  wl_run<BIG>(f01, f23, f45, f67, f89, fab, m01, m23, m45, m67, m89, mab, (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + WL_HEAD_BYTES);
}
template <bool BIG>
__global__ __launch_bounds__(256, 4) void worklist_dev_kernel(unsigned f01, unsigned f23, unsigned f45, unsigned f67, unsigned f89, unsigned fab,
                                                                        unsigned m01, unsigned m23, unsigned m45, unsigned m67, unsigned m89, unsigned mab,
                                                                        const nasrec_worklist_desc_t* wl) {
  wl_run<BIG>(f01, f23, f45, f67, f89, fab, m01, m23, m45, m67, m89, mab, (unsigned long long)wl);
}
