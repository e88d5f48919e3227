// Per-sample bodies of the interaction / glue kernels (see interact.hip), shared with chain.hip.
#pragma once
#include "common.h"

#define TRI_MAXK1 64
#define TRI_LD 20

__device__ __forceinline__ void tri_decode(int p, int& i, int& j) {
  // p = i(i-1)/2 + j, 0 <= j < i  (row-major strictly-lower triangle == torch.tril_indices(offset=-1))
  i = (int)((1.f + sqrtf(1.f + 8.f * (float)p)) * 0.5f);
  while (i * (i - 1) / 2 > p) --i;
  while ((i + 1) * i / 2 <= p) ++i;
  j = p - i * (i - 1) / 2;
}


// One wavefront copies n floats global -> LDS (dst index by a functor), eight loads in flight per lane: a plain loop compiles to
// load -> wait -> store per trip, i.e. n / 64 dependent memory round trips (12 - 16 of them for a 46-feature DotProduct sample:
// most of that kernel's time on the cold L2 of a batch-256 step).  Indices past n are clamped for the load and skipped for the store.
template <typename Dst>
__device__ __forceinline__ void wave_copy_in(const float* src, int n, int lane, Dst dst) {
  for (int q0 = 0; q0 < n; q0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[min(q0 + 64 * u + lane, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + 64 * u + lane;
      if (q < n) dst(q, v[u]);
    }
  }
}

// The T rows of one sample (k1 x 16 floats, 64-byte aligned) global -> registers by 16-byte loads, every load of the wavefront in
// flight at once (k1 <= 64: at most four per lane), and registers -> LDS rows of TRI_LD floats.
struct TriRows {
  f32x4 v[TRI_MAXK1 * 4 / 64];
};
__device__ __forceinline__ void tri_rows_load(const float* Tb, int k1, int lane, TriRows& r) {
#pragma unroll
  for (int u = 0; u < TRI_MAXK1 * 4 / 64; ++u)
    if (64 * u < k1 * 4) r.v[u] = *reinterpret_cast<const f32x4*>(Tb + 4 * min(64 * u + lane, k1 * 4 - 1));  // (uniform guard)
}
__device__ __forceinline__ void tri_rows_store(float* ts, int k1, int lane, const TriRows& r) {
#pragma unroll
  for (int u = 0; u < TRI_MAXK1 * 4 / 64; ++u) {
    const int q = 64 * u + lane;  // 16-byte piece q = (row q >> 2, columns 4 (q & 3) ..)
    if (q < k1 * 4) *reinterpret_cast<f32x4*>(ts + (q >> 2) * TRI_LD + 4 * (q & 3)) = r.v[u];
  }
}

#ifndef TRI_MFMA
#define TRI_MFMA 1  // 0: the LDS / vector-ALU bodies of rounds 1-3 (A/B builds)
#endif
#ifndef TRI_MFMA_FWD
// The FORWARD keeps the sequential 16-FMA chain on LDS-resident rows by default: it feeds the logits, and on the golden network with the
// largest logits (fixed_criteo_autoctr, |logit| up to 39.5) the chain is 2.63e-5 from the reference's fp64 evaluation where the matrix
// pipe's four-term sums land at 3.39e-5 (the reference's own fp32 run: 2.75e-5) — for 0.3 % of a cfg-5 step and
// nothing measurable at batch 256 (A/B -DTRI_MFMA_FWD=1).  The backward (gradients only) runs on the matrix pipe.
#define TRI_MFMA_FWD 0
#endif

// The DotProduct core on the matrix pipe (round 4).  out[p(i, j)] = <T[i], T[j]>, j < i, is the strict lower triangle of the Gram matrix
// G = T T^T ([k1, 16] x [16, k1]): per 16 x 16 block of G four v_mfma_f32_16x16x4_f32, and BOTH operands are the sample's memory as it
// lies — lane (r, g) of MFMA j supplies A[i = r][k = g] = T[16 bi + r][4 g + j] and B[k = g][n = r] = T[16 bj + r][4 g + j], i.e. ONE
// 16-byte load per lane and 16-row block (the 64 lanes read a contiguous 1 KB), no LDS, no barrier.  D: row 4 g + q, column r —
// sixteen consecutive outputs of a row per store.  k1 = 46: 3 loads, 24 MFMAs, 24 stores per lane against 17 pairs x 8 ds_read_b128 +
// 16 FMAs.  (Each output is the same 16 products; the matrix pipe adds them in its own order — k = j, 4 + j, 8 + j, 12 + j inside MFMA j.)
__device__ __forceinline__ void dot_tri_fwd_sample_mfma(const nasrec_dot_tri_desc_t& d, int b, int lane) {
  const int k1 = d.k1;
  const int r = lane & 15, g = lane >> 4;
  const float* Tb = d.T + (long)b * k1 * 16;
  float* ob = d.out + (long)b * d.ld_out;
  f32x4 t[TRI_MAXK1 / 16];
#pragma unroll
  for (int bi = 0; bi < TRI_MAXK1 / 16; ++bi)
    if (16 * bi < k1) t[bi] = *reinterpret_cast<const f32x4*>(Tb + min(16 * bi + r, k1 - 1) * 16 + 4 * g);  // (uniform guard; rows >= k1 clamped, never stored)
#pragma unroll
  for (int bi = 0; bi < TRI_MAXK1 / 16; ++bi) {
    if (16 * bi >= k1) break;
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t[bi][j], t[bj][j], acc, 0, 0, 0);
      const int jc = 16 * bj + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * bi + 4 * g + q;
        if (i < k1 && jc < i) ob[i * (i - 1) / 2 + jc] = acc[q];
      }
    }
  }
}

// ... and its backward: dT = S T with S the symmetric [k1, k1] matrix of the output gradients (S[i][k] = dO[p(max, min)], zero diagonal).
// Lane (r, g) of step s supplies A[i = r][k = g] = S[16 bi + r][4 s + g] — a gather from the sample's 4 KB of dO — and
// B[k = g][n = r] = T[4 s + g][r] — 64 consecutive floats per step.  Every load of the sample is issued before the first MFMA (one round
// trip); k1 = 46: 12 + 36 dwords per lane, 36 MFMAs, 12 stores, no LDS.
template <int NB>  // 16-row blocks: k1 <= 16 NB (static trip counts: every load below is unconditional — a guard per load is a branch, a wait and a select per load)
__device__ __forceinline__ void dot_tri_bwd_sample_mfma_t(const nasrec_dot_tri_desc_t& d, int b, int lane) {
  constexpr int KS = 4 * NB;
  const int k1 = d.k1;
  const int P = k1 * (k1 - 1) / 2;
  const int r = lane & 15, g = lane >> 4;
  const float* Tb = d.T + (long)b * k1 * 16;
  const float* dob = d.dout + (long)b * d.ld_out;
  float tb[KS], a[NB][KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) tb[s] = Tb[min(4 * s + g, k1 - 1) * 16 + r];  // (a row >= k1 is clamped and meets a zero of S)
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    const int i = 16 * bi + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + g;
      const int hi = max(i, k), lo = min(i, k);
      a[bi][s] = dob[min(hi * (hi - 1) / 2 + lo, max(P - 1, 0))];  // (clamped; masked below)
    }
  }
  // every load of the sample has been issued: pin the values here, or the compiler sinks a block's gathers behind the previous block's
  // stores (and then waits for those stores' acknowledgements, block after block)
#pragma unroll
  for (int s = 0; s < KS; s += 4) asm volatile("" ::"v"(tb[s]), "v"(tb[s + 1]), "v"(tb[s + 2]), "v"(tb[s + 3]));
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
#pragma unroll
    for (int s = 0; s < KS; s += 4) asm volatile("" ::"v"(a[bi][s]), "v"(a[bi][s + 1]), "v"(a[bi][s + 2]), "v"(a[bi][s + 3]));
  float* dTb = d.dT + (long)b * k1 * 16;
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    const int i = 16 * bi + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + g;
      const float av = (i != k && i < k1 && k < k1) ? a[bi][s] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tb[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int io = 16 * bi + 4 * g + q;
      if (io < k1) dTb[io * 16 + r] = acc[q];
    }
  }
}
__device__ __forceinline__ void dot_tri_bwd_sample_mfma(const nasrec_dot_tri_desc_t& d, int b, int lane) {
  const int nb = (d.k1 + 15) >> 4;  // (uniform)
  if (nb <= 1) dot_tri_bwd_sample_mfma_t<1>(d, b, lane);
  else if (nb == 2) dot_tri_bwd_sample_mfma_t<2>(d, b, lane);
  else if (nb == 3) dot_tri_bwd_sample_mfma_t<3>(d, b, lane);
  else dot_tri_bwd_sample_mfma_t<4>(d, b, lane);
}

// one wavefront = one sample b; ts = TRI_MAXK1 * TRI_LD floats of LDS owned by that wavefront
__device__ __forceinline__ void dot_tri_fwd_sample(const nasrec_dot_tri_desc_t& d, int b, int lane, float* ts) {
  if (TRI_MFMA_FWD) {
    dot_tri_fwd_sample_mfma(d, b, lane);
    return;
  }
  const int k1 = d.k1;
  const float* Tb = d.T + (long)b * k1 * 16;
  TriRows tr;
  tri_rows_load(Tb, k1, lane, tr);
  tri_rows_store(ts, k1, lane, tr);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  const int P = k1 * (k1 - 1) / 2;
  float* ob = d.out + (long)b * d.ld_out;
  for (int p = lane; p < P; p += 64) {
    int i, j;
    tri_decode(p, i, j);
    const f32x4* ri = reinterpret_cast<const f32x4*>(ts + i * TRI_LD);
    const f32x4* rj = reinterpret_cast<const f32x4*>(ts + j * TRI_LD);
    float acc = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f32x4 a = ri[v], c = rj[v];
      acc = fmaf(a[0], c[0], acc);
      acc = fmaf(a[1], c[1], acc);
      acc = fmaf(a[2], c[2], acc);
      acc = fmaf(a[3], c[3], acc);
    }
    ob[p] = acc;
  }
}

__device__ __forceinline__ void fm_fwd_sample(const nasrec_fm_desc_t& d, int b, int lane) {
  const int g = lane >> 4, e = lane & 15;
  const float* x = d.x + (long)b * d.ldx + e;
  float s = 0.f, q = 0.f;
  for (int n0 = g; n0 < d.N; n0 += 4 * 8) {  // eight loads in flight (same summation order as the plain loop)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[min(n0 + 4 * u, d.N - 1) * 16];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (n0 + 4 * u < d.N) {
        s += v[u];
        q = fmaf(v[u], v[u], q);
      }
  }
  s += __shfl_xor(s, 16, 64);
  q += __shfl_xor(q, 16, 64);
  s += __shfl_xor(s, 32, 64);
  q += __shfl_xor(q, 32, 64);
  if (g == 0) {
    float r = s * s - q;
    float* o = d.ix + (long)b * d.ld_ix + e;
    if (d.add) r += d.add[(long)b * d.ld_ix + e];
    *o = d.accumulate ? *o + r : r;
  }
}


// element (b, j) of the segmented view
__device__ __forceinline__ void copy_segs_element(const nasrec_copy_segs_desc_t& d, int b, int j) {
  int q = 0;
  for (; q < d.nseg; ++q)
    if (j >= d.off[q] && j < d.off[q] + d.width[q]) break;
  if (q == d.nseg) return;
  const int jj = j - d.off[q];
  float* dp = d.dst + (long)b * d.ld_dst + j;
  if (!d.reverse) {
    float v = d.seg[q] ? d.seg[q][(long)b * d.ld[q] + jj] : 0.f;
    *dp = d.accumulate ? *dp + v : v;
  } else if (d.seg[q]) {
    float* sp = d.seg[q] + (long)b * d.ld[q] + jj;
    *sp = d.seg_accumulate[q] ? *sp + *dp : *dp;
  }
}


// dT[b,i,:] = sum_{j<i} dO[p(i,j)] T[j] + sum_{j>i} dO[p(j,i)] T[j]; one wavefront = one sample; ts / ds = this wavefront's LDS
// (k1 * TRI_LD and k1 (k1 - 1) / 2 floats)
__device__ __forceinline__ void dot_tri_bwd_sample(const nasrec_dot_tri_desc_t& d, int b, int lane, float* ts, float* ds) {
  if (TRI_MFMA) {
    dot_tri_bwd_sample_mfma(d, b, lane);
    return;
  }
  const int k1 = d.k1;
  const int P = k1 * (k1 - 1) / 2;
  const float* Tb = d.T + (long)b * k1 * 16;
  const float* dob = d.dout + (long)b * d.ld_out;
  // every global load of the sample before the first LDS store: the T rows (<= 4 16-byte loads per lane) and the P <= 2016 output
  // gradients (<= 32 dwords per lane; their row start is not 16-byte aligned).  In batches of 8 per lane these were five dependent
  // round trips on the cold L2 of a batch-256 step — half of the item's 10 us.
  TriRows tr;
  tri_rows_load(Tb, k1, lane, tr);
  constexpr int DMAX = (TRI_MAXK1 * (TRI_MAXK1 - 1) / 2 + 63) / 64;
  float dv[DMAX];
#pragma unroll
  for (int u = 0; u < DMAX; ++u)
    if (64 * u < P) dv[u] = dob[min(64 * u + lane, P - 1)];  // (uniform guard)
  tri_rows_store(ts, k1, lane, tr);
#pragma unroll
  for (int u = 0; u < DMAX; ++u)
    if (64 * u + lane < P) ds[64 * u + lane] = dv[u];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  float* dTb = d.dT + (long)b * k1 * 16;
  // item = (row i, 4 columns q); a round of 64 items covers rows imin .. imax.  Both loops run over a UNIFORM range of j with every
  // LDS read unconditional and the lane's own bound (j < i, j > i) applied to the weight: with the lane's bound as the loop bound
  // the trip count diverged over the wavefront, nothing was unrolled and every trip exposed its LDS round trip (~110 clocks per j,
  // 7 of the item's 10 us at k1 = 46).  A lane adds the same terms in the same order as before (the others are + 0 * T[j]).
  for (int item0 = 0; item0 < k1 * 4; item0 += 64) {
    const int item = item0 + lane;
    const int i = min(item >> 2, k1 - 1), q = item & 3;
    const int imin = item0 >> 2, imax = min((item0 + 63) >> 2, k1 - 1);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wrow = ds + i * (i - 1) / 2;
    const float* tq = ts + 4 * q;
#pragma unroll 8
    for (int j = 0; j < imax; ++j) {  // rows below the lane's: dO[p(i, j)], j < i
      const float w = wrow[j];
      const f32x4 t = *reinterpret_cast<const f32x4*>(tq + j * TRI_LD);
      acc += (j < i ? w : 0.f) * t;
    }
    const float* wcol = ds + i;
    int tj = (imin + 1) * imin / 2;  // p(j, 0) of the first j
#pragma unroll 8
    for (int j = imin + 1; j < k1; ++j) {  // rows above: dO[p(j, i)], j > i
      const float w = wcol[tj];
      tj += j;
      const f32x4 t = *reinterpret_cast<const f32x4*>(tq + j * TRI_LD);
      acc += (j > i ? w : 0.f) * t;
    }
    if (item < k1 * 4) *reinterpret_cast<f32x4*>(dTb + i * 16 + 4 * q) = acc;
  }
}

__device__ __forceinline__ void fm_bwd_sample(const nasrec_fm_desc_t& d, int b, int lane) {
  const int g = lane >> 4, e = lane & 15;
  const float* x = d.x + (long)b * d.ldx + e;
  float* dx = d.dx + (long)b * d.ldx + e;
  float s = 0.f;
  const float g2 = 2.f * d.dix[(long)b * d.ld_ix + e];  // (issued with the first batch)
  for (int n0 = g; n0 < d.N; n0 += 4 * 8) {  // eight loads in flight (same summation order as the plain loop)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[min(n0 + 4 * u, d.N - 1) * 16];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (n0 + 4 * u < d.N) s += v[u];
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  for (int n0 = g; n0 < d.N; n0 += 4 * 8) {
    float v[8], o[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = min(n0 + 4 * u, d.N - 1);
      v[u] = x[n * 16];
      o[u] = d.accumulate ? dx[n * 16] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (n0 + 4 * u < d.N) dx[(n0 + 4 * u) * 16] = o[u] + g2 * (s - v[u]);
  }
}

// SigmoidGating backward, element t = (b, j) (modules.py:578-582)
__device__ __forceinline__ void gate_bwd_element(const nasrec_gate_bwd_desc_t& d, long t) {
  if (t >= (long)d.B * d.D) return;
  const int b = (int)(t / d.D), j = (int)(t % d.D);
  const float go = d.dout[(long)b * d.ld_dout + j];
  const float g = d.g[(long)b * d.ld_g + j];
  float r = 0.f;
  for (int q = 0; q < d.nseg; ++q) {
    const int jj = j - d.r_off[q];
    if (jj >= 0 && jj < d.r_width[q]) {
      if (d.r_ptr[q]) r = d.r_ptr[q][(long)b * d.r_ld[q] + jj];
      if (d.dr_ptr[q]) {
        float* p = d.dr_ptr[q] + (long)b * d.r_ld[q] + jj;
        const float v = go * g;
        *p = d.dr_accumulate[q] ? *p + v : v;
      }
      break;
    }
  }
  d.dz[(long)b * d.ld_dz + j] = go * r * g * (1.f - g);
}

// out[c] = sum_r in[r*ld + c] in fixed order for the 16 columns of workgroup vb, scattered to the destinations by column range.
// D: nasrec_reduce_rows_desc_t or the compact form a worklist launch carries (same field names).  red = 16 x 17 floats of LDS.
template <typename D>
__device__ __forceinline__ void reduce_rows_block(const D& d, int vb, int tid, float* red) {
  const int cl = tid & 15, rq = tid >> 4;
  const int c = vb * 16 + cl;
  float s = 0.f;
  if (c < d.C) {
    // four independent chains, sixteen loads in flight per trip (256 rows: one trip to memory instead of four)
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rq;
    for (; r + 240 < d.R; r += 256) {
      float a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = d.in[(long)(r + 16 * u) * d.ld + c];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        s += a[u];
        s1 += a[u + 1];
        s2 += a[u + 2];
        s3 += a[u + 3];
      }
    }
    for (; r + 48 < d.R; r += 64) {
      const float a0 = d.in[(long)r * d.ld + c], a1 = d.in[(long)(r + 16) * d.ld + c];
      const float a2 = d.in[(long)(r + 32) * d.ld + c], a3 = d.in[(long)(r + 48) * d.ld + c];
      s += a0;
      s1 += a1;
      s2 += a2;
      s3 += a3;
    }
    for (; r < d.R; r += 16) s += d.in[(long)r * d.ld + c];
    s = (s + s1) + (s2 + s3);
  }
  red[rq * 17 + cl] = s;
  __syncthreads();
  if (rq == 0 && c < d.C) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q * 17 + cl];
    for (int q = 0; q < d.ndst; ++q) {
      const int cc = c - d.dst_off[q];
      if (cc >= 0 && cc < d.dst_len[q]) {
        if (d.dst[q]) d.dst[q][cc] = tot;
        break;
      }
    }
  }
}
