// Device bodies of the optimizer tail (clip_grad_norm_ + Adagrad, train_utils.py:285-286 / main_train.py:152-154), shared
// by the stand-alone kernels and by the two fused launches NASREC_OP_OPT_REDUCE / NASREC_OP_OPT_APPLY.
#pragma once
#include "common.h"

// sum of squares of x[0, n) (or of the chunk table's ranges): workgroup `blk` of `nblk` (256 threads), fixed-order tree -> partial[blk]
__device__ __forceinline__ void sumsq_body(const nasrec_sumsq_desc_t& d, int blk, int nblk, float* red) {
  const int tid = threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains keep several cold loads in flight
  if (d.chunks) {
    for (long c = blk; c < d.nchunks; c += nblk) {
      const float* x = d.x + d.chunks[2 * c];
      const long n = d.chunks[2 * c + 1], n4 = n >> 2;
      long i = tid;
      for (; i + 768 < n4; i += 1024) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * i), b = *reinterpret_cast<const f32x4*>(x + 4 * (i + 256));
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(x + 4 * (i + 512)), e = *reinterpret_cast<const f32x4*>(x + 4 * (i + 768));
        s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
        s1 += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
        s2 += (c2[0] * c2[0] + c2[1] * c2[1]) + (c2[2] * c2[2] + c2[3] * c2[3]);
        s3 += (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
      }
      for (; i < n4; i += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * i);
        s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
      }
      for (long j = 4 * n4 + tid; j < n; j += 256) s1 = fmaf(x[j], x[j], s1);
    }
  } else {
    // 16-byte loads, four in flight per thread and trip (the arena is 16-byte aligned): with one float per load a thread of the
    // batch-256 step (2.2 M gradients over 256 workgroups) made nine dependent trips to memory, now two
    const long stride = (long)nblk * 256, n4 = d.n >> 2;
    long i = (long)blk * 256 + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d.x + 4 * i), b = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + stride));
      const f32x4 c = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + 2 * stride)), e = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + 3 * stride));
      s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
      s1 += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
      s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
      s3 += (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
    }
    {  // up to three more pieces per thread, loaded together
      f32x4 r[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) r[u] = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + u * stride < n4 ? i + u * stride : 0));
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (i + u * stride < n4) s0 += (r[u][0] * r[u][0] + r[u][1] * r[u][1]) + (r[u][2] * r[u][2] + r[u][3] * r[u][3]);
    }
    for (long j = 4 * n4 + (long)blk * 256 + tid; j < d.n; j += stride) s1 = fmaf(d.x[j], d.x[j], s1);
  }
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) d.partial[blk] = red[0];
}

// one wavefront: lanes take the partials round-robin (fixed assignment), fp64 butterfly -> the same value in every lane
__device__ __forceinline__ float clip_coef_wave(const nasrec_clip_coef_desc_t& d, int lane, float* total_out) {
  // eight partials per lane and trip in flight, the first trip of list b beside list a's (a plain loop is load -> wait -> add per
  // partial: 256 + 26 partials were five dependent round trips at the head of EVERY workgroup of the apply launch).  The additions
  // per lane are in the same order as before — all of a, then all of b: same bits.
  double s = 0.0;
  const int na = d.n_a, nb = d.n_b;
  if (na > 0 || nb > 0) {
    const float* pa = na > 0 ? d.partial_a : d.partial_b;  // (an empty list is read at the other list's first element and dropped)
    const float* pb = nb > 0 ? d.partial_b : d.partial_a;
    const int ca = max(na - 1, 0), cb = max(nb - 1, 0);
    float vb0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) vb0[u] = pb[min(lane + 64 * u, cb)];
    for (int i0 = lane; i0 < na; i0 += 64 * 8) {
      float va[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) va[u] = pa[min(i0 + 64 * u, ca)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 64 * u < na) s += (double)va[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (lane + 64 * u < nb) s += (double)vb0[u];
    for (int i0 = lane + 64 * 8; i0 < nb; i0 += 64 * 8) {
      float vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vb[u] = pb[min(i0 + 64 * u, cb)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 64 * u < nb) s += (double)vb[u];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float total = (float)sqrt(s);
  float coef = 1.f;
  if (d.max_norm > 0.f) coef = fminf(d.max_norm / (total + 1e-6f), 1.f);
  *total_out = total;
  return coef;
}

__device__ __forceinline__ void adagrad_dense_body(const nasrec_adagrad_dense_desc_t& d, int blk, int nblk, float lr, float coef) {
  if (d.chunks) {
    for (long c = blk; c < d.nchunks; c += nblk) {
      const long off = d.chunks[2 * c], n = d.chunks[2 * c + 1], n4 = n >> 2;
      const float* gp = d.g + off;
      float* sp = d.state + off;
      float* pp = d.p + off;
      for (long i = threadIdx.x; i < n4; i += 256) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp + 4 * i);
        f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + 4 * i), p4 = *reinterpret_cast<const f32x4*>(pp + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = g4[e] * coef;
          s4[e] = fmaf(g, g, s4[e]);
          p4[e] = p4[e] - lr * (g / (sqrtf(s4[e]) + d.eps));
        }
        *reinterpret_cast<f32x4*>(sp + 4 * i) = s4;
        *reinterpret_cast<f32x4*>(pp + 4 * i) = p4;
      }
      for (long j = 4 * n4 + threadIdx.x; j < n; j += 256) {
        const float g = gp[j] * coef;
        const float s = fmaf(g, g, sp[j]);
        sp[j] = s;
        pp[j] = pp[j] - lr * (g / (sqrtf(s) + d.eps));
      }
    }
    return;
  }
  for (long i = (long)blk * 256 + threadIdx.x; i < d.n; i += (long)nblk * 256) {
    const float g = d.g[i] * coef;
    const float s = fmaf(g, g, d.state[i]);
    d.state[i] = s;
    d.p[i] = d.p[i] - lr * (g / (sqrtf(s) + d.eps));
  }
}

// Row-sparse clip + Adagrad on the touched rows only (4 lanes x float4 per row); workgroup `blk` of 256 threads
__device__ __forceinline__ void adagrad_rows_body(const nasrec_adagrad_rows_desc_t& d, int blk, float lr, float coef) {
  const long t = (long)blk * 256 + threadIdx.x;
  const long pair = t >> 2;
  const int q = (int)(t & 3);
  if (pair >= (long)d.B * d.Fs) return;
  // Two memory round trips instead of five: everything that does not depend on the row id — leader flag, id, the field's row count and
  // its two base pointers (per-lane loads from the argument arrays: f differs inside a wavefront), the gradient piece — is issued
  // together; written as `if (!leader) return; row = idx; if (row >= rows[f]) return; ...` every test was a load, a wait and a branch.
  const int f = (int)(pair % d.Fs);
  const int lead = d.leader[pair];
  const long row = d.idx[pair];
  const long nrows = d.rows[f];
  float* const sbase = d.state[f];
  float* const tbase = d.table[f];
  const float* grow = d.gsum + pair * 16;
  if (d.rank_B > 0) {  // gsum = the receive buffer of an all-gather whose per-rank chunks are rank_stride floats apart
    const long b = pair / d.Fs, r = b / d.rank_B;
    grow = d.gsum + r * d.rank_stride + ((b - r * d.rank_B) * d.Fs + f) * 16;
  }
  float4 g = *reinterpret_cast<const float4*>(grow + q * 4);
  asm volatile("" ::"v"(lead), "v"((int)row), "v"((int)nrows), "v"(sbase), "v"(tbase), "v"(g.x));  // (all six in flight before the first test)
  if (!lead) return;
  if (row < 0 || row >= nrows) return;  // never write outside a table
  float4* sp = reinterpret_cast<float4*>(sbase + row * 16 + q * 4);
  float4* pp = reinterpret_cast<float4*>(tbase + row * 16 + q * 4);
  float4 s = *sp, p = *pp;
  float gg[4] = {g.x * coef, g.y * coef, g.z * coef, g.w * coef};
  float ssv[4] = {s.x, s.y, s.z, s.w};
  float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ssv[e] = fmaf(gg[e], gg[e], ssv[e]);
    pv[e] = pv[e] - lr * (gg[e] / (sqrtf(ssv[e]) + d.eps));
  }
  *sp = make_float4(ssv[0], ssv[1], ssv[2], ssv[3]);
  *pp = make_float4(pv[0], pv[1], pv[2], pv[3]);
}

static inline int adagrad_dense_blocks(long n) {
  long blocks = (n + 255) / 256;
  return (int)(blocks > 2048 ? 2048 : blocks);
}
