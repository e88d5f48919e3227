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

// one wavefront = one sample b; ts = TRI_MAXK1 * TRI_LD floats of LDS owned by that wavefront
__device__ __forceinline__ void dot_tri_fwd_sample(const nasrec_dot_tri_desc_t& d, int b, int lane, float* ts) {
  const int k1 = d.k1;
  const float* Tb = d.T + (long)b * k1 * 16;
  for (int q = lane; q < k1 * 16; q += 64) ts[(q >> 4) * TRI_LD + (q & 15)] = Tb[q];
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
  for (int n = g; n < d.N; n += 4) {
    float v = x[n * 16];
    s += v;
    q = fmaf(v, v, q);
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

