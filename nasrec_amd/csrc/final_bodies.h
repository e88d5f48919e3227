// Device bodies of the final-logit kernels (supernet.py:592-598 / 657-664 and the fused BCEWithLogits, main_train.py:122), shared by
// the stand-alone kernels (norm_loss_opt.hip) and by the worklist launches (worklist_body.h: the final-logit backward sits in the
// same level as other operators of the step; as an item it runs beside them instead of before them).
#pragma once
#include "common.h"

// weight column of feature j of segment q, and its inverse (feature of weight column k, or -1): contiguous, or token-strided for the
// interleaved sparse outputs of last_n_blocks_out > 1 (nasrec_final_desc_t.tok_stride)
__device__ __forceinline__ int final_wcol(const nasrec_final_desc_t& d, int q, int j) {
  const int ts = d.tok_stride[q];
  return d.off[q] + (ts ? (j >> 4) * ts + (j & 15) : j);
}
__device__ __forceinline__ int final_feat(const nasrec_final_desc_t& d, int q, int k) {
  const int r = k - d.off[q], ts = d.tok_stride[q];
  if (r < 0) return -1;
  if (!ts) return r < d.width[q] ? r : -1;
  const int t = r / ts, e = r - t * ts;
  const int j = t * 16 + e;
  return (e < 16 && j < d.width[q]) ? j : -1;
}
__host__ __device__ inline int final_seg_extent(const nasrec_final_desc_t& d, int q) {
  const int ts = d.tok_stride[q], W = d.width[q];
  return (ts && W > 0) ? ((W - 1) >> 4) * ts + ((W - 1) & 15) + 1 : W;
}

// ---------------------------------------------------------------------------------------------------
// final logit: one WORKGROUP per sample (round 5), its four wavefronts take every fourth 64-feature trip of the (segmented) feature axis
// — four trips each in flight — and meet through four floats of LDS: z = ((s0 + s1) + (s2 + s3)) + bias.  (One wavefront per sample
// walked the ~1.6 k features of the bench network in 25 trips of 64, four in flight: 3.5 us for the forward at batch 256, 5.7 us
// with the per-sample backward behind it — on the step's critical path, alone in its level.)
// ---------------------------------------------------------------------------------------------------
// -> the sample's logit in every thread; red: 4 floats of LDS
__device__ __forceinline__ float final_logit(const nasrec_final_desc_t& d, int b, float* red) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  for (int q = 0; q < d.nseg; ++q) {
    if (!d.seg[q]) continue;
    const float* x = d.seg[q] + (long)b * d.ld[q];
    const float* w = d.w + d.off[q];
    const int W = d.width[q], ts = d.tok_stride[q];
    for (int j0 = lane + 64 * wave; j0 < W; j0 += 256 * 4) {
      float xv[4], wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + 256 * u, W - 1);
        xv[u] = x[j];
        wv[u] = w[ts ? (j >> 4) * ts + (j & 15) : j];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + 256 * u < W) s = fmaf(xv[u], wv[u], s);
    }
  }
  s = wave_sum(s);
  __syncthreads();  // (red may still be read from an earlier use of the buffer)
  if (lane == 0) red[wave] = s;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + d.bias[0];
}

// workgroup vb of B
__device__ __forceinline__ void final_fwd_block(const nasrec_final_desc_t& d, int vb, float* red) {
  if (vb >= d.B) return;
  const float z = final_logit(d, vb, red);
  if (threadIdx.x == 0) d.logits[vb] = z;
}

__device__ __forceinline__ float bce_grad(float z, float y, float scale) { return (1.f / (1.f + expf(-z)) - y) * scale; }
__device__ __forceinline__ float bce_term(float z, float y) { return fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z))); }

// NASREC_OP_FINAL_FUSED: final_fwd_block, and behind it — in the workgroup that holds the sample's logit — the part of the backward that
// needs nothing else: dseg[b, j] (+)= dlogits[b] * w[col(j)], the expression of final_bwd_block's part A.  workgroup vb of B.
__device__ __forceinline__ void final_fused_block(const nasrec_final_desc_t& d, int vb, float* red) {
  const int b = vb;
  if (b >= d.B) return;
  const float yb = d.y[b];  // (in flight under the sum)
  const float z = final_logit(d, b, red);
  if (threadIdx.x == 0) d.logits[b] = z;
  const float g = bce_grad(z, yb, d.grad_scale);
  const int tid = threadIdx.x;
  for (int q = 0; q < d.nseg; ++q) {
    float* p = d.dseg[q];
    if (!p || !d.seg[q]) continue;
    p += (long)b * d.ld[q];
    const float* w = d.w + d.off[q];
    const int W = d.width[q], ts = d.tok_stride[q];
    const bool acc = d.dseg_accumulate[q] != 0;
    for (int j0 = tid; j0 < W; j0 += 256 * 4) {  // reads first (weights; the accumulation targets), then the stores
      float wv[4], cv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + 256 * u, W - 1);
        wv[u] = w[ts ? (j >> 4) * ts + (j & 15) : j];
        cv[u] = acc ? p[j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = g * wv[u];
        if (j0 + 256 * u < W) p[j0 + 256 * u] = acc ? cv[u] + v : v;
      }
    }
  }
}

// backward: part A (blocks [0, nA)): dseg[b,j] (+)= dlogits[b] * w[off+j]
//           part B (blocks [nA, nA+nB)): dw[k] = sum_b dlogits[b] * feat[b,k]; dbias = sum_b dlogits[b]
//           part C (block nA+nB, only with the fused BCE): loss and the per-sample gradient
// With d.y != NULL, dlogits[b] is derived on the fly from logits[b] and y[b] (BCEWithLogits fused into this launch).
// workgroup vb of nA + nB (+ 1 with the fused loss); lds: 272 floats
#define FINAL_BWD_LDS_FLOATS 272
#define FINAL_BWD_ROWS 16  // samples per part-A workgroup at large batch
__host__ __device__ inline int final_bwd_rows(int B) { return B >= 1024 ? FINAL_BWD_ROWS : 1; }
__device__ __forceinline__ void final_bwd_block(const nasrec_final_desc_t& d, int K, int nA, int nB, int vb_, float* lds) {
  // dispatch order: the few long-running workgroups first (part C: one workgroup walks the batch; part B: 16 columns x the whole batch
  // each), the many one-element workgroups of part A behind them — as the item's tail they made a level that holds this item last
  // 3 us longer than the item itself (workgroups are dispatched in blockIdx order).  Same work per workgroup: same bits.
  const int nC = d.y != nullptr ? 1 : 0;
  const int vb = vb_ < nB + nC ? nA + vb_ : vb_ - (nB + nC);  // (below: [0, nA) = part A, [nA, nA + nB) = part B, nA + nB = part C)
  const bool fused = d.y != nullptr;
  // the per-sample gradient: both loads unconditional (a branch on `fused` around them, per use, is a branch per load — the compiler then
  // waits for each before it issues the next: fifteen serial round trips in part B's eight-deep loop); without the fused loss the second
  // pointer aliases the first and its value is dropped
  const float* const dl_a = fused ? d.logits : d.dlogits;
  const float* const dl_b = fused ? d.y : d.dlogits;
  const float dl_scale = d.grad_scale;
  auto dl = [&](int b) {
    const float z = dl_a[b], y = dl_b[b];
    return fused ? bce_grad(z, y, dl_scale) : z;
  };
  if (vb < nA && final_bwd_rows(d.B) > 1) {
    // part A at large batch: a workgroup = 256 consecutive columns x FINAL_BWD_ROWS samples.  The column's segment search, its weight and
    // the index arithmetic are done once per thread instead of once per element (one element per thread spent ~40 instructions on t / K,
    // t % K and the search to move 4 bytes: 1 TB/s at B = 8192); the samples' gradients are the same address in every lane.  Reads
    // first (gradients, the accumulation targets), one wait, then the stores.  Same value per element: same bits.
    constexpr int RB = FINAL_BWD_ROWS;
    const int kblocks = (K + 255) / 256;
    const int rblk = vb / kblocks, k = (vb - rblk * kblocks) * 256 + (int)threadIdx.x;
    if (k >= K) return;
    float* p0 = nullptr;
    long ldq = 0;
    bool accq = false;
    for (int q = 0; q < d.nseg; ++q) {
      const int jj = final_feat(d, q, k);
      if (jj >= 0) {
        if (d.dseg[q]) {
          p0 = d.dseg[q] + jj;
          ldq = d.ld[q];
          accq = d.dseg_accumulate[q] != 0;
        }
        break;
      }
    }
    if (!p0) return;
    const float wk = d.w[k];
    const int r0 = rblk * RB;
    float z[RB], y[RB], c[RB], g[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int b = min(r0 + u, d.B - 1);
      z[u] = dl_a[b];
      y[u] = dl_b[b];
      c[u] = 0.f;
    }
    if (accq) {
#pragma unroll
      for (int u = 0; u < RB; ++u) c[u] = p0[(long)min(r0 + u, d.B - 1) * ldq];
    }
    if (fused) {
#pragma unroll
      for (int u = 0; u < RB; ++u) g[u] = bce_grad(z[u], y[u], dl_scale);
    } else {
#pragma unroll
      for (int u = 0; u < RB; ++u) g[u] = z[u];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (r0 + u >= d.B) break;
      const float v = g[u] * wk;
      p0[(long)(r0 + u) * ldq] = accq ? c[u] + v : v;
    }
    return;
  }
  if (vb < nA) {
    const long t = (long)vb * 256 + threadIdx.x;
    if (t >= (long)d.B * K) return;
    const int b = (int)(t / K), k = (int)(t % K);
    for (int q = 0; q < d.nseg; ++q) {
      const int jj = final_feat(d, q, k);
      if (jj >= 0) {
        if (d.dseg[q]) {
          float* p = d.dseg[q] + (long)b * d.ld[q] + jj;
          const float v = dl(b) * d.w[k];
          *p = d.dseg_accumulate[q] ? *p + v : v;
        }
        break;
      }
    }
    return;
  }
  if (vb >= nA + nB) {  // part C
    float* redl = lds;
    float s = 0.f;
    for (int b = threadIdx.x; b < d.B; b += 256) {
      const float z = d.logits[b], y = d.y[b];
      s += bce_term(z, y);
      if (d.dlogits_out) d.dlogits_out[b] = bce_grad(z, y, d.grad_scale);
    }
    redl[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) redl[threadIdx.x] += redl[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0 && d.loss) d.loss[0] = redl[0] / (float)d.B;
    return;
  }
  if (d.nsplit > 1) {
    // large batch: workgroup = 64 columns (256-byte coalesced rows) x one of nsplit batch slices; 4 waves take every 4th row
    float(*reds)[64] = reinterpret_cast<float(*)[64]>(lds);
    const int nBk = nB / d.nsplit;
    const int blk = vb - nA, slice = blk / nBk, kl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = (blk - slice * nBk) * 64 + kl;
    const int b0 = (int)((long)d.B * slice / d.nsplit), b1 = (int)((long)d.B * (slice + 1) / d.nsplit);
    float s0 = 0.f, s1 = 0.f;
    if (k <= K) {
      const float* src = nullptr;
      int ld = 0, jj = 0;
      if (k < K) {
        for (int q = 0; q < d.nseg; ++q) {
          jj = final_feat(d, q, k);
          if (jj >= 0) {
            src = d.seg[q];
            ld = d.ld[q];
            break;
          }
        }
      }
      const bool has = k < K && src != nullptr;
      const float* sp = has ? src + jj : d.w;  // (an address that is always valid: the loads below are unconditional)
      const long sld = has ? ld : 0;
      int b = b0 + w;
      // eight rows' loads in flight — feature, logit, label: 24 — then ONE branch on the fused loss for the eight.  (Two rows per trip
      // with the branch per row — the label's load and its wait inside — were serial round trips, the kernel's time at B = 4096 / 8192.)
      // Rows alternate between the two accumulators exactly as before: same sums, same bits.
      for (; b + 28 < b1; b += 32) {
        float f[8], z[8], y[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          f[u] = sp[(long)(b + 4 * u) * sld];
          z[u] = dl_a[b + 4 * u];
          y[u] = dl_b[b + 4 * u];
        }
        if (fused) {
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = bce_grad(z[u], y[u], dl_scale);
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = z[u];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          s0 = fmaf(g[u], (k == K) ? 1.f : (has ? f[u] : 0.f), s0);
          s1 = fmaf(g[u + 1], (k == K) ? 1.f : (has ? f[u + 1] : 0.f), s1);
        }
      }
      for (; b + 4 < b1; b += 8) {
        const float f0 = (k == K) ? 1.f : (src ? src[(long)b * ld + jj] : 0.f);
        const float f1 = (k == K) ? 1.f : (src ? src[(long)(b + 4) * ld + jj] : 0.f);
        s0 = fmaf(dl(b), f0, s0);
        s1 = fmaf(dl(b + 4), f1, s1);
      }
      for (; b < b1; b += 4) s0 = fmaf(dl(b), (k == K) ? 1.f : (src ? src[(long)b * ld + jj] : 0.f), s0);
    }
    reds[w][kl] = s0 + s1;
    __syncthreads();
    if (w == 0 && k <= K) d.dw[(long)slice * (K + 1) + k] = (reds[0][kl] + reds[1][kl]) + (reds[2][kl] + reds[3][kl]);
    return;
  }
  float(*red)[17] = reinterpret_cast<float(*)[17]>(lds);
  const int kl = threadIdx.x & 15, bq = threadIdx.x >> 4;
  const int k = (vb - nA) * 16 + kl;  // k == K is the bias column
  float s = 0.f;
  if (k <= K) {
    const float* src = nullptr;
    int ld = 0, jj = 0;
    if (k < K) {
      for (int q = 0; q < d.nseg; ++q) {
        jj = final_feat(d, q, k);
        if (jj >= 0) {
          src = d.seg[q];
          ld = d.ld[q];
          break;
        }
      }
    }
    // eight samples' loads in flight (same summation order as the plain loop: sixteen dependent round trips at batch 256)
    const bool has = k < K && src != nullptr;
    const float* sp = has ? src + jj : d.w;  // (an address that is always valid: the loads below are unconditional)
    const int sld = has ? ld : 0;
    for (int b0 = bq; b0 < d.B; b0 += 16 * 8) {
      float f[8], g[8], z[8], y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // every load of the eight samples, nothing else: 24 in flight
        const int b = min(b0 + 16 * u, d.B - 1);
        f[u] = sp[(long)b * sld];
        z[u] = dl_a[b];
        y[u] = dl_b[b];
      }
      if (fused) {  // (ONE branch for the eight: inside the loop above it was a branch per sample with the label's load and a wait in it)
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = bce_grad(z[u], y[u], dl_scale);
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = z[u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float fu = (k == K) ? 1.f : (has ? f[u] : 0.f);
        if (b0 + 16 * u < d.B) s = fmaf(g[u], fu, s);
      }
    }
  }
  red[bq][kl] = s;
  __syncthreads();
  if (bq == 0 && k <= K) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][kl];
    if (k < K)
      d.dw[k] = tot;
    else
      d.dbias[0] = tot;
  }
}

// grid of the backward: nA element-wise workgroups (dseg), nB column blocks (dw, dbias), + 1 for the fused loss
__host__ __device__ inline void final_bwd_geometry(const nasrec_final_desc_t& d, int& K, int& nA, int& nB) {
  K = 0;
  for (int q = 0; q < d.nseg; ++q) K = K > d.off[q] + final_seg_extent(d, q) ? K : d.off[q] + final_seg_extent(d, q);
  const long tA = (long)d.B * K;
  nA = (int)((tA + 255) / 256);
  if (final_bwd_rows(d.B) > 1) nA = ((K + 255) / 256) * ((d.B + FINAL_BWD_ROWS - 1) / FINAL_BWD_ROWS);
  if (d.dseg_done) nA = 0;  // (NASREC_OP_FINAL_FUSED wrote the per-sample part)
  nB = (K + 1 + 15) / 16;
  if (d.nsplit > 1) nB = ((K + 1 + 63) / 64) * d.nsplit;
}
