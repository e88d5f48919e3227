// Embedding stem kernels (reference: nasrec/supernet/supernet.py:404-430 forward; the backward +
// optimizer semantics of nn.Embedding(sparse=False) + clip_grad_norm_ + Adagrad, train_utils.py:283-286).
//
// Layout: Fs independent tables [rows_f, 16] fp32 (64-byte rows), indices int64 [B, Fs], one id per
// field per sample.  A row is moved by 4 lanes x float4, so a wavefront moves 16 rows = 1 KiB per
// instruction and the output [B,Fs,16] is written fully coalesced.
#include "common.h"

__global__ __launch_bounds__(256) void embed_gather_kernel(const nasrec_embed_desc_t d) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long pair = t >> 2;
  const int q = (int)(t & 3);
  if (pair >= (long)d.B * d.Fs) return;
  const int f = (int)(pair % d.Fs);
  long row = d.idx[pair];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row >= 0 && row < d.rows[f]) {
    v = *reinterpret_cast<const float4*>(d.table[f] + row * NASREC_EMB_DIM + q * 4);
  } else if (d.oob) {
    *d.oob = 1;  // torch raises IndexError; the host shim turns this flag into one
  }
  *reinterpret_cast<float4*>(d.out + pair * NASREC_EMB_DIM + q * 4) = v;
}

int launch_embed_gather(hipStream_t st, const nasrec_embed_desc_t* d) {
  if (d->Fs < 1 || d->Fs > NASREC_MAX_TABLES) return nasrec_set_error(-2, "embed: Fs=%d out of range", d->Fs);
  long threads = (long)d->B * d->Fs * 4;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d);
  return nasrec_check_launch("embed_gather");
}

// Row-sparse backward: leader election + ordered duplicate summation, per field.
// grid = (Fs, ceil(B/256)); thread = one (b, f).  The B ids of the field are streamed through LDS in
// chunks; every lane compares against the same id at a time (LDS broadcast).  Leaders (first occurrence)
// add up the 64-byte gradient rows of all their occurrences in ascending b, so the result is
// reproducible run to run.
#define DEDUP_CHUNK 1024
__global__ __launch_bounds__(256) void emb_dedup_kernel(const nasrec_emb_dedup_desc_t d) {
  __shared__ long sidx[DEDUP_CHUNK];
  __shared__ float red[256];
  const int f = blockIdx.x;
  const int b = blockIdx.y * 256 + threadIdx.x;
  const bool live = b < d.B;
  const long my = live ? d.idx[(long)b * d.Fs + f] : -1;
  int lead = live ? 1 : 0;
  float g[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) g[e] = 0.f;
  for (int c0 = 0; c0 < d.B; c0 += DEDUP_CHUNK) {
    const int cn = min(DEDUP_CHUNK, d.B - c0);
    __syncthreads();
    for (int q = threadIdx.x; q < cn; q += 256) sidx[q] = d.idx[(long)(c0 + q) * d.Fs + f];
    __syncthreads();
    if (live) {
      // 8 ids per trip: the LDS broadcast reads are issued back to back, the (rare) match branch comes after
      for (int q8 = 0; q8 < cn; q8 += 8) {
        long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (q8 + u < cn) ? sidx[q8 + u] : -2;
        bool any = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) any |= (v[u] == my);
        if (!any) continue;
#pragma unroll 1
        for (int q = q8; q < min(q8 + 8, cn); ++q) {
        if (sidx[q] == my) {
          const int bp = c0 + q;
          if (bp < b) {
            lead = 0;
          } else if (lead) {
            const float4* src = reinterpret_cast<const float4*>(d.dout + ((long)bp * d.Fs + f) * 16);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              float4 x = src[v];
              g[4 * v + 0] += x.x;
              g[4 * v + 1] += x.y;
              g[4 * v + 2] += x.z;
              g[4 * v + 3] += x.w;
            }
          }
        }
        }
      }
    }
  }
  float ss = 0.f;
  if (live) {
    d.leader[(long)b * d.Fs + f] = lead;
    if (lead) {
      float4* dst = reinterpret_cast<float4*>(d.gsum + ((long)b * d.Fs + f) * 16);
#pragma unroll
      for (int v = 0; v < 4; ++v) dst[v] = make_float4(g[4 * v], g[4 * v + 1], g[4 * v + 2], g[4 * v + 3]);
#pragma unroll
      for (int e = 0; e < 16; ++e) ss += g[e] * g[e];
    }
  }
  red[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) d.sumsq_partial[(long)f * gridDim.y + blockIdx.y] = red[0];
}

int launch_emb_dedup(hipStream_t st, const nasrec_emb_dedup_desc_t* d) {
  if (d->B == 0) return 0;
  dim3 grid(d->Fs, (d->B + 255) / 256);
  hipLaunchKernelGGL(emb_dedup_kernel, grid, dim3(256), 0, st, *d);
  return nasrec_check_launch("emb_dedup");
}

// Row-sparse clip + Adagrad on the touched rows only (4 lanes x float4 per row).
__global__ __launch_bounds__(256) void adagrad_rows_kernel(const nasrec_adagrad_rows_desc_t d) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long pair = t >> 2;
  const int q = (int)(t & 3);
  if (pair >= (long)d.B * d.Fs) return;
  if (!d.leader[pair]) return;
  const int f = (int)(pair % d.Fs);
  const long row = d.idx[pair];
  const float lr = *d.lr, coef = *d.coef;
  float4 g = *reinterpret_cast<const float4*>(d.gsum + pair * 16 + q * 4);
  float4* sp = reinterpret_cast<float4*>(d.state[f] + row * 16 + q * 4);
  float4* pp = reinterpret_cast<float4*>(d.table[f] + row * 16 + q * 4);
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

int launch_adagrad_rows(hipStream_t st, const nasrec_adagrad_rows_desc_t* d) {
  long threads = (long)d->B * d->Fs * 4;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(adagrad_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d);
  return nasrec_check_launch("adagrad_rows");
}
