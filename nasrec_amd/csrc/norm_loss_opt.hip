// Final logit + BCE loss, LayerNorm, and the optimizer tail (clip_grad_norm_ + Adagrad) of the training step
// (reference: supernet.py:592-598/657-664, main_train.py:122, train_utils.py:262-286, main_train.py:152-154).
#include "common.h"
#include "optimizer_bodies.h"

#include "final_bodies.h"

__global__ __launch_bounds__(256) void final_fwd_kernel(const nasrec_final_desc_t d) {
  __shared__ float red[4];
  final_fwd_block(d, blockIdx.x, red);
}

__global__ __launch_bounds__(256) void final_bwd_kernel(const nasrec_final_desc_t d, int K, int nA, int nB) {
  __shared__ float lds[FINAL_BWD_LDS_FLOATS];
  final_bwd_block(d, K, nA, nB, blockIdx.x, lds);
}

__global__ __launch_bounds__(256) void final_fused_kernel(const nasrec_final_desc_t d) {
  __shared__ float red[4];
  final_fused_block(d, blockIdx.x, red);
}

int final_fused_check(const nasrec_final_desc_t* d) {
  if (!d->y || !d->logits || !d->bias || !d->w) return nasrec_set_error(-2, "final_fused: needs y, logits, bias and w");
  if (d->nsplit > 1) return nasrec_set_error(-2, "final_fused: nsplit=%d (the batch-sliced backward keeps its own launch)", d->nsplit);
  for (int q = 0; q < d->nseg; ++q)
    if (d->dseg[q] && !d->seg[q]) return nasrec_set_error(-2, "final_fused: segment %d has a gradient destination and no input", q);
  return 0;
}

int launch_final(hipStream_t st, const nasrec_final_desc_t* d) {
  if (d->B == 0) return 0;
  if (d->kind == NASREC_OP_FINAL_FUSED) {
    const int rc = final_fused_check(d);
    if (rc) return rc;
    hipLaunchKernelGGL(final_fused_kernel, dim3(d->B), dim3(256), 0, st, *d);
  } else if (d->kind == NASREC_OP_FINAL_FWD) {
    hipLaunchKernelGGL(final_fwd_kernel, dim3(d->B), dim3(256), 0, st, *d);
  } else {
    int K, nA, nB;
    final_bwd_geometry(*d, K, nA, nB);
    if (d->y != nullptr && d->logits == nullptr) return nasrec_set_error(-2, "final_bwd: fused BCE needs desc.logits");
    hipLaunchKernelGGL(final_bwd_kernel, dim3(nA + nB + (d->y != nullptr ? 1 : 0)), dim3(256), 0, st, *d, K, nA, nB);
  }
  return nasrec_check_launch("final");
}

// ---------------------------------------------------------------------------------------------------
// BCE-with-logits (mean) + dlogits; one workgroup, fixed-order reduction
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bce_kernel(const nasrec_bce_desc_t d) {
  __shared__ float red[1024];
  float s = 0.f;
  for (int b = threadIdx.x; b < d.B; b += 1024) {
    const float z = d.logits[b], y = d.y[b];
    s += bce_term(z, y);
    d.dlogits[b] = bce_grad(z, y, d.grad_scale);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) d.loss[0] = red[0] / (float)d.B;
}

int launch_bce(hipStream_t st, const nasrec_bce_desc_t* d) {
  if (d->B == 0) return 0;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, st, *d);
  return nasrec_check_launch("bce");
}

// ---------------------------------------------------------------------------------------------------
// global-norm clip + Adagrad on the flat dense parameter arena
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const nasrec_sumsq_desc_t d) {
  __shared__ float red[256];
  sumsq_body(d, blockIdx.x, gridDim.x, red);
}

int launch_sumsq(hipStream_t st, const nasrec_sumsq_desc_t* d) {
  if (d->nblocks < 1) return nasrec_set_error(-2, "sumsq: nblocks=%d", d->nblocks);
  hipLaunchKernelGGL(sumsq_kernel, dim3(d->nblocks), dim3(256), 0, st, *d);
  return nasrec_check_launch("sumsq");
}

__global__ __launch_bounds__(64) void clip_coef_kernel(const nasrec_clip_coef_desc_t d) {
  float total;
  const float coef = clip_coef_wave(d, threadIdx.x, &total);
  if (threadIdx.x == 0) {
    d.out[0] = coef;
    d.out[1] = total;
  }
}

int launch_clip_coef(hipStream_t st, const nasrec_clip_coef_desc_t* d) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, *d);
  return nasrec_check_launch("clip_coef");
}

__global__ __launch_bounds__(256) void adagrad_dense_kernel(const nasrec_adagrad_dense_desc_t d) {
  adagrad_dense_body(d, blockIdx.x, gridDim.x, *d.lr, *d.coef);
}

int launch_adagrad_dense(hipStream_t st, const nasrec_adagrad_dense_desc_t* d) {
  if (d->n == 0 || (d->chunks && d->nchunks < 1)) return 0;
  const long blocks = d->chunks ? (d->nchunks > 2048 ? 2048 : d->nchunks) : adagrad_dense_blocks(d->n);
  hipLaunchKernelGGL(adagrad_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *d);
  return nasrec_check_launch("adagrad_dense");
}

// ---------------------------------------------------------------------------------------------------
// chunk tables (the arena ranges of one sampled path): zero them / write the table itself
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void memset_chunks_kernel(const nasrec_memset_desc_t d) {
  for (long c = blockIdx.x; c < d.nchunks; c += gridDim.x) {
    float* x = reinterpret_cast<float*>(d.ptr) + d.chunks[2 * c];
    const long n = d.chunks[2 * c + 1], n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) *reinterpret_cast<f32x4*>(x + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long j = 4 * n4 + threadIdx.x; j < n; j += 256) x[j] = 0.f;
  }
}

// NASREC_OP_MEMSET over one flat range, as a KERNEL (round 5): hipMemsetAsync becomes a memset node of whatever graph captures the step,
// and inside a torch-captured graph those nodes were not ordered against the kernels around them on this stack (tests/
// test_sharded_tables_gpu.py: gradients accumulated onto a buffer the memset had not cleared yet, from the second replay on) — a kernel
// node is ordered like every other launch of the program.  bytes is a multiple of 4 (fp32 / int32 buffers).
__global__ __launch_bounds__(256) void memset_flat_kernel(float* x, long n) {
  const long n4 = n >> 2, stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) *reinterpret_cast<f32x4*>(x + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long j = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) x[j] = 0.f;
}

int launch_memset_flat(hipStream_t st, const nasrec_memset_desc_t* d) {
  if (d->bytes <= 0) return 0;
  if ((d->bytes & 3) || ((unsigned long long)d->ptr & 15)) {  // (not a 16-byte aligned fp32 range: the runtime's memset)
    hipError_t e = hipMemsetAsync(d->ptr, 0, (size_t)d->bytes, st);
    if (e != hipSuccess) return nasrec_set_error((int)e, "memset: %s", hipGetErrorString(e));
    return 0;
  }
  const long n = d->bytes >> 2;
  long blocks = ((n >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(memset_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float*>(d->ptr), n);
  return nasrec_check_launch("memset");
}

int launch_memset_chunks(hipStream_t st, const nasrec_memset_desc_t* d) {
  if (d->nchunks < 1) return 0;
  hipLaunchKernelGGL(memset_chunks_kernel, dim3((unsigned)(d->nchunks > 4096 ? 4096 : d->nchunks)), dim3(256), 0, st, *d);
  return nasrec_check_launch("memset_chunks");
}

__global__ __launch_bounds__(256) void const_i64_kernel(const nasrec_const_i64_desc_t d) {
  for (int i = threadIdx.x; i < d.n; i += 256) d.dst[i] = d.vals[i];
}

int launch_const_i64(hipStream_t st, const nasrec_const_i64_desc_t* d) {
  if (d->n < 0 || d->n > NASREC_CONST_I64_MAX) return nasrec_set_error(-2, "const_i64: n=%d outside [0,%d]", d->n, NASREC_CONST_I64_MAX);
  if (d->n == 0) return 0;
  hipLaunchKernelGGL(const_i64_kernel, dim3(1), dim3(256), 0, st, *d);
  return nasrec_check_launch("const_i64");
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm (supernet mode: every projection is followed by LN -> activation -> prefix mask,
// modules.py:174-178).  Row r, element i:  KC: x[r*ld + i];  TOKR: x[(r>>4)*ld + (r&15) + i*16].
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ long ln_off(int r, int i, int ld) {
  if (MODE == NASREC_AM_KC) return (long)r * ld + i;
  return (long)(r >> 4) * ld + (r & 15) + (long)i * 16;
}

__device__ __forceinline__ float act_grad(float u, int act) {
  if (act == NASREC_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == NASREC_ACT_SILU) {
    const float s = 1.f / (1.f + __expf(-u));
    return s * (1.f + u * (1.f - s));
  }
  if (act == NASREC_ACT_SIGMOID) {
    const float s = 1.f / (1.f + __expf(-u));
    return s * (1.f - s);
  }
  return 1.f;
}

// dense rows: one wavefront per row
__global__ __launch_bounds__(256) void ln_fwd_kc_kernel(const nasrec_layernorm_desc_t d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= d.R) return;
  const float* x = d.x + (long)r * d.ldx;
  float s = 0.f;
  for (int i = lane; i < d.D; i += 64) s += x[i];
  const float mu = wave_sum(s) / (float)d.D;
  float v = 0.f;
  for (int i = lane; i < d.D; i += 64) {
    const float c = x[i] - mu;
    v = fmaf(c, c, v);
  }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)d.D + d.eps);
  if (lane == 0) {
    d.stats[2 * r] = mu;
    d.stats[2 * r + 1] = rstd;
  }
  float* y = d.y + (long)r * d.ldy;
  for (int i = lane; i < d.D; i += 64) {
    float u = fmaf((x[i] - mu) * rstd, d.w[i], d.b[i]);
    u = act_apply(u, d.act);
    if (d.dims_in_use >= 0 && i >= d.dims_in_use) u = 0.f;
    y[i] = d.accumulate ? y[i] + u : u;
  }
}

// token-axis rows: one thread per (b,e) row, D = N' <= 64 elements strided by 16 floats
__global__ __launch_bounds__(256) void ln_fwd_tokr_kernel(const nasrec_layernorm_desc_t d) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= d.R) return;
  float s = 0.f;
  for (int i = 0; i < d.D; ++i) s += d.x[ln_off<NASREC_AM_TOKR>(r, i, d.ldx)];
  const float mu = s / (float)d.D;
  float v = 0.f;
  for (int i = 0; i < d.D; ++i) {
    const float c = d.x[ln_off<NASREC_AM_TOKR>(r, i, d.ldx)] - mu;
    v = fmaf(c, c, v);
  }
  const float rstd = 1.f / sqrtf(v / (float)d.D + d.eps);
  d.stats[2 * r] = mu;
  d.stats[2 * r + 1] = rstd;
  for (int i = 0; i < d.D; ++i) {
    float u = fmaf((d.x[ln_off<NASREC_AM_TOKR>(r, i, d.ldx)] - mu) * rstd, d.w[i], d.b[i]);
    u = act_apply(u, d.act);
    if (d.dims_in_use >= 0 && i >= d.dims_in_use) u = 0.f;
    float* y = d.y + ln_off<NASREC_AM_TOKR>(r, i, d.ldy);
    *y = d.accumulate ? *y + u : u;
  }
}

// backward, dense rows: workgroup = 4 waves, grid-strided over rows; per-lane partial (dw, db) for its
// columns, combined across the 4 waves in LDS -> dwb_partial[blk][2D].  D <= 1024.
#define LN_MAXD 1024
__global__ __launch_bounds__(256) void ln_bwd_kc_kernel(const nasrec_layernorm_desc_t d) {
  __shared__ float sdw[4][LN_MAXD];
  __shared__ float sdb[4][LN_MAXD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float pdw[LN_MAXD / 64], pdb[LN_MAXD / 64];
#pragma unroll
  for (int c = 0; c < LN_MAXD / 64; ++c) pdw[c] = pdb[c] = 0.f;
  for (int r = blockIdx.x * 4 + wave; r < d.R; r += gridDim.x * 4) {
    const float* x = d.x + (long)r * d.ldx;
    const float* dy = d.dy + (long)r * d.ldy;
    const float mu = d.stats[2 * r], rstd = d.stats[2 * r + 1];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXD / 64; ++c) {
      const int i = lane + 64 * c;
      if (i < d.D) {
        const float xh = (x[i] - mu) * rstd;
        float g = dy[i];
        if (d.dims_in_use >= 0 && i >= d.dims_in_use) g = 0.f;
        if (d.act != NASREC_ACT_NONE) g *= act_grad(fmaf(xh, d.w[i], d.b[i]), d.act);
        pdw[c] = fmaf(g, xh, pdw[c]);
        pdb[c] += g;
        const float gw = g * d.w[i];
        c1 += gw;
        c2 = fmaf(gw, xh, c2);
      }
    }
    c1 = wave_sum(c1) / (float)d.D;
    c2 = wave_sum(c2) / (float)d.D;
    float* dx = d.dx + (long)r * d.ldx;
#pragma unroll
    for (int c = 0; c < LN_MAXD / 64; ++c) {
      const int i = lane + 64 * c;
      if (i < d.D) {
        const float xh = (x[i] - mu) * rstd;
        float g = dy[i];
        if (d.dims_in_use >= 0 && i >= d.dims_in_use) g = 0.f;
        if (d.act != NASREC_ACT_NONE) g *= act_grad(fmaf(xh, d.w[i], d.b[i]), d.act);
        const float v = rstd * (g * d.w[i] - c1 - xh * c2);
        dx[i] = d.accumulate ? dx[i] + v : v;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < LN_MAXD / 64; ++c) {
    const int i = lane + 64 * c;
    if (i < d.D) {
      sdw[wave][i] = pdw[c];
      sdb[wave][i] = pdb[c];
    }
  }
  __syncthreads();
  float* out = d.dwb_partial + (long)blockIdx.x * 2 * d.D;
  for (int i = threadIdx.x; i < d.D; i += 256) {
    out[i] = (sdw[0][i] + sdw[1][i]) + (sdw[2][i] + sdw[3][i]);
    out[d.D + i] = (sdb[0][i] + sdb[1][i]) + (sdb[2][i] + sdb[3][i]);
  }
}

// backward, token-axis rows: thread = (b,e) row; dw/db reduced over the workgroup's 256 rows per element i
__global__ __launch_bounds__(256) void ln_bwd_tokr_kernel(const nasrec_layernorm_desc_t d) {
  __shared__ float red[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 256 + threadIdx.x;
  const bool live = r < d.R;
  const float mu = live ? d.stats[2 * r] : 0.f, rstd = live ? d.stats[2 * r + 1] : 0.f;
  float c1 = 0.f, c2 = 0.f;
  if (live) {
    for (int i = 0; i < d.D; ++i) {
      const float xh = (d.x[ln_off<NASREC_AM_TOKR>(r, i, d.ldx)] - mu) * rstd;
      float g = d.dy[ln_off<NASREC_AM_TOKR>(r, i, d.ldy)];
      if (d.dims_in_use >= 0 && i >= d.dims_in_use) g = 0.f;
      if (d.act != NASREC_ACT_NONE) g *= act_grad(fmaf(xh, d.w[i], d.b[i]), d.act);
      const float gw = g * d.w[i];
      c1 += gw;
      c2 = fmaf(gw, xh, c2);
    }
    c1 /= (float)d.D;
    c2 /= (float)d.D;
  }
  for (int i = 0; i < d.D; ++i) {
    float g = 0.f, xh = 0.f;
    if (live) {
      xh = (d.x[ln_off<NASREC_AM_TOKR>(r, i, d.ldx)] - mu) * rstd;
      g = d.dy[ln_off<NASREC_AM_TOKR>(r, i, d.ldy)];
      if (d.dims_in_use >= 0 && i >= d.dims_in_use) g = 0.f;
      if (d.act != NASREC_ACT_NONE) g *= act_grad(fmaf(xh, d.w[i], d.b[i]), d.act);
      const float v = rstd * (g * d.w[i] - c1 - xh * c2);
      float* dx = d.dx + ln_off<NASREC_AM_TOKR>(r, i, d.ldx);
      *dx = d.accumulate ? *dx + v : v;
    }
    const float sw = wave_sum(g * xh), sb = wave_sum(g);
    if (lane == 0) {
      red[wave][0][i] = sw;
      red[wave][1][i] = sb;
    }
  }
  __syncthreads();
  float* out = d.dwb_partial + (long)blockIdx.x * 2 * d.D;
  for (int i = threadIdx.x; i < d.D; i += 256) {
    out[i] = (red[0][0][i] + red[1][0][i]) + (red[2][0][i] + red[3][0][i]);
    out[d.D + i] = (red[0][1][i] + red[1][1][i]) + (red[2][1][i] + red[3][1][i]);
  }
}

// ---- register-resident rows ---------------------------------------------------------------------------------------------------
// A LayerNorm row is read ONCE: dense rows (D <= 1024) live in 4 * NV registers of a wavefront's lanes (16-byte loads and
// stores), token-axis rows (D = N' <= 64) in up to 64 registers of one thread.  Mean, variance, normalisation (forward) and
// both reductions plus dx (backward) run on the registers; the general kernels above re-read the row from global memory for
// every pass and serve unaligned rows (ld or D not a multiple of 4).  Bytes per row: forward 2 * 4D, backward 3 * 4D — the
// HBM-streaming roofline these kernels are measured against.
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kc_vec_kernel(const nasrec_layernorm_desc_t d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= d.R) return;
  const float* x = d.x + (long)r * d.ldx;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i = 4 * lane + 256 * c;
    v[c] = (i < d.D) ? *reinterpret_cast<const f32x4*>(x + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  }
  const float mu = wave_sum(s) / (float)d.D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i = 4 * lane + 256 * c;
    if (i < d.D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = v[c][e] - mu;
        q = fmaf(t, t, q);
      }
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)d.D + d.eps);
  if (lane == 0) {
    d.stats[2 * r] = mu;
    d.stats[2 * r + 1] = rstd;
  }
  float* y = d.y + (long)r * d.ldy;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i = 4 * lane + 256 * c;
    if (i < d.D) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(d.w + i), bb = *reinterpret_cast<const f32x4*>(d.b + i);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = fmaf((v[c][e] - mu) * rstd, w[e], bb[e]);
        u = act_apply(u, d.act);
        if (d.dims_in_use >= 0 && i + e >= d.dims_in_use) u = 0.f;
        o[e] = u;
      }
      f32x4* yp = reinterpret_cast<f32x4*>(y + i);
      if (d.accumulate) o = o + *yp;
      *yp = o;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kc_vec_kernel(const nasrec_layernorm_desc_t d) {
  __shared__ float sdw[4][LN_MAXD];
  __shared__ float sdb[4][LN_MAXD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 pdw[NV], pdb[NV], w[NV], bb[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i = 4 * lane + 256 * c;
    pdw[c] = pdb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    w[c] = (i < d.D) ? *reinterpret_cast<const f32x4*>(d.w + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bb[c] = (i < d.D && d.act != NASREC_ACT_NONE) ? *reinterpret_cast<const f32x4*>(d.b + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int r = blockIdx.x * 4 + wave; r < d.R; r += gridDim.x * 4) {
    const float* x = d.x + (long)r * d.ldx;
    const float* dy = d.dy + (long)r * d.ldy;
    const float mu = d.stats[2 * r], rstd = d.stats[2 * r + 1];
    f32x4 xh[NV], g[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i = 4 * lane + 256 * c;
      if (i < d.D) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i);
        g[c] = *reinterpret_cast<const f32x4*>(dy + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[c][e] = (xv[e] - mu) * rstd;
          if (d.dims_in_use >= 0 && i + e >= d.dims_in_use) g[c][e] = 0.f;
          if (d.act != NASREC_ACT_NONE) g[c][e] *= act_grad(fmaf(xh[c][e], w[c][e], bb[c][e]), d.act);
          pdw[c][e] = fmaf(g[c][e], xh[c][e], pdw[c][e]);
          pdb[c][e] += g[c][e];
          const float gw = g[c][e] * w[c][e];
          c1 += gw;
          c2 = fmaf(gw, xh[c][e], c2);
        }
      }
    }
    c1 = wave_sum(c1) / (float)d.D;
    c2 = wave_sum(c2) / (float)d.D;
    float* dx = d.dx + (long)r * d.ldx;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i = 4 * lane + 256 * c;
      if (i < d.D) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (g[c][e] * w[c][e] - c1 - xh[c][e] * c2);
        f32x4* dp = reinterpret_cast<f32x4*>(dx + i);
        if (d.accumulate) o = o + *dp;
        *dp = o;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i = 4 * lane + 256 * c;
    if (i < d.D) {
      *reinterpret_cast<f32x4*>(&sdw[wave][i]) = pdw[c];
      *reinterpret_cast<f32x4*>(&sdb[wave][i]) = pdb[c];
    }
  }
  __syncthreads();
  float* out = d.dwb_partial + (long)blockIdx.x * 2 * d.D;
  for (int i = threadIdx.x; i < d.D; i += 256) {
    out[i] = (sdw[0][i] + sdw[1][i]) + (sdw[2][i] + sdw[3][i]);
    out[d.D + i] = (sdb[0][i] + sdb[1][i]) + (sdb[2][i] + sdb[3][i]);
  }
}

// token-axis rows held in registers: thread = (b,e) row, NR = 16 / 32 / 48 / 64 >= D elements
template <int NR>
__global__ __launch_bounds__(256) void ln_fwd_tokr_reg_kernel(const nasrec_layernorm_desc_t d) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= d.R) return;
  const float* x = d.x + (long)(r >> 4) * d.ldx + (r & 15);
  float v[NR];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    v[i] = (i < d.D) ? x[i * 16] : 0.f;
    s += v[i];
  }
  const float mu = s / (float)d.D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const float c = (i < d.D) ? v[i] - mu : 0.f;
    q = fmaf(c, c, q);
  }
  const float rstd = 1.f / sqrtf(q / (float)d.D + d.eps);
  d.stats[2 * r] = mu;
  d.stats[2 * r + 1] = rstd;
  float* y = d.y + (long)(r >> 4) * d.ldy + (r & 15);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (i < d.D) {
      float u = fmaf((v[i] - mu) * rstd, d.w[i], d.b[i]);
      u = act_apply(u, d.act);
      if (d.dims_in_use >= 0 && i >= d.dims_in_use) u = 0.f;
      y[i * 16] = d.accumulate ? y[i * 16] + u : u;
    }
  }
}

template <int NR>
__global__ __launch_bounds__(256) void ln_bwd_tokr_reg_kernel(const nasrec_layernorm_desc_t d) {
  __shared__ float red[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 256 + threadIdx.x;
  const bool live = r < d.R;
  const float mu = live ? d.stats[2 * r] : 0.f, rstd = live ? d.stats[2 * r + 1] : 0.f;
  const float* x = d.x + (long)(r >> 4) * d.ldx + (r & 15);
  const float* dy = d.dy + (long)(r >> 4) * d.ldy + (r & 15);
  float xh[NR], g[NR];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    xh[i] = 0.f;
    g[i] = 0.f;
    if (live && i < d.D) {
      xh[i] = (x[i * 16] - mu) * rstd;
      float gg = dy[i * 16];
      if (d.dims_in_use >= 0 && i >= d.dims_in_use) gg = 0.f;
      if (d.act != NASREC_ACT_NONE) gg *= act_grad(fmaf(xh[i], d.w[i], d.b[i]), d.act);
      g[i] = gg;
      const float gw = gg * d.w[i];
      c1 += gw;
      c2 = fmaf(gw, xh[i], c2);
    }
  }
  c1 /= (float)d.D;
  c2 /= (float)d.D;
  float* dx = d.dx + (long)(r >> 4) * d.ldx + (r & 15);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (i < d.D) {  // uniform
      if (live) {
        const float v = rstd * (g[i] * d.w[i] - c1 - xh[i] * c2);
        dx[i * 16] = d.accumulate ? dx[i * 16] + v : v;
      }
      const float sw = wave_sum(g[i] * xh[i]), sb = wave_sum(g[i]);
      if (lane == 0) {
        red[wave][0][i] = sw;
        red[wave][1][i] = sb;
      }
    }
  }
  __syncthreads();
  float* out = d.dwb_partial + (long)blockIdx.x * 2 * d.D;
  for (int i = threadIdx.x; i < d.D; i += 256) {
    out[i] = (red[0][0][i] + red[1][0][i]) + (red[2][0][i] + red[3][0][i]);
    out[d.D + i] = (red[0][1][i] + red[1][1][i]) + (red[2][1][i] + red[3][1][i]);
  }
}

// token-axis rows, a wavefront per SAMPLE: the sample's [D, 16] block is contiguous, so the wave reads it with up to four 16-byte
// loads per lane (1 KB per instruction, every byte used once) instead of D 4-byte loads per thread 64 bytes apart.  float4 #q of
// lane l covers token q * 16 + (l >> 2), columns e = 4 (l & 3) .. + 3: the per-(sample, e) statistics are sums over the lanes
// with equal l & 3 (shuffles over lane bits 2..5), the per-token parameter gradients sums over lane bits 0..1 and the samples.
__device__ __forceinline__ float tok_col_sum(float v) {  // over the 16 lanes that hold the same columns
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__global__ __launch_bounds__(256) void ln_fwd_tok_wave_kernel(const nasrec_layernorm_desc_t d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave, B = d.R >> 4, n4 = d.D * 4;
  if (b >= B) return;
  const float* x = d.x + (long)b * d.ldx;
  f32x4 v[4];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = q * 64 + lane;
    v[q] = f < n4 ? *reinterpret_cast<const f32x4*>(x + 4 * f) : (f32x4){0.f, 0.f, 0.f, 0.f};
    s = s + v[q];
  }
  f32x4 mu, rstd;
#pragma unroll
  for (int c = 0; c < 4; ++c) mu[c] = tok_col_sum(s[c]) / (float)d.D;
  f32x4 qq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (q * 64 + lane < n4) {
      const f32x4 t = v[q] - mu;
      qq = qq + t * t;
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) rstd[c] = 1.f / sqrtf(tok_col_sum(qq[c]) / (float)d.D + d.eps);
  if (lane < 4) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = b * 16 + 4 * lane + c;
      d.stats[2 * r] = mu[c];
      d.stats[2 * r + 1] = rstd[c];
    }
  }
  float* y = d.y + (long)b * d.ldy;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = q * 64 + lane;
    if (f < n4) {
      const int i = f >> 2;
      const float w = d.w[i], bb = d.b[i];
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float u = fmaf((v[q][c] - mu[c]) * rstd[c], w, bb);
        u = act_apply(u, d.act);
        if (d.dims_in_use >= 0 && i >= d.dims_in_use) u = 0.f;
        o[c] = u;
      }
      f32x4* yp = reinterpret_cast<f32x4*>(y + 4 * f);
      if (d.accumulate) o = o + *yp;
      *yp = o;
    }
  }
}

// backward: workgroup blk owns the 16 samples [16 blk, 16 blk + 16) (= the 256 rows the plan sized dwb_partial for), 4 per wave
__global__ __launch_bounds__(256) void ln_bwd_tok_wave_kernel(const nasrec_layernorm_desc_t d) {
  __shared__ float red[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int B = d.R >> 4, n4 = d.D * 4;
  float w[4], bb[4], pdw[4], pdb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = q * 16 + (lane >> 2);
    w[q] = i < d.D ? d.w[i] : 0.f;
    bb[q] = (i < d.D && d.act != NASREC_ACT_NONE) ? d.b[i] : 0.f;
    pdw[q] = 0.f;
    pdb[q] = 0.f;
  }
  for (int sidx = 0; sidx < 4; ++sidx) {
    const int b = blockIdx.x * 16 + sidx * 4 + wave;  // (uniform per wave)
    if (b >= B) break;
    const float* x = d.x + (long)b * d.ldx;
    const float* dy = d.dy + (long)b * d.ldy;
    f32x4 mu, rstd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = b * 16 + 4 * (lane & 3) + c;
      mu[c] = d.stats[2 * r];
      rstd[c] = d.stats[2 * r + 1];
    }
    f32x4 xh[4], g[4];
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = q * 64 + lane;
      xh[q] = g[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (f < n4) {
        const int i = f >> 2;
        xh[q] = (*reinterpret_cast<const f32x4*>(x + 4 * f) - mu) * rstd;
        f32x4 gg = *reinterpret_cast<const f32x4*>(dy + 4 * f);
        if (d.dims_in_use >= 0 && i >= d.dims_in_use) gg = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (d.act != NASREC_ACT_NONE) {
#pragma unroll
          for (int c = 0; c < 4; ++c) gg[c] *= act_grad(fmaf(xh[q][c], w[q], bb[q]), d.act);
        }
        g[q] = gg;
        pdw[q] += (gg[0] * xh[q][0] + gg[1] * xh[q][1]) + (gg[2] * xh[q][2] + gg[3] * xh[q][3]);
        pdb[q] += (gg[0] + gg[1]) + (gg[2] + gg[3]);
        const f32x4 gw = gg * w[q];
        c1 = c1 + gw;
        c2 = c2 + gw * xh[q];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      c1[c] = tok_col_sum(c1[c]) / (float)d.D;
      c2[c] = tok_col_sum(c2[c]) / (float)d.D;
    }
    float* dx = d.dx + (long)b * d.ldx;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = q * 64 + lane;
      if (f < n4) {
        f32x4 o = rstd * (g[q] * w[q] - c1 - xh[q] * c2);
        f32x4* dp = reinterpret_cast<f32x4*>(dx + 4 * f);
        if (d.accumulate) o = o + *dp;
        *dp = o;
      }
    }
  }
  // per-token parameter gradients: the 4 lanes of a token, then the 4 waves in fixed order
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float sw = pdw[q], sb = pdb[q];
    sw += __shfl_xor(sw, 1, 64);
    sw += __shfl_xor(sw, 2, 64);
    sb += __shfl_xor(sb, 1, 64);
    sb += __shfl_xor(sb, 2, 64);
    if ((lane & 3) == 0) {
      red[wave][0][q * 16 + (lane >> 2)] = sw;
      red[wave][1][q * 16 + (lane >> 2)] = sb;
    }
  }
  __syncthreads();
  float* out = d.dwb_partial + (long)blockIdx.x * 2 * d.D;
  for (int i = threadIdx.x; i < d.D; i += 256) {
    out[i] = (red[0][0][i] + red[1][0][i]) + (red[2][0][i] + red[3][0][i]);
    out[d.D + i] = (red[0][1][i] + red[1][1][i]) + (red[2][1][i] + red[3][1][i]);
  }
}

static bool ln_tok_wave_ok(const nasrec_layernorm_desc_t* d, bool fwd) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if ((d->R & 15) || (d->ldx & 3) || (d->ldy & 3) || !al(d->x)) return false;
  return fwd ? al(d->y) : (al(d->dy) && al(d->dx));
}

static bool ln_vec_ok(const nasrec_layernorm_desc_t* d, bool fwd) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if ((d->D & 3) || (d->ldx & 3) || (d->ldy & 3)) return false;
  if (!al(d->x) || !al(d->w) || !al(d->b)) return false;
  return fwd ? al(d->y) : (al(d->dy) && al(d->dx));
}

int launch_layernorm(hipStream_t st, const nasrec_layernorm_desc_t* d) {
  if (d->R == 0) return 0;
  const bool fwd = d->kind == NASREC_OP_LAYERNORM_FWD;
  if (d->mode == NASREC_AM_KC) {
    if (d->D > LN_MAXD) return nasrec_set_error(-2, "layernorm: D=%d > %d", d->D, LN_MAXD);
    const bool vec = ln_vec_ok(d, fwd);
    const int nv = d->D <= 256 ? 1 : (d->D <= 512 ? 2 : 4);
    if (fwd) {
      const dim3 grid((d->R + 3) / 4);
      if (vec && nv == 1) hipLaunchKernelGGL(ln_fwd_kc_vec_kernel<1>, grid, dim3(256), 0, st, *d);
      else if (vec && nv == 2) hipLaunchKernelGGL(ln_fwd_kc_vec_kernel<2>, grid, dim3(256), 0, st, *d);
      else if (vec) hipLaunchKernelGGL(ln_fwd_kc_vec_kernel<4>, grid, dim3(256), 0, st, *d);
      else hipLaunchKernelGGL(ln_fwd_kc_kernel, grid, dim3(256), 0, st, *d);
    } else {
      if (d->nblk < 1) return nasrec_set_error(-2, "layernorm bwd: nblk=%d", d->nblk);
      const dim3 grid(d->nblk);
      if (vec && nv == 1) hipLaunchKernelGGL(ln_bwd_kc_vec_kernel<1>, grid, dim3(256), 0, st, *d);
      else if (vec && nv == 2) hipLaunchKernelGGL(ln_bwd_kc_vec_kernel<2>, grid, dim3(256), 0, st, *d);
      else if (vec) hipLaunchKernelGGL(ln_bwd_kc_vec_kernel<4>, grid, dim3(256), 0, st, *d);
      else hipLaunchKernelGGL(ln_bwd_kc_kernel, grid, dim3(256), 0, st, *d);
    }
  } else if (d->mode == NASREC_AM_TOKR) {
    if (d->D > 64) return nasrec_set_error(-2, "layernorm(tok): D=%d > 64", d->D);
    const int nb = (d->R + 255) / 256;
    const bool wave_form = ln_tok_wave_ok(d, fwd);  // a wavefront per sample, 16-byte accesses (needs aligned sample blocks)
    if (fwd) {
      if (wave_form) hipLaunchKernelGGL(ln_fwd_tok_wave_kernel, dim3(((d->R >> 4) + 3) / 4), dim3(256), 0, st, *d);
      else if (d->D <= 16) hipLaunchKernelGGL(ln_fwd_tokr_reg_kernel<16>, dim3(nb), dim3(256), 0, st, *d);
      else if (d->D <= 32) hipLaunchKernelGGL(ln_fwd_tokr_reg_kernel<32>, dim3(nb), dim3(256), 0, st, *d);
      else if (d->D <= 48) hipLaunchKernelGGL(ln_fwd_tokr_reg_kernel<48>, dim3(nb), dim3(256), 0, st, *d);
      else hipLaunchKernelGGL(ln_fwd_tokr_reg_kernel<64>, dim3(nb), dim3(256), 0, st, *d);
    } else {
      if (d->nblk != nb) return nasrec_set_error(-2, "layernorm(tok) bwd: nblk=%d, want %d", d->nblk, nb);
      if (wave_form) hipLaunchKernelGGL(ln_bwd_tok_wave_kernel, dim3(nb), dim3(256), 0, st, *d);
      else if (d->D <= 16) hipLaunchKernelGGL(ln_bwd_tokr_reg_kernel<16>, dim3(nb), dim3(256), 0, st, *d);
      else if (d->D <= 32) hipLaunchKernelGGL(ln_bwd_tokr_reg_kernel<32>, dim3(nb), dim3(256), 0, st, *d);
      else if (d->D <= 48) hipLaunchKernelGGL(ln_bwd_tokr_reg_kernel<48>, dim3(nb), dim3(256), 0, st, *d);
      else hipLaunchKernelGGL(ln_bwd_tokr_reg_kernel<64>, dim3(nb), dim3(256), 0, st, *d);
    }
  } else {
    return nasrec_set_error(-2, "layernorm: unsupported mode %d", d->mode);
  }
  return nasrec_check_launch("layernorm");
}

