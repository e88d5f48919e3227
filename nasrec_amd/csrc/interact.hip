// Feature-interaction and glue kernels: DotProduct triangle, FactorizationMachine, SigmoidGating backward,
// segmented copies, bias-gradient row sums, fixed-order row reductions.
// All are HBM/latency-bound fp32 work.  The DotProduct core's BACKWARD runs on the matrix pipe since round 4 (interact_bodies.h: S T with
// operands straight from memory, no LDS: cfg 2 0.2786 -> 0.2771 ms, cfg 5 +1.3 %); the forward keeps the pairwise dots as 16 FMAs on
// LDS-resident rows (its matrix-pipe form, T T^T, is built and tested — -DTRI_MFMA_FWD=1 — and left off for the logits' accuracy margin).
#include "common.h"
#include "interact_bodies.h"

// ---------------------------------------------------------------------------------------------------
// DotProduct core (modules.py:366-383).  One wavefront per sample, 4 samples per workgroup.
// (T[b] staged with 20-float rows for ds_read_b128 in the forward; the backward's LDS arrays belong to its -DTRI_MFMA=0 form and vanish from the default build.)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dot_tri_fwd_kernel(const nasrec_dot_tri_desc_t d) {
  __shared__ __attribute__((aligned(16))) float Ts[4][TRI_MAXK1 * TRI_LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= d.B) return;  // whole wave exits together; no block-level barrier below
  dot_tri_fwd_sample(d, b, lane, Ts[wave]);
}

// dT[b,i,:] = sum_{j<i} dO[p(i,j)] T[j] + sum_{j>i} dO[p(j,i)] T[j]
__global__ __launch_bounds__(256) void dot_tri_bwd_kernel(const nasrec_dot_tri_desc_t d) {
  __shared__ __attribute__((aligned(16))) float Ts[4][TRI_MAXK1 * TRI_LD];
  __shared__ float Ds[4][TRI_MAXK1 * (TRI_MAXK1 - 1) / 2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= d.B) return;
  dot_tri_bwd_sample(d, b, lane, Ts[wave], Ds[wave]);
}

int launch_dot_tri(hipStream_t st, const nasrec_dot_tri_desc_t* d) {
  if (d->k1 < 2 || d->k1 > TRI_MAXK1) return nasrec_set_error(-2, "dot_tri: k1=%d out of range [2,%d]", d->k1, TRI_MAXK1);
  if (d->B == 0) return 0;
  dim3 grid((d->B + 3) / 4);
  if (d->kind == NASREC_OP_DOT_TRI_FWD)
    hipLaunchKernelGGL(dot_tri_fwd_kernel, grid, dim3(256), 0, st, *d);
  else
    hipLaunchKernelGGL(dot_tri_bwd_kernel, grid, dim3(256), 0, st, *d);
  return nasrec_check_launch("dot_tri");
}

// ---------------------------------------------------------------------------------------------------
// FactorizationMachine3D core (modules.py:736-738): thread = (b, e); 16 lanes read one 64-byte token row.
// ---------------------------------------------------------------------------------------------------
// one wavefront per sample: lane = (token group g = lane>>4, e = lane&15); 4 token rows (256 B) per load instruction,
// then the 4 groups are folded with two xor-shuffles.
__global__ __launch_bounds__(256) void fm_fwd_kernel(const nasrec_fm_desc_t d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= d.B) return;
  fm_fwd_sample(d, b, lane);
}

__global__ __launch_bounds__(256) void fm_bwd_kernel(const nasrec_fm_desc_t d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= d.B) return;
  fm_bwd_sample(d, b, lane);
}

int launch_fm(hipStream_t st, const nasrec_fm_desc_t* d) {
  if (d->B == 0) return 0;
  dim3 grid((unsigned)((d->B + 3) / 4));
  if (d->kind == NASREC_OP_FM_FWD)
    hipLaunchKernelGGL(fm_fwd_kernel, grid, dim3(256), 0, st, *d);
  else
    hipLaunchKernelGGL(fm_bwd_kernel, grid, dim3(256), 0, st, *d);
  return nasrec_check_launch("fm");
}

// ---------------------------------------------------------------------------------------------------
// Segmented copy / gradient fan-in
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_segs_kernel(const nasrec_copy_segs_desc_t d, int W) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)d.B * W) return;
  copy_segs_element(d, (int)(t / W), (int)(t % W));
}

int launch_copy_segs(hipStream_t st, const nasrec_copy_segs_desc_t* d) {
  int W = 0;
  for (int q = 0; q < d->nseg; ++q) W = max(W, d->off[q] + d->width[q]);
  long threads = (long)d->B * W;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(copy_segs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d, W);
  return nasrec_check_launch("copy_segs");
}

// ---------------------------------------------------------------------------------------------------
// SigmoidGating backward, elementwise part (modules.py:578-582)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_bwd_kernel(const nasrec_gate_bwd_desc_t d) {
  gate_bwd_element(d, (long)blockIdx.x * 256 + threadIdx.x);
}

int launch_gate_bwd(hipStream_t st, const nasrec_gate_bwd_desc_t* d) {
  long threads = (long)d->B * d->D;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d);
  return nasrec_check_launch("gate_bwd");
}

// ---------------------------------------------------------------------------------------------------
// Bias gradients: out[r] = sum_k P(r,k)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowsum_rc_kernel(const nasrec_rowsum_desc_t d) {
  __shared__ float red[16][17];
  const int rl = threadIdx.x & 15, kq = threadIdx.x >> 4;
  const int r = blockIdx.x * 16 + rl;
  float s = 0.f;
  if (r < d.R && r < d.rvalid) {
    for (int k = kq; k < d.K; k += 16) {
      long o = (long)k * d.ld + r;
      float v = d.p[o];
      if (d.aux && !(d.aux[o] > 0.f)) v = 0.f;
      s += v;
    }
  }
  red[kq][rl] = s;
  __syncthreads();
  if (kq == 0 && r < d.R) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][rl];
    d.out[r] = tot;
  }
}

__global__ __launch_bounds__(256) void rowsum_tokk_kernel(const nasrec_rowsum_desc_t d) {
  __shared__ float red[256];
  const int r = blockIdx.x;
  float s = 0.f;
  if (r < d.rvalid) {
    for (int k = threadIdx.x; k < d.K; k += 256) {
      long o = (long)(k >> 4) * d.ld + (k & 15) + (long)r * 16;
      float v = d.p[o];
      if (d.aux && !(d.aux[o] > 0.f)) v = 0.f;
      s += v;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) d.out[r] = red[0];
}

int launch_rowsum(hipStream_t st, const nasrec_rowsum_desc_t* d) {
  if (d->R == 0) return 0;
  if (d->mode == NASREC_AM_RC)
    hipLaunchKernelGGL(rowsum_rc_kernel, dim3((d->R + 15) / 16), dim3(256), 0, st, *d);
  else if (d->mode == NASREC_AM_TOKK)
    hipLaunchKernelGGL(rowsum_tokk_kernel, dim3(d->R), dim3(256), 0, st, *d);
  else
    return nasrec_set_error(-2, "rowsum: unsupported mode %d", d->mode);
  return nasrec_check_launch("rowsum");
}

// ---------------------------------------------------------------------------------------------------
// out[c] = sum_r in[r*ld + c] in fixed order, scattered to destination tensors by column range
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_rows_kernel(const nasrec_reduce_rows_desc_t d) {
  __shared__ float red[16 * 17];
  reduce_rows_block(d, blockIdx.x, threadIdx.x, red);
}

int launch_reduce_rows(hipStream_t st, const nasrec_reduce_rows_desc_t* d) {
  if (d->C == 0) return 0;
  if (d->ndst < 1 || d->ndst > NASREC_REDUCE_MAX_DST) return nasrec_set_error(-2, "reduce_rows: ndst=%d", d->ndst);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((d->C + 15) / 16), dim3(256), 0, st, *d);
  return nasrec_check_launch("reduce_rows");
}

// ---------------------------------------------------------------------------------------------------
// small elementwise helpers
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_kernel(const nasrec_scale_desc_t d) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long)gridDim.x * 256) d.y[i] = d.x[i] * d.s;
}

int launch_scale(hipStream_t st, const nasrec_scale_desc_t* d) {
  if (d->n == 0) return 0;
  long blocks = (d->n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *d);
  return nasrec_check_launch("scale");
}

// dz(r,i) = dy(r,i) * act'(z(r,i)) * [i < dims]   for r-major dense views (KC) or token views (TOKR)
__global__ __launch_bounds__(256) void act_bwd_kernel(const nasrec_act_bwd_desc_t d) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)d.R * d.D) return;
  int r, i;
  long ody, oz, odz;
  if (d.mode == NASREC_AM_KC) {
    r = (int)(t / d.D);
    i = (int)(t % d.D);
    ody = (long)r * d.ld_dy + i;
    oz = (long)r * d.ld_z + i;
    odz = (long)r * d.ld_dz + i;
  } else {  // TOKR: r = (b,e), i = token; walk e fastest
    const int bb = (int)(t / ((long)d.D * 16));
    const int rem = (int)(t % ((long)d.D * 16));
    i = rem >> 4;
    const int e = rem & 15;
    ody = (long)bb * d.ld_dy + i * 16 + e;
    oz = (long)bb * d.ld_z + i * 16 + e;
    odz = (long)bb * d.ld_dz + i * 16 + e;
  }
  float g = d.dy[ody];
  const float z = d.z[oz];
  if (d.act == NASREC_ACT_RELU) {
    g = z > 0.f ? g : 0.f;
  } else if (d.act == NASREC_ACT_SILU) {
    const float s = 1.f / (1.f + __expf(-z));
    g *= s * (1.f + z * (1.f - s));
  } else if (d.act == NASREC_ACT_SIGMOID) {
    const float s = 1.f / (1.f + __expf(-z));
    g *= s * (1.f - s);
  }
  if (d.dims_in_use >= 0 && i >= d.dims_in_use) g = 0.f;
  d.dz[odz] = g;
}

int launch_act_bwd(hipStream_t st, const nasrec_act_bwd_desc_t* d) {
  long threads = (long)d->R * d->D;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d);
  return nasrec_check_launch("act_bwd");
}
