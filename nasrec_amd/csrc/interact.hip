// Feature-interaction and glue kernels: DotProduct triangle, FactorizationMachine, SigmoidGating backward,
// segmented copies, bias-gradient row sums, fixed-order row reductions.
// All are HBM/latency-bound integer-indexed fp32 work: no MFMA here by design (E = 16 fits one lane's
// registers, so a pairwise dot is 16 FMAs on LDS-resident rows — cheaper than a cross-lane reduction).
#include "common.h"
#include "interact_bodies.h"

// ---------------------------------------------------------------------------------------------------
// DotProduct core (modules.py:366-383).  One wavefront per sample, 4 samples per workgroup.
// T[b] (k1 x 16, k1 <= 46) is staged in LDS with 20-float rows (bank spread for ds_read_b128).
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
  const int k1 = d.k1;
  const int P = k1 * (k1 - 1) / 2;
  const float* Tb = d.T + (long)b * k1 * 16;
  const float* dob = d.dout + (long)b * d.ld_out;
  float* ts = Ts[wave];
  float* ds = Ds[wave];
  for (int q = lane; q < k1 * 16; q += 64) ts[(q >> 4) * TRI_LD + (q & 15)] = Tb[q];
  for (int q = lane; q < P; q += 64) ds[q] = dob[q];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  float* dTb = d.dT + (long)b * k1 * 16;
  for (int item = lane; item < k1 * 4; item += 64) {
    const int i = item >> 2, q = item & 3;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rowbase = i * (i - 1) / 2;
    for (int j = 0; j < i; ++j) {
      const float w = ds[rowbase + j];
      f32x4 t = *reinterpret_cast<const f32x4*>(ts + j * TRI_LD + 4 * q);
      acc += w * t;
    }
    for (int j = i + 1; j < k1; ++j) {
      const float w = ds[j * (j - 1) / 2 + i];
      f32x4 t = *reinterpret_cast<const f32x4*>(ts + j * TRI_LD + 4 * q);
      acc += w * t;
    }
    *reinterpret_cast<f32x4*>(dTb + i * 16 + 4 * q) = acc;
  }
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
  const int g = lane >> 4, e = lane & 15;
  const float* x = d.x + (long)b * d.ldx + e;
  float* dx = d.dx + (long)b * d.ldx + e;
  float s = 0.f;
  for (int n = g; n < d.N; n += 4) s += x[n * 16];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  const float g2 = 2.f * d.dix[(long)b * d.ld_ix + e];
  for (int n = g; n < d.N; n += 4) {
    float r = g2 * (s - x[n * 16]);
    dx[n * 16] = d.accumulate ? dx[n * 16] + r : r;
  }
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
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
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
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rq = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < d.C) {
    // four independent chains: the loads of a cold [R, C] slab pipeline instead of queueing behind one accumulator
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rq;
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
  red[rq][cl] = s;
  __syncthreads();
  if (rq == 0 && c < d.C) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][cl];
    for (int q = 0; q < d.ndst; ++q) {
      const int cc = c - d.dst_off[q];
      if (cc >= 0 && cc < d.dst_len[q]) {
        if (d.dst[q]) d.dst[q][cc] = tot;
        break;
      }
    }
  }
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
