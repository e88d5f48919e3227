// Fused Transformer body (modules.py:664-686): nn.MultiheadAttention(embed 16, 8 heads x head_dim 2,
// batch_first, no dropout/mask) + residual + LayerNorm(16) + Linear(16,16) + ReLU + Linear(16,16) + residual +
// LayerNorm(16) (+ supernet token prefix mask).
//
// Mapping: one wavefront = one sample, lane = token (N <= 64).  A token row is 16 fp32 = 16 VGPRs, so the
// in/out projections, both LayerNorms and the FFN are pure in-lane math with wave-uniform weights (scalar
// loads).  Attention at head_dim 2 has nothing for MFMA to chew on (K = 2): K/V rows are parked in LDS and
// every lane walks the keys with LDS broadcast reads (2 FMA + 1 exp per key and head).
// The backward kernel recomputes the forward from x (nothing but x is saved), runs the flash-style
// two-phase attention backward (lane = query for dq, lane = key for dk/dv), and reduces the 1696 parameter
// gradients of the node over the sample's tokens through LDS outer-product stages; per-sample partials are
// summed across the batch in fixed order by NASREC_OP_REDUCE_ROWS (deterministic).
#include "common.h"

#define MHA_N 64
#define MHA_SCALE 0.70710678118654752440f  // 1/sqrt(head_dim = 2)

// parameter offsets inside the 1696-float gradient record (order of nasrec_mha_desc_t::params)
#define OFF_WIN 0
#define OFF_BIN 768
#define OFF_WOUT 816
#define OFF_BOUT 1072
#define OFF_L1W 1088
#define OFF_L1B 1104
#define OFF_W1 1120
#define OFF_C1 1376
#define OFF_W2 1392
#define OFF_C2 1648
#define OFF_L2W 1664
#define OFF_L2B 1680

// layout of the per-token forward state kept for the backward (NASREC_MHA_SAVED floats)
#define SV_Q 0      // scaled query
#define SV_K 16
#define SV_V 32
#define SV_O 48     // attention output (before the out-projection)
#define SV_H1 64    // LayerNorm-1 output
#define SV_XH1 80   // LayerNorm-1 x-hat
#define SV_F1 96    // FFN hidden (post-ReLU)
#define SV_XH2 112  // LayerNorm-2 x-hat
#define SV_M 128    // per-head softmax max (8) and 1/sum (8)
#define SV_RSTD 144 // 1/std of both LayerNorms

// All 1696 parameters of the node are staged once per workgroup into LDS (6.8 KB) and read back with
// wave-uniform (broadcast) ds_reads: keeping them in SGPRs instead blows the scalar register file.
static __device__ const int kParamOff[12] = {OFF_WIN, OFF_BIN, OFF_WOUT, OFF_BOUT, OFF_L1W, OFF_L1B,
                                             OFF_W1,  OFF_C1,  OFF_W2,   OFF_C2,   OFF_L2W, OFF_L2B};
static __device__ const int kParamLen[12] = {768, 48, 256, 16, 16, 16, 256, 16, 256, 16, 16, 16};

__device__ __forceinline__ void stage_params(const nasrec_mha_desc_t& d, float* Wsh, int lane) {
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const float* src = d.params[q];
    for (int i = lane; i < kParamLen[q]; i += 64) Wsh[kParamOff[q] + i] = src[i];
  }
}

// y = W x + b for one token row held in registers.  The output loop is deliberately NOT unrolled and the
// result goes through the lane's private LDS scratch row: with full unrolling hipcc hoists all 256 weight loads
// of a matvec (and of its neighbours) and spills hundreds of VGPRs.
#define SCR_LD 20
__device__ __forceinline__ void matvec16(const float* W, const float* b, const float* x, float* y, float* scr) {
#pragma unroll 1
  for (int o = 0; o < 16; ++o) {
    const f32x4* wr = reinterpret_cast<const f32x4*>(W + o * 16);
    float s = b[o];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f32x4 w = wr[v];
      s = fmaf(w[0], x[4 * v], s);
      s = fmaf(w[1], x[4 * v + 1], s);
      s = fmaf(w[2], x[4 * v + 2], s);
      s = fmaf(w[3], x[4 * v + 3], s);
    }
    scr[o] = s;
  }
  const f32x4* sr = reinterpret_cast<const f32x4*>(scr);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    f32x4 t = sr[v];
    y[4 * v] = t[0];
    y[4 * v + 1] = t[1];
    y[4 * v + 2] = t[2];
    y[4 * v + 3] = t[3];
  }
}

// y[i] += sum_o W[o*16+i] * g[o]
__device__ __forceinline__ void matvec16_t_acc(const float* W, const float* g, float* y, float* scr) {
  f32x4* sw = reinterpret_cast<f32x4*>(scr);
#pragma unroll
  for (int v = 0; v < 4; ++v) sw[v] = (f32x4){g[4 * v], g[4 * v + 1], g[4 * v + 2], g[4 * v + 3]};
#pragma unroll 1
  for (int o = 0; o < 16; ++o) {
    const f32x4* wr = reinterpret_cast<const f32x4*>(W + o * 16);
    const float go = scr[o];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f32x4 w = wr[v];
      y[4 * v] = fmaf(w[0], go, y[4 * v]);
      y[4 * v + 1] = fmaf(w[1], go, y[4 * v + 1]);
      y[4 * v + 2] = fmaf(w[2], go, y[4 * v + 2]);
      y[4 * v + 3] = fmaf(w[3], go, y[4 * v + 3]);
    }
  }
}

__device__ __forceinline__ void ln16_fwd(const float* r, const float* __restrict__ w, const float* __restrict__ b, float* y,
                                         float* xhat, float& rstd) {
  float mu = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) mu += r[e];
  mu *= (1.f / 16.f);
  float var = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float c = r[e] - mu;
    var = fmaf(c, c, var);
  }
  var *= (1.f / 16.f);
  rstd = 1.f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    xhat[e] = (r[e] - mu) * rstd;
    y[e] = fmaf(xhat[e], w[e], b[e]);
  }
}

// dx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat))
__device__ __forceinline__ void ln16_bwd(const float* g, const float* __restrict__ w, const float* xhat, float rstd, float* dx) {
  float c1 = 0.f, c2 = 0.f;
  float gw[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    gw[e] = g[e] * w[e];
    c1 += gw[e];
    c2 = fmaf(gw[e], xhat[e], c2);
  }
  c1 *= (1.f / 16.f);
  c2 *= (1.f / 16.f);
#pragma unroll
  for (int e = 0; e < 16; ++e) dx[e] = rstd * (gw[e] - c1 - xhat[e] * c2);
}

__device__ __forceinline__ void load_row16(const float* p, float* x) {
  const f32x4* s = reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    f32x4 t = s[v];
    x[4 * v] = t[0];
    x[4 * v + 1] = t[1];
    x[4 * v + 2] = t[2];
    x[4 * v + 3] = t[3];
  }
}

__device__ __forceinline__ void store_row16(float* p, const float* x) {
  f32x4* s = reinterpret_cast<f32x4*>(p);
#pragma unroll
  for (int v = 0; v < 4; ++v) s[v] = (f32x4){x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]};
}

// attention forward for the lane's query against all N keys (two passes: max, then exp/accumulate)
__device__ __forceinline__ void attn_fwd_lane(const float* qs, const float* Ks, const float* Vs, int N, float* o, float* m,
                                              float* linv) {
#pragma unroll
  for (int h = 0; h < 8; ++h) m[h] = -INFINITY;
  for (int j = 0; j < N; ++j) {
    float kj[16];
    load_row16(Ks + j * 16, kj);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      float s = fmaf(qs[2 * h], kj[2 * h], qs[2 * h + 1] * kj[2 * h + 1]);
      m[h] = fmaxf(m[h], s);
    }
  }
  float l[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) l[h] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  for (int j = 0; j < N; ++j) {
    float kj[16], vj[16];
    load_row16(Ks + j * 16, kj);
    load_row16(Vs + j * 16, vj);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      float s = fmaf(qs[2 * h], kj[2 * h], qs[2 * h + 1] * kj[2 * h + 1]);
      float p = __expf(s - m[h]);
      l[h] += p;
      o[2 * h] = fmaf(p, vj[2 * h], o[2 * h]);
      o[2 * h + 1] = fmaf(p, vj[2 * h + 1], o[2 * h + 1]);
    }
  }
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    linv[h] = 1.f / l[h];
    o[2 * h] *= linv[h];
    o[2 * h + 1] *= linv[h];
  }
}

__global__ __launch_bounds__(64) void mha_fwd_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float Ks[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Vs[MHA_N * 16];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = d.N;
  const bool active = lane < N;
  __shared__ __attribute__((aligned(16))) float Wsh[NASREC_MHA_PARAMS];
  __shared__ __attribute__((aligned(16))) float Scr[MHA_N * SCR_LD];
  float* scr = Scr + lane * SCR_LD;
  stage_params(d, Wsh, lane);
  const float* Win = Wsh + OFF_WIN;
  const float* bin = Wsh + OFF_BIN;
  float x[16];
  if (active) {
    load_row16(d.x + (long)b * d.ldx + lane * 16, x);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
  }
  __syncthreads();
  float q[16], k[16], v[16];
  matvec16(Win, bin, x, q, scr);
  matvec16(Win + 256, bin + 16, x, k, scr);
  matvec16(Win + 512, bin + 32, x, v, scr);
  store_row16(Ks + lane * 16, k);
  store_row16(Vs + lane * 16, v);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) q[e] *= MHA_SCALE;
  float o[16], m[8], linv[8];
  attn_fwd_lane(q, Ks, Vs, N, o, m, linv);
  float a[16];
  matvec16(Wsh + OFF_WOUT, Wsh + OFF_BOUT, o, a, scr);
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] += x[e];
  float h1[16], xh1[16], rstd1;
  ln16_fwd(a, Wsh + OFF_L1W, Wsh + OFF_L1B, h1, xh1, rstd1);
  float f1[16], f2[16];
  matvec16(Wsh + OFF_W1, Wsh + OFF_C1, h1, f1, scr);
#pragma unroll
  for (int e = 0; e < 16; ++e) f1[e] = fmaxf(f1[e], 0.f);
  matvec16(Wsh + OFF_W2, Wsh + OFF_C2, f1, f2, scr);
#pragma unroll
  for (int e = 0; e < 16; ++e) f2[e] += h1[e];
  float out[16], xh2[16], rstd2;
  ln16_fwd(f2, Wsh + OFF_L2W, Wsh + OFF_L2B, out, xh2, rstd2);
  if (d.dims_in_use >= 0 && lane >= d.dims_in_use) {
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = 0.f;
  }
  if (active) store_row16(d.out + (long)b * d.ldo + lane * 16, out);
  if (d.saved != nullptr && active) {  // training: keep what the backward needs instead of recomputing it there
    float* sv = d.saved + ((long)b * N + lane) * NASREC_MHA_SAVED;
    store_row16(sv + SV_Q, q);
    store_row16(sv + SV_K, k);
    store_row16(sv + SV_V, v);
    store_row16(sv + SV_O, o);
    store_row16(sv + SV_H1, h1);
    store_row16(sv + SV_XH1, xh1);
    store_row16(sv + SV_F1, f1);
    store_row16(sv + SV_XH2, xh2);
    float ml[16];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      ml[h] = m[h];
      ml[8 + h] = linv[h];
    }
    store_row16(sv + SV_M, ml);
    *reinterpret_cast<f32x4*>(sv + SV_RSTD) = (f32x4){rstd1, rstd2, 0.f, 0.f};
  }
}

// LDS outer-product stage: entry(o,i) = sum_tok L[tok][o] * R[tok][i], i in [0,16]; R[tok][16] == 1 gives the
// column sums of L.  diag_only: only entries (e,e) and (e,16).
#define ST_LD 17
__device__ __forceinline__ void stage_rows(float* Ls, float* Rs, int lane, const float* l, const float* r) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    Ls[lane * ST_LD + e] = l[e];
    Rs[lane * ST_LD + e] = r[e];
  }
  Rs[lane * ST_LD + 16] = 1.f;
}

__device__ __forceinline__ void stage_reduce(const float* Ls, const float* Rs, int lane, int N, float* outW, float* outB) {
  // 16 x 17 entries over 64 lanes
  for (int ent = lane; ent < 16 * 17; ent += 64) {
    const int o = ent / 17, i = ent % 17;
    float s = 0.f;
    for (int t = 0; t < N; ++t) s = fmaf(Ls[t * ST_LD + o], Rs[t * ST_LD + i], s);
    if (i < 16)
      outW[o * 16 + i] = s;
    else
      outB[o] = s;
  }
}

__device__ __forceinline__ void stage_reduce_diag(const float* Ls, const float* Rs, int lane, int N, float* outW, float* outB) {
  if (lane < 32) {
    const int o = lane & 15, i = lane < 16 ? o : 16;
    float s = 0.f;
    for (int t = 0; t < N; ++t) s = fmaf(Ls[t * ST_LD + o], Rs[t * ST_LD + i], s);
    if (lane < 16)
      outW[o] = s;
    else
      outB[o] = s;
  }
}

__global__ __launch_bounds__(64) void mha_bwd_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float Ks[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Vs[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Qs[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float DOs[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Ms[MHA_N * 8];
  __shared__ __attribute__((aligned(16))) float Li[MHA_N * 8];
  __shared__ __attribute__((aligned(16))) float Dl[MHA_N * 8];
  __shared__ float Ls[MHA_N * ST_LD];
  __shared__ float Rs[MHA_N * ST_LD];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = d.N;
  const bool active = lane < N;
  __shared__ __attribute__((aligned(16))) float Wsh[NASREC_MHA_PARAMS];
  __shared__ __attribute__((aligned(16))) float Scr[MHA_N * SCR_LD];
  float* scr = Scr + lane * SCR_LD;
  stage_params(d, Wsh, lane);
  const float* Win = Wsh + OFF_WIN;
  const float* bin = Wsh + OFF_BIN;
  const float* Wout = Wsh + OFF_WOUT;
  const float* l1w = Wsh + OFF_L1W;
  const float* W1 = Wsh + OFF_W1;
  const float* W2 = Wsh + OFF_W2;
  const float* l2w = Wsh + OFF_L2W;
  float* gp = d.dparams_partial + (long)b * NASREC_MHA_PARAMS;

  // ---------------- forward state: saved by the forward launch, or recomputed from x ----------------
  float x[16];
  if (active) {
    load_row16(d.x + (long)b * d.ldx + lane * 16, x);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
  }
  __syncthreads();
  float qs[16], k[16], v[16], o[16], m[8], linv[8], h1[16], xhat1[16], rstd1, f1[16], xhat2[16], rstd2;
  if (d.saved != nullptr) {
    if (active) {
      const float* sv = d.saved + ((long)b * N + lane) * NASREC_MHA_SAVED;
      load_row16(sv + SV_Q, qs);
      load_row16(sv + SV_K, k);
      load_row16(sv + SV_V, v);
      load_row16(sv + SV_O, o);
      load_row16(sv + SV_H1, h1);
      load_row16(sv + SV_XH1, xhat1);
      load_row16(sv + SV_F1, f1);
      load_row16(sv + SV_XH2, xhat2);
      float ml[16];
      load_row16(sv + SV_M, ml);
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        m[h] = ml[h];
        linv[h] = ml[8 + h];
      }
      const f32x4 rs = *reinterpret_cast<const f32x4*>(sv + SV_RSTD);
      rstd1 = rs[0];
      rstd2 = rs[1];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) qs[e] = k[e] = v[e] = o[e] = h1[e] = xhat1[e] = f1[e] = xhat2[e] = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        m[h] = 0.f;
        linv[h] = 1.f;
      }
      rstd1 = rstd2 = 1.f;
    }
    store_row16(Ks + lane * 16, k);
    store_row16(Vs + lane * 16, v);
    store_row16(Qs + lane * 16, qs);
    __syncthreads();
  } else {
    matvec16(Win, bin, x, qs, scr);
    matvec16(Win + 256, bin + 16, x, k, scr);
    matvec16(Win + 512, bin + 32, x, v, scr);
#pragma unroll
    for (int e = 0; e < 16; ++e) qs[e] *= MHA_SCALE;
    store_row16(Ks + lane * 16, k);
    store_row16(Vs + lane * 16, v);
    store_row16(Qs + lane * 16, qs);
    __syncthreads();
    attn_fwd_lane(qs, Ks, Vs, N, o, m, linv);
    float r1[16];
    matvec16(Wout, Wsh + OFF_BOUT, o, r1, scr);
#pragma unroll
    for (int e = 0; e < 16; ++e) r1[e] += x[e];
    ln16_fwd(r1, l1w, Wsh + OFF_L1B, h1, xhat1, rstd1);
    float r2[16], y2[16];
    matvec16(W1, Wsh + OFF_C1, h1, f1, scr);
#pragma unroll
    for (int e = 0; e < 16; ++e) f1[e] = fmaxf(f1[e], 0.f);
    matvec16(W2, Wsh + OFF_C2, f1, r2, scr);
#pragma unroll
    for (int e = 0; e < 16; ++e) r2[e] += h1[e];
    ln16_fwd(r2, l2w, Wsh + OFF_L2B, y2, xhat2, rstd2);
  }

  // ---------------- backward ----------------
  float dout[16];
  const bool has_grad = active && !(d.dims_in_use >= 0 && lane >= d.dims_in_use);
  if (has_grad) {
    load_row16(d.dout + (long)b * d.ldo + lane * 16, dout);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) dout[e] = 0.f;
  }
  // LN2 parameter grads: dl2w = sum dout*xhat2 (diagonal), dl2b = sum dout
  stage_rows(Ls, Rs, lane, dout, xhat2);
  __syncthreads();
  stage_reduce_diag(Ls, Rs, lane, N, gp + OFF_L2W, gp + OFF_L2B);
  __syncthreads();
  float dr2[16];
  ln16_bwd(dout, l2w, xhat2, rstd2, dr2);
  // FFN second layer: f2 = W2 f1 + c2
  stage_rows(Ls, Rs, lane, dr2, f1);
  __syncthreads();
  stage_reduce(Ls, Rs, lane, N, gp + OFF_W2, gp + OFF_C2);
  __syncthreads();
  float df1[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) df1[e] = 0.f;
  matvec16_t_acc(W2, dr2, df1, scr);
#pragma unroll
  for (int e = 0; e < 16; ++e) df1[e] = f1[e] > 0.f ? df1[e] : 0.f;
  // FFN first layer
  stage_rows(Ls, Rs, lane, df1, h1);
  __syncthreads();
  stage_reduce(Ls, Rs, lane, N, gp + OFF_W1, gp + OFF_C1);
  __syncthreads();
  float dh1[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dh1[e] = dr2[e];
  matvec16_t_acc(W1, df1, dh1, scr);
  // LN1
  stage_rows(Ls, Rs, lane, dh1, xhat1);
  __syncthreads();
  stage_reduce_diag(Ls, Rs, lane, N, gp + OFF_L1W, gp + OFF_L1B);
  __syncthreads();
  float dr1[16];
  ln16_bwd(dh1, l1w, xhat1, rstd1, dr1);
  // out-projection: a = Wout o + bout
  stage_rows(Ls, Rs, lane, dr1, o);
  __syncthreads();
  stage_reduce(Ls, Rs, lane, N, gp + OFF_WOUT, gp + OFF_BOUT);
  __syncthreads();
  float dO[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dO[e] = 0.f;
  matvec16_t_acc(Wout, dr1, dO, scr);
  // attention backward
  float delta[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) delta[h] = fmaf(dO[2 * h], o[2 * h], dO[2 * h + 1] * o[2 * h + 1]);
  store_row16(DOs + lane * 16, dO);
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    Ms[lane * 8 + h] = m[h];
    Li[lane * 8 + h] = linv[h];
    Dl[lane * 8 + h] = delta[h];
  }
  __syncthreads();
  // phase A: lane = query -> dq
  float dq[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dq[e] = 0.f;
  for (int j = 0; j < N; ++j) {
    float kj[16], vj[16];
    load_row16(Ks + j * 16, kj);
    load_row16(Vs + j * 16, vj);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      float s = fmaf(qs[2 * h], kj[2 * h], qs[2 * h + 1] * kj[2 * h + 1]);
      float p = __expf(s - m[h]) * linv[h];
      float dp = fmaf(dO[2 * h], vj[2 * h], dO[2 * h + 1] * vj[2 * h + 1]);
      float ds = p * (dp - delta[h]);
      dq[2 * h] = fmaf(ds, kj[2 * h], dq[2 * h]);
      dq[2 * h + 1] = fmaf(ds, kj[2 * h + 1], dq[2 * h + 1]);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) dq[e] *= MHA_SCALE;
  // phase B: lane = key -> dk, dv
  float dk[16], dv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    dk[e] = 0.f;
    dv[e] = 0.f;
  }
  for (int i = 0; i < N; ++i) {
    float qi[16], doi[16], mi[8], li[8], di[8];
    load_row16(Qs + i * 16, qi);
    load_row16(DOs + i * 16, doi);
    {
      const f32x4* pm = reinterpret_cast<const f32x4*>(Ms + i * 8);
      const f32x4* pl = reinterpret_cast<const f32x4*>(Li + i * 8);
      const f32x4* pd = reinterpret_cast<const f32x4*>(Dl + i * 8);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 a = pm[u], c = pl[u], e4 = pd[u];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          mi[4 * u + w] = a[w];
          li[4 * u + w] = c[w];
          di[4 * u + w] = e4[w];
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      float s = fmaf(qi[2 * h], k[2 * h], qi[2 * h + 1] * k[2 * h + 1]);
      float p = __expf(s - mi[h]) * li[h];
      dv[2 * h] = fmaf(p, doi[2 * h], dv[2 * h]);
      dv[2 * h + 1] = fmaf(p, doi[2 * h + 1], dv[2 * h + 1]);
      float dp = fmaf(doi[2 * h], v[2 * h], doi[2 * h + 1] * v[2 * h + 1]);
      float ds = p * (dp - di[h]);
      dk[2 * h] = fmaf(ds, qi[2 * h], dk[2 * h]);
      dk[2 * h + 1] = fmaf(ds, qi[2 * h + 1], dk[2 * h + 1]);
    }
  }
  if (!active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dk[e] = 0.f;
      dv[e] = 0.f;
      dq[e] = 0.f;
    }
  }
  // in-projection parameter grads (three 16-row stages) and dx
  float dx[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dx[e] = dr1[e];
  stage_rows(Ls, Rs, lane, dq, x);
  __syncthreads();
  stage_reduce(Ls, Rs, lane, N, gp + OFF_WIN, gp + OFF_BIN);
  __syncthreads();
  matvec16_t_acc(Win, dq, dx, scr);
  stage_rows(Ls, Rs, lane, dk, x);
  __syncthreads();
  stage_reduce(Ls, Rs, lane, N, gp + OFF_WIN + 256, gp + OFF_BIN + 16);
  __syncthreads();
  matvec16_t_acc(Win + 256, dk, dx, scr);
  stage_rows(Ls, Rs, lane, dv, x);
  __syncthreads();
  stage_reduce(Ls, Rs, lane, N, gp + OFF_WIN + 512, gp + OFF_BIN + 32);
  matvec16_t_acc(Win + 512, dv, dx, scr);
  if (active) store_row16(d.dx + (long)b * d.ldx + lane * 16, dx);
}

int launch_mha(hipStream_t st, const nasrec_mha_desc_t* d) {
  if (d->N < 1 || d->N > MHA_N) return nasrec_set_error(-2, "mha: N=%d out of range [1,%d]", d->N, MHA_N);
  if (d->B == 0) return 0;
  if (d->kind == NASREC_OP_MHA_FWD)
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(d->B), dim3(64), 0, st, *d);
  else
    hipLaunchKernelGGL(mha_bwd_kernel, dim3(d->B), dim3(64), 0, st, *d);
  return nasrec_check_launch("mha");
}
