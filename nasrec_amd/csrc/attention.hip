// Fused Transformer body (modules.py:664-686): nn.MultiheadAttention(embed 16, 8 heads x head_dim 2,
// batch_first, no dropout/mask) + residual + LayerNorm(16) + Linear(16,16) + ReLU + Linear(16,16) + residual +
// LayerNorm(16) (+ supernet token prefix mask).
//
// Mapping: one workgroup (4 wavefronts) = one sample; lane = token (N <= 64), wave w owns the 4-column slice
// [4w, 4w+4) of every 16-wide vector — i.e. heads 2w and 2w+1 of the attention, rows 4w..4w+3 of every projection.
// Full 16-vectors that a slice computation needs (the other waves' columns) go through LDS rows [token][16].
// With one sample per wavefront only 256 of the chip's 1024 SIMDs had work at batch 256; this layout gives every
// SIMD one wave and cuts the serial instruction stream per wave by ~4x.
// Attention at head_dim 2 has nothing for MFMA to chew on (K = 2): K/V slices are parked in LDS and every lane walks
// the keys (one ds_read_b128 per key for K, one for V; 2 FMA + 1 exp per key and head).
// The forward launch saves per-token state (NASREC_MHA_SAVED floats); the backward launch reads it, runs the
// flash-style two-phase attention backward (lane = query for dq, lane = key for dk/dv), and reduces the 1696
// parameter gradients of the node over the sample's tokens: weight gradients as LDS outer products (lane = one
// (row, column) entry of the wave's 4x16 slice), bias / LayerNorm gradients as wave reductions; per-sample
// partials are summed across the batch in fixed order by NASREC_OP_REDUCE_ROWS (deterministic).
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

__device__ __forceinline__ void stage_params(const nasrec_mha_desc_t& d, float* Wsh, int tid) {
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const float* src = d.params[q];
    for (int i = tid; i < kParamLen[q]; i += 256) Wsh[kParamOff[q] + i] = src[i];
  }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ void ld_row(const float* p, float* x) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    f32x4 t = ld4(p + 4 * v);
    x[4 * v] = t[0];
    x[4 * v + 1] = t[1];
    x[4 * v + 2] = t[2];
    x[4 * v + 3] = t[3];
  }
}

// y[r] = b[c0+r] + sum_i W[(c0+r)*16 + i] * x[i], r = 0..3  (W, b in LDS; x = full 16-vector in registers)
__device__ __forceinline__ f32x4 mv_slice(const float* W, const float* b, int c0, const float* x) {
  f32x4 y;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = b[c0 + r];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f32x4 w = ld4(W + (c0 + r) * 16 + 4 * v);
      s = fmaf(w[0], x[4 * v], s);
      s = fmaf(w[1], x[4 * v + 1], s);
      s = fmaf(w[2], x[4 * v + 2], s);
      s = fmaf(w[3], x[4 * v + 3], s);
    }
    y[r] = s;
  }
  return y;
}

// y[ii] += sum_o W[o*16 + c0+ii] * g[o], ii = 0..3  (transposed product restricted to the wave's columns)
__device__ __forceinline__ void mvt_slice_acc(const float* W, int c0, const float* g, f32x4& y) {
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    f32x4 w = ld4(W + o * 16 + c0);
    y[0] = fmaf(w[0], g[o], y[0]);
    y[1] = fmaf(w[1], g[o], y[1]);
    y[2] = fmaf(w[2], g[o], y[2]);
    y[3] = fmaf(w[3], g[o], y[3]);
  }
}

__device__ __forceinline__ float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// mean and 1/std of a 16-vector whose four slices live in the four waves (lane = token); two LDS exchanges
__device__ __forceinline__ void ln_stats(f32x4 v, float* redA, float* redB, int w, int lane, float& mu, float& rstd) {
  redA[w * 64 + lane] = sum4(v);
  __syncthreads();
  mu = ((redA[lane] + redA[64 + lane]) + (redA[128 + lane] + redA[192 + lane])) * (1.f / 16.f);
  f32x4 c = v - mu;
  redB[w * 64 + lane] = sum4(c * c);
  __syncthreads();
  const float var = ((redB[lane] + redB[64 + lane]) + (redB[128 + lane] + redB[192 + lane])) * (1.f / 16.f);
  rstd = 1.f / sqrtf(var + 1e-5f);
}

__global__ __launch_bounds__(256) void mha_fwd_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float Wsh[NASREC_MHA_PARAMS];
  __shared__ __attribute__((aligned(16))) float Ks[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Vs[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Ob[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Hb[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Fb[MHA_N * 16];
  __shared__ float red[4][256];
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c0 = 4 * w;
  const int N = d.N;
  const bool active = lane < N;
  stage_params(d, Wsh, tid);
  float x[16];
  if (active) {
    ld_row(d.x + (long)b * d.ldx + lane * 16, x);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
  }
  __syncthreads();
  // in-projection, the wave's 4 columns of q, k, v
  f32x4 q4 = mv_slice(Wsh + OFF_WIN, Wsh + OFF_BIN, c0, x) * MHA_SCALE;
  const f32x4 k4 = mv_slice(Wsh + OFF_WIN + 256, Wsh + OFF_BIN + 16, c0, x);
  const f32x4 v4 = mv_slice(Wsh + OFF_WIN + 512, Wsh + OFF_BIN + 32, c0, x);
  st4(Ks + lane * 16 + c0, k4);
  st4(Vs + lane * 16 + c0, v4);
  __syncthreads();
  // attention, heads 2w (columns c0, c0+1) and 2w+1 (columns c0+2, c0+3)
  float mA = -INFINITY, mB = -INFINITY;
#pragma unroll 4
  for (int j = 0; j < N; ++j) {
    const f32x4 kj = ld4(Ks + j * 16 + c0);
    mA = fmaxf(mA, fmaf(q4[0], kj[0], q4[1] * kj[1]));
    mB = fmaxf(mB, fmaf(q4[2], kj[2], q4[3] * kj[3]));
  }
  float lA = 0.f, lB = 0.f;
  f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int j = 0; j < N; ++j) {
    const f32x4 kj = ld4(Ks + j * 16 + c0);
    const f32x4 vj = ld4(Vs + j * 16 + c0);
    const float pA = __expf(fmaf(q4[0], kj[0], q4[1] * kj[1]) - mA);
    const float pB = __expf(fmaf(q4[2], kj[2], q4[3] * kj[3]) - mB);
    lA += pA;
    lB += pB;
    o4[0] = fmaf(pA, vj[0], o4[0]);
    o4[1] = fmaf(pA, vj[1], o4[1]);
    o4[2] = fmaf(pB, vj[2], o4[2]);
    o4[3] = fmaf(pB, vj[3], o4[3]);
  }
  const float liA = 1.f / lA, liB = 1.f / lB;
  o4[0] *= liA;
  o4[1] *= liA;
  o4[2] *= liB;
  o4[3] *= liB;
  st4(Ob + lane * 16 + c0, o4);
  __syncthreads();
  // out-projection + residual + LayerNorm 1
  float row[16];
  ld_row(Ob + lane * 16, row);
  f32x4 r1 = mv_slice(Wsh + OFF_WOUT, Wsh + OFF_BOUT, c0, row);
#pragma unroll
  for (int r = 0; r < 4; ++r) r1[r] += x[c0 + r];
  float mu1, rstd1;
  ln_stats(r1, red[0], red[1], w, lane, mu1, rstd1);
  const f32x4 xh1 = (r1 - mu1) * rstd1;
  const f32x4 h1 = xh1 * ld4(Wsh + OFF_L1W + c0) + ld4(Wsh + OFF_L1B + c0);
  st4(Hb + lane * 16 + c0, h1);
  __syncthreads();
  // FFN
  ld_row(Hb + lane * 16, row);
  f32x4 f1 = mv_slice(Wsh + OFF_W1, Wsh + OFF_C1, c0, row);
#pragma unroll
  for (int r = 0; r < 4; ++r) f1[r] = fmaxf(f1[r], 0.f);
  st4(Fb + lane * 16 + c0, f1);
  __syncthreads();
  ld_row(Fb + lane * 16, row);
  const f32x4 r2 = mv_slice(Wsh + OFF_W2, Wsh + OFF_C2, c0, row) + h1;
  float mu2, rstd2;
  ln_stats(r2, red[2], red[3], w, lane, mu2, rstd2);
  const f32x4 xh2 = (r2 - mu2) * rstd2;
  f32x4 out = xh2 * ld4(Wsh + OFF_L2W + c0) + ld4(Wsh + OFF_L2B + c0);
  if (d.dims_in_use >= 0 && lane >= d.dims_in_use) out = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (active) st4(d.out + (long)b * d.ldo + lane * 16 + c0, out);
  if (d.saved != nullptr && active) {  // training: keep what the backward needs instead of recomputing it there
    float* sv = d.saved + ((long)b * N + lane) * NASREC_MHA_SAVED;
    st4(sv + SV_Q + c0, q4);
    st4(sv + SV_K + c0, k4);
    st4(sv + SV_V + c0, v4);
    st4(sv + SV_O + c0, o4);
    st4(sv + SV_H1 + c0, h1);
    st4(sv + SV_XH1 + c0, xh1);
    st4(sv + SV_F1 + c0, f1);
    st4(sv + SV_XH2 + c0, xh2);
    sv[SV_M + 2 * w] = mA;
    sv[SV_M + 2 * w + 1] = mB;
    sv[SV_M + 8 + 2 * w] = liA;
    sv[SV_M + 8 + 2 * w + 1] = liB;
    if (w == 0) st4(sv + SV_RSTD, (f32x4){rstd1, rstd2, 0.f, 0.f});
  }
}

// Weight-gradient slice of one product y = W v (W [16,16]): dW[c0+o][i] = sum_tok G[tok][c0+o] * V[tok][i], o < 4.
// lane = one of the wave's 64 entries (o = lane >> 4, i = lane & 15); G, V are LDS rows [token][16].
__device__ __forceinline__ void wgrad_slice(const float* G, const float* V, int c0, int lane, int N, float* out) {
  const int o = lane >> 4, i = lane & 15;
  float s = 0.f;
#pragma unroll 8
  for (int t = 0; t < N; ++t) s = fmaf(G[t * 16 + c0 + o], V[t * 16 + i], s);
  out[(c0 + o) * 16 + i] = s;
}

// bias-like gradient slice: out[c0+r] = sum over tokens (lanes) of g[r]
__device__ __forceinline__ void bgrad_slice(f32x4 g, int c0, int lane, float* out) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s = wave_sum(g[r]);
    if (lane == 0) out[c0 + r] = s;
  }
}

__global__ __launch_bounds__(256) void mha_bwd_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float Wsh[NASREC_MHA_PARAMS];
  __shared__ __attribute__((aligned(16))) float Bf[9][MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Mb[MHA_N * 8];
  __shared__ __attribute__((aligned(16))) float Lb[MHA_N * 8];
  __shared__ __attribute__((aligned(16))) float Db[MHA_N * 8];
  __shared__ float red[4][256];
  // LDS rows [token][16]; buffers are re-used once their previous content is dead (a barrier separates the uses)
  float* Xb = Bf[0];
  float* Qb = Bf[1];
  float* Kb = Bf[2];
  float* Vb = Bf[3];
  float* Ob = Bf[4];
  float* F1b = Bf[5];
  float* DOb = Bf[5];   // after the FFN-2 stage
  float* H1b = Bf[6];
  float* DQb = Bf[6];   // after the FFN-1 stage
  float* DR2b = Bf[7];
  float* DR1b = Bf[7];  // after the FFN-2 stage
  float* DKb = Bf[7];   // after the out-projection stage
  float* DF1b = Bf[8];
  float* DVb = Bf[8];   // after the FFN-1 stage
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c0 = 4 * w;
  const int N = d.N;
  const bool active = lane < N;
  float* gp = d.dparams_partial + (long)b * NASREC_MHA_PARAMS;
  stage_params(d, Wsh, tid);
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 x4 = zero4, q4 = zero4, k4 = zero4, v4 = zero4, o4 = zero4, h1 = zero4, xh1 = zero4, f1 = zero4, xh2 = zero4, dout = zero4;
  float mA = 0.f, mB = 0.f, liA = 1.f, liB = 1.f, rstd1 = 1.f, rstd2 = 1.f;
  if (active) {
    const float* sv = d.saved + ((long)b * N + lane) * NASREC_MHA_SAVED;
    x4 = ld4(d.x + (long)b * d.ldx + lane * 16 + c0);
    q4 = ld4(sv + SV_Q + c0);
    k4 = ld4(sv + SV_K + c0);
    v4 = ld4(sv + SV_V + c0);
    o4 = ld4(sv + SV_O + c0);
    h1 = ld4(sv + SV_H1 + c0);
    xh1 = ld4(sv + SV_XH1 + c0);
    f1 = ld4(sv + SV_F1 + c0);
    xh2 = ld4(sv + SV_XH2 + c0);
    mA = sv[SV_M + 2 * w];
    mB = sv[SV_M + 2 * w + 1];
    liA = sv[SV_M + 8 + 2 * w];
    liB = sv[SV_M + 8 + 2 * w + 1];
    rstd1 = sv[SV_RSTD];
    rstd2 = sv[SV_RSTD + 1];
    if (!(d.dims_in_use >= 0 && lane >= d.dims_in_use)) dout = ld4(d.dout + (long)b * d.ldo + lane * 16 + c0);
  }
  st4(Xb + lane * 16 + c0, x4);
  st4(Qb + lane * 16 + c0, q4);
  st4(Kb + lane * 16 + c0, k4);
  st4(Vb + lane * 16 + c0, v4);
  st4(Ob + lane * 16 + c0, o4);
  st4(F1b + lane * 16 + c0, f1);
  st4(H1b + lane * 16 + c0, h1);
  Mb[lane * 8 + 2 * w] = mA;
  Mb[lane * 8 + 2 * w + 1] = mB;
  Lb[lane * 8 + 2 * w] = liA;
  Lb[lane * 8 + 2 * w + 1] = liB;
  // ---- LayerNorm 2 ----
  bgrad_slice(dout * xh2, c0, lane, gp + OFF_L2W);
  bgrad_slice(dout, c0, lane, gp + OFF_L2B);
  __syncthreads();  // parameters and the token rows are in LDS
  f32x4 gw = dout * ld4(Wsh + OFF_L2W + c0);
  red[0][w * 64 + lane] = sum4(gw);
  red[1][w * 64 + lane] = sum4(gw * xh2);
  __syncthreads();
  float c1 = ((red[0][lane] + red[0][64 + lane]) + (red[0][128 + lane] + red[0][192 + lane])) * (1.f / 16.f);
  float c2 = ((red[1][lane] + red[1][64 + lane]) + (red[1][128 + lane] + red[1][192 + lane])) * (1.f / 16.f);
  const f32x4 dr2 = (gw - c1 - xh2 * c2) * rstd2;
  st4(DR2b + lane * 16 + c0, dr2);
  __syncthreads();
  // ---- FFN 2: f2 = W2 f1 + c2 ----
  wgrad_slice(DR2b, F1b, c0, lane, N, gp + OFF_W2);
  bgrad_slice(dr2, c0, lane, gp + OFF_C2);
  float row[16];
  ld_row(DR2b + lane * 16, row);
  f32x4 df1 = zero4;
  mvt_slice_acc(Wsh + OFF_W2, c0, row, df1);
#pragma unroll
  for (int r = 0; r < 4; ++r) df1[r] = f1[r] > 0.f ? df1[r] : 0.f;
  st4(DF1b + lane * 16 + c0, df1);
  __syncthreads();
  // ---- FFN 1: f1 = relu(W1 h1 + c1) ----
  wgrad_slice(DF1b, H1b, c0, lane, N, gp + OFF_W1);
  bgrad_slice(df1, c0, lane, gp + OFF_C1);
  ld_row(DF1b + lane * 16, row);
  f32x4 dh1 = dr2;
  mvt_slice_acc(Wsh + OFF_W1, c0, row, dh1);
  // ---- LayerNorm 1 ----
  bgrad_slice(dh1 * xh1, c0, lane, gp + OFF_L1W);
  bgrad_slice(dh1, c0, lane, gp + OFF_L1B);
  gw = dh1 * ld4(Wsh + OFF_L1W + c0);
  red[2][w * 64 + lane] = sum4(gw);
  red[3][w * 64 + lane] = sum4(gw * xh1);
  __syncthreads();  // also: every wave is done reading DR2b / F1b / H1b / DF1b
  c1 = ((red[2][lane] + red[2][64 + lane]) + (red[2][128 + lane] + red[2][192 + lane])) * (1.f / 16.f);
  c2 = ((red[3][lane] + red[3][64 + lane]) + (red[3][128 + lane] + red[3][192 + lane])) * (1.f / 16.f);
  const f32x4 dr1 = (gw - c1 - xh1 * c2) * rstd1;
  st4(DR1b + lane * 16 + c0, dr1);
  __syncthreads();
  // ---- out-projection: a = Wout o + bout ----
  wgrad_slice(DR1b, Ob, c0, lane, N, gp + OFF_WOUT);
  bgrad_slice(dr1, c0, lane, gp + OFF_BOUT);
  ld_row(DR1b + lane * 16, row);
  f32x4 dO = zero4;
  mvt_slice_acc(Wsh + OFF_WOUT, c0, row, dO);
  const float dA = fmaf(dO[0], o4[0], dO[1] * o4[1]);
  const float dB = fmaf(dO[2], o4[2], dO[3] * o4[3]);
  st4(DOb + lane * 16 + c0, dO);
  Db[lane * 8 + 2 * w] = dA;
  Db[lane * 8 + 2 * w + 1] = dB;
  __syncthreads();  // also: every wave is done reading DR1b
  // ---- attention backward, heads 2w and 2w+1 ----
  f32x4 dq = zero4;  // phase A: lane = query
#pragma unroll 4
  for (int j = 0; j < N; ++j) {
    const f32x4 kj = ld4(Kb + j * 16 + c0);
    const f32x4 vj = ld4(Vb + j * 16 + c0);
    const float pA = __expf(fmaf(q4[0], kj[0], q4[1] * kj[1]) - mA) * liA;
    const float pB = __expf(fmaf(q4[2], kj[2], q4[3] * kj[3]) - mB) * liB;
    const float dsA = pA * (fmaf(dO[0], vj[0], dO[1] * vj[1]) - dA);
    const float dsB = pB * (fmaf(dO[2], vj[2], dO[3] * vj[3]) - dB);
    dq[0] = fmaf(dsA, kj[0], dq[0]);
    dq[1] = fmaf(dsA, kj[1], dq[1]);
    dq[2] = fmaf(dsB, kj[2], dq[2]);
    dq[3] = fmaf(dsB, kj[3], dq[3]);
  }
  dq = dq * MHA_SCALE;
  f32x4 dk = zero4, dv = zero4;  // phase B: lane = key
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    const f32x4 qi = ld4(Qb + i * 16 + c0);
    const f32x4 doi = ld4(DOb + i * 16 + c0);
    const float miA = Mb[i * 8 + 2 * w], miB = Mb[i * 8 + 2 * w + 1];
    const float lA_ = Lb[i * 8 + 2 * w], lB_ = Lb[i * 8 + 2 * w + 1];
    const float diA = Db[i * 8 + 2 * w], diB = Db[i * 8 + 2 * w + 1];
    const float pA = __expf(fmaf(qi[0], k4[0], qi[1] * k4[1]) - miA) * lA_;
    const float pB = __expf(fmaf(qi[2], k4[2], qi[3] * k4[3]) - miB) * lB_;
    dv[0] = fmaf(pA, doi[0], dv[0]);
    dv[1] = fmaf(pA, doi[1], dv[1]);
    dv[2] = fmaf(pB, doi[2], dv[2]);
    dv[3] = fmaf(pB, doi[3], dv[3]);
    const float dsA = pA * (fmaf(doi[0], v4[0], doi[1] * v4[1]) - diA);
    const float dsB = pB * (fmaf(doi[2], v4[2], doi[3] * v4[3]) - diB);
    dk[0] = fmaf(dsA, qi[0], dk[0]);
    dk[1] = fmaf(dsA, qi[1], dk[1]);
    dk[2] = fmaf(dsB, qi[2], dk[2]);
    dk[3] = fmaf(dsB, qi[3], dk[3]);
  }
  if (!active) {
    dq = zero4;
    dk = zero4;
    dv = zero4;
  }
  __syncthreads();  // every wave is done with Qb/DOb (DQb aliases H1b, DKb aliases DR1b, DVb aliases DF1b: all dead)
  st4(DQb + lane * 16 + c0, dq);
  st4(DKb + lane * 16 + c0, dk);
  st4(DVb + lane * 16 + c0, dv);
  __syncthreads();
  // ---- in-projection: [q;k;v] = Win x + bin ----
  wgrad_slice(DQb, Xb, c0, lane, N, gp + OFF_WIN);
  wgrad_slice(DKb, Xb, c0, lane, N, gp + OFF_WIN + 256);
  wgrad_slice(DVb, Xb, c0, lane, N, gp + OFF_WIN + 512);
  bgrad_slice(dq, c0, lane, gp + OFF_BIN);
  bgrad_slice(dk, c0, lane, gp + OFF_BIN + 16);
  bgrad_slice(dv, c0, lane, gp + OFF_BIN + 32);
  f32x4 dx = dr1;
  ld_row(DQb + lane * 16, row);
  mvt_slice_acc(Wsh + OFF_WIN, c0, row, dx);
  ld_row(DKb + lane * 16, row);
  mvt_slice_acc(Wsh + OFF_WIN + 256, c0, row, dx);
  ld_row(DVb + lane * 16, row);
  mvt_slice_acc(Wsh + OFF_WIN + 512, c0, row, dx);
  if (active) st4(d.dx + (long)b * d.ldx + lane * 16 + c0, dx);
}

int launch_mha(hipStream_t st, const nasrec_mha_desc_t* d) {
  if (d->N < 1 || d->N > MHA_N) return nasrec_set_error(-2, "mha: N=%d out of range [1,%d]", d->N, MHA_N);
  if (d->B == 0) return 0;
  if (d->kind == NASREC_OP_MHA_FWD) {
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(d->B), dim3(256), 0, st, *d);
  } else {
    if (d->saved == nullptr) return nasrec_set_error(-2, "mha backward needs the state saved by the forward launch (desc.saved)");
    hipLaunchKernelGGL(mha_bwd_kernel, dim3(d->B), dim3(256), 0, st, *d);
  }
  return nasrec_check_launch("mha");
}
