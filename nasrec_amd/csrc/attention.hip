// Fused Transformer body (modules.py:664-686): nn.MultiheadAttention(embed 16, 8 heads x head_dim 2,
// batch_first, no dropout/mask) + residual + LayerNorm(16) + Linear(16,16) + ReLU + Linear(16,16) + residual +
// LayerNorm(16) (+ supernet token prefix mask).
//
// Mapping: one workgroup = one sample; lane = token (N <= 64), wave w owns the S-column slice [S*w, S*w+S) of every
// 16-wide vector — i.e. S/2 heads of the attention, S rows of every projection (template parameter S: 2 -> 8 waves, one
// head each; 4 -> 4 waves).  Full 16-vectors that a slice computation needs (the other waves' columns) go through LDS rows
// [token][16].  With one sample per wavefront only 256 of the chip's 1024 SIMDs had work at batch 256; 4 waves give every
// SIMD one wave; 8 waves give it two, so that one wave's LDS / dependent-FMA latency hides under the other's issue —
// which pays in the backward launch (long dependent chains) and not in the forward one (MHA_SLICE_FWD / _BWD).
// Attention at head_dim 2 has nothing for MFMA to chew on (K = 2): K/V slices are parked in LDS and every lane walks
// the keys (one ds_read_b128 per key for K, one for V; 2 FMA + 1 exp per key and head).
// The forward launch saves per-token state (NASREC_MHA_SAVED floats); the backward launch reads it, runs the
// flash-style two-phase attention backward (lane = query for dq, lane = key for dk/dv), and reduces the 1696
// parameter gradients of the node over the sample's tokens: weight gradients as LDS outer products (lane = one
// (row, column) entry of the wave's 4x16 slice), bias / LayerNorm gradients as wave reductions; per-sample
// partials are summed across the batch in fixed order by NASREC_OP_REDUCE_ROWS (deterministic).
#include "attention_body.h"

template <int S>
__global__ __launch_bounds__(1024 / S) void mha_fwd_kernel(const nasrec_mha_desc_t d) {
  mha_fwd_sample<S>(d, blockIdx.x);
}

// Weight-gradient slice of one product y = W v (W [16,16]): dW[c0+o][i] = sum_tok G[tok][c0+o] * V[tok][i], o < S.
// The wave's S*16 entries are spread over the lanes; with S = 2 the two half-waves take alternate tokens and are
// combined with one cross-lane add.  G, V are LDS rows [token][16].
template <int S>
__device__ __forceinline__ void wgrad_slice(const float* G, const float* V, int c0, int lane, int N, float* out) {
  constexpr int TG = 4 / S;  // token groups per entry
  const int o = (lane >> 4) & (S - 1), i = lane & 15, tg = lane >> (4 + (S == 4 ? 2 : 1));
  float s = 0.f;
#pragma unroll 8
  for (int t = tg; t < N; t += TG) s = fmaf(G[t * 16 + c0 + o], V[t * 16 + i], s);
  if (TG == 2) s += __shfl_xor(s, 32, 64);
  if (tg == 0) out[(c0 + o) * 16 + i] = s;
}

// The same product for the whole 16 x 16 matrix by ONE wave on the matrix cores: dW = G^T V is a [16, N] x [N, 16] product, i.e.
// ceil(N / 4) v_mfma_f32_16x16x4_f32 with k = token.  Lane (r = lane & 15, g = lane >> 4) feeds A(o = r, k = 4 step + g) =
// G[4 step + g][r] and B(k, i = r) = V[4 step + g][r]: both are 64 consecutive LDS floats per step (rows 4 step .. 4 step + 3),
// conflict-free; tokens >= N contribute zeros.  16 MFMAs + 32 LDS reads instead of 64 x 3 instructions in each of the waves
// (exact fp32 FMA chains; only the summation order over the tokens differs from the loop form).  Used by the 4-wave (large
// batch) backward: 222 -> 213 us at B = 4096; at batch 256 the serial MFMA chain of one wave costs 0.6 us more than the slices.
__device__ __forceinline__ void wgrad_mfma(const float* G, const float* V, int lane, int N, float* out) {
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < N; t0 += 4) {
    const int t = t0 + g;
    const float a = t < N ? G[t * 16 + r] : 0.f;
    const float b = t < N ? V[t * 16 + r] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  // D: row o = 4 * (lane >> 4) + reg, column i = lane & 15
#pragma unroll
  for (int q = 0; q < 4; ++q) out[(4 * g + q) * 16 + r] = acc[q];
}

// bias-like gradient slice: out[c0+r] = sum over tokens (lanes) of g[r]
template <int S>
__device__ __forceinline__ void bgrad_slice(const Vec<S>& g, int c0, int lane, float* out) {
#pragma unroll
  for (int r = 0; r < S; ++r) {
    const float s = wave_sum(g[r]);
    if (lane == 0) out[c0 + r] = s;
  }
}

template <int S>
__device__ __forceinline__ Vec<S> vmul(const Vec<S>& a, const Vec<S>& b) {
  Vec<S> r;
#pragma unroll
  for (int i = 0; i < S; ++i) r[i] = a[i] * b[i];
  return r;
}

template <int S>
__global__ __launch_bounds__(1024 / S) __attribute__((amdgpu_waves_per_eu(3))) void mha_bwd_kernel(const nasrec_mha_desc_t d) {
  constexpr int NW = 16 / S, NT = 64 * NW, HP = S / 2;
  __shared__ __attribute__((aligned(16))) float Wsh[NASREC_MHA_PARAMS];
  __shared__ __attribute__((aligned(16))) float Bf[9][MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Mb[MHA_N * 8];
  __shared__ __attribute__((aligned(16))) float Lb[MHA_N * 8];
  __shared__ __attribute__((aligned(16))) float Db[MHA_N * 8];
  __shared__ float red[2][NT];  // both LayerNorm stages (barriers separate their uses)
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
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c0 = S * w;
  const int N = d.N;
  const bool active = lane < N;
  float* gp = d.dparams_partial + (long)b * (d.partial_ld > 0 ? d.partial_ld : NASREC_MHA_PARAMS);
  stage_params<NT>(d, Wsh, tid);
  // ---- the sample's planes: global -> LDS with contiguous 16-byte accesses (x and the forward state are [token][16] planes) -----
  {
    const int n4 = N * 4;
    auto plane_in = [&](float* lds, const float* src) {
      for (int t = tid; t < n4; t += NT) *reinterpret_cast<f32x4*>(lds + 4 * t) = *reinterpret_cast<const f32x4*>(src + 4 * t);
    };
    plane_in(Xb, d.x + (long)b * d.ldx);
    plane_in(Qb, sv_plane(d.saved, b, N, SV_Q));
    plane_in(Kb, sv_plane(d.saved, b, N, SV_K));
    plane_in(Vb, sv_plane(d.saved, b, N, SV_V));
    plane_in(Ob, sv_plane(d.saved, b, N, SV_O));
    plane_in(H1b, sv_plane(d.saved, b, N, SV_H1));
    plane_in(F1b, sv_plane(d.saved, b, N, SV_F1));
    plane_in(Bf[7], sv_plane(d.saved, b, N, SV_XH1));  // x-hats: only on their way to registers
    plane_in(Bf[8], sv_plane(d.saved, b, N, SV_XH2));
    const float* ml = sv_plane(d.saved, b, N, SV_M);
    for (int t = tid; t < n4; t += NT) {  // [token][8 max | 8 1/sum] -> Mb, Lb
      const f32x4 v = *reinterpret_cast<const f32x4*>(ml + 4 * t);
      float* dst = ((t & 2) ? Lb : Mb) + (t >> 2) * 8 + (t & 1) * 4;
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  }
  __syncthreads();  // parameters and the token rows are in LDS
  Vec<S> x4 = vzero<S>(), q4 = vzero<S>(), k4 = vzero<S>(), v4 = vzero<S>(), o4 = vzero<S>(), h1 = vzero<S>(), xh1 = vzero<S>(),
         f1 = vzero<S>(), xh2 = vzero<S>(), dout = vzero<S>();
  float mx[HP], li[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    mx[h] = 0.f;
    li[h] = 1.f;
  }
  float rstd1 = 1.f, rstd2 = 1.f;
  if (active) {
    const int o = lane * 16 + c0;
    x4 = ldv<S>(Xb + o);
    q4 = ldv<S>(Qb + o);
    k4 = ldv<S>(Kb + o);
    v4 = ldv<S>(Vb + o);
    o4 = ldv<S>(Ob + o);
    h1 = ldv<S>(H1b + o);
    f1 = ldv<S>(F1b + o);
    xh1 = ldv<S>(Bf[7] + o);
    xh2 = ldv<S>(Bf[8] + o);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      mx[h] = Mb[lane * 8 + HP * w + h];
      li[h] = Lb[lane * 8 + HP * w + h];
    }
    const float* rs = sv_plane(d.saved, b, N, SV_RSTD) + lane * 4;
    rstd1 = rs[0];
    rstd2 = rs[1];
    if (!(d.dims_in_use >= 0 && lane >= d.dims_in_use)) dout = ldv<S>(d.dout + (long)b * d.ldo + lane * 16 + c0);
  }
  // ---- LayerNorm 2 ----
  bgrad_slice<S>(vmul<S>(dout, xh2), c0, lane, gp + OFF_L2W);
  bgrad_slice<S>(dout, c0, lane, gp + OFF_L2B);
  Vec<S> gw;
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int r = 0; r < S; ++r) {
    gw[r] = dout[r] * Wsh[OFF_L2W + c0 + r];
    sa += gw[r];
    sb += gw[r] * xh2[r];
  }
  red[0][w * 64 + lane] = sa;
  red[1][w * 64 + lane] = sb;
  __syncthreads();
  float c1 = slice_sum<NW>(red[0], lane) * (1.f / 16.f);
  float c2 = slice_sum<NW>(red[1], lane) * (1.f / 16.f);
  Vec<S> dr2;
#pragma unroll
  for (int r = 0; r < S; ++r) dr2[r] = (gw[r] - c1 - xh2[r] * c2) * rstd2;
  stv<S>(DR2b + lane * 16 + c0, dr2);
  __syncthreads();
  // ---- FFN 2: f2 = W2 f1 + c2 ----
  if (S == 4) {
    if (w == 0) wgrad_mfma(DR2b, F1b, lane, N, gp + OFF_W2);
  } else {
    wgrad_slice<S>(DR2b, F1b, c0, lane, N, gp + OFF_W2);
  }
  bgrad_slice<S>(dr2, c0, lane, gp + OFF_C2);
  float row[16];
  ld_row(DR2b + lane * 16, row);
  Vec<S> df1 = vzero<S>();
  mvt_slice_acc<S>(Wsh + OFF_W2, c0, row, df1);
#pragma unroll
  for (int r = 0; r < S; ++r) df1[r] = f1[r] > 0.f ? df1[r] : 0.f;
  stv<S>(DF1b + lane * 16 + c0, df1);
  __syncthreads();
  // ---- FFN 1: f1 = relu(W1 h1 + c1) ----
  if (S == 4) {
    if (w == 1) wgrad_mfma(DF1b, H1b, lane, N, gp + OFF_W1);
  } else {
    wgrad_slice<S>(DF1b, H1b, c0, lane, N, gp + OFF_W1);
  }
  bgrad_slice<S>(df1, c0, lane, gp + OFF_C1);
  ld_row(DF1b + lane * 16, row);
  Vec<S> dh1 = dr2;
  mvt_slice_acc<S>(Wsh + OFF_W1, c0, row, dh1);
  // ---- LayerNorm 1 ----
  bgrad_slice<S>(vmul<S>(dh1, xh1), c0, lane, gp + OFF_L1W);
  bgrad_slice<S>(dh1, c0, lane, gp + OFF_L1B);
  sa = 0.f;
  sb = 0.f;
#pragma unroll
  for (int r = 0; r < S; ++r) {
    gw[r] = dh1[r] * Wsh[OFF_L1W + c0 + r];
    sa += gw[r];
    sb += gw[r] * xh1[r];
  }
  red[0][w * 64 + lane] = sa;
  red[1][w * 64 + lane] = sb;
  __syncthreads();  // also: every wave is done reading DR2b / F1b / H1b / DF1b
  c1 = slice_sum<NW>(red[0], lane) * (1.f / 16.f);
  c2 = slice_sum<NW>(red[1], lane) * (1.f / 16.f);
  Vec<S> dr1;
#pragma unroll
  for (int r = 0; r < S; ++r) dr1[r] = (gw[r] - c1 - xh1[r] * c2) * rstd1;
  stv<S>(DR1b + lane * 16 + c0, dr1);
  __syncthreads();
  // ---- out-projection: a = Wout o + bout ----
  if (S == 4) {
    if (w == 2) wgrad_mfma(DR1b, Ob, lane, N, gp + OFF_WOUT);
  } else {
    wgrad_slice<S>(DR1b, Ob, c0, lane, N, gp + OFF_WOUT);
  }
  bgrad_slice<S>(dr1, c0, lane, gp + OFF_BOUT);
  ld_row(DR1b + lane * 16, row);
  Vec<S> dO = vzero<S>();
  mvt_slice_acc<S>(Wsh + OFF_WOUT, c0, row, dO);
  float dd[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    dd[h] = fmaf(dO[2 * h], o4[2 * h], dO[2 * h + 1] * o4[2 * h + 1]);
    Db[lane * 8 + HP * w + h] = dd[h];
  }
  stv<S>(DOb + lane * 16 + c0, dO);
  __syncthreads();  // also: every wave is done reading DR1b
  // ---- attention backward, the wave's HP heads ----
  Vec<S> dq = vzero<S>();  // phase A: lane = query
#pragma unroll 4
  for (int j = 0; j < N; ++j) {
    const Vec<S> kj = ldv<S>(Kb + j * 16 + c0);
    const Vec<S> vj = ldv<S>(Vb + j * 16 + c0);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      const float p = __expf(fmaf(q4[2 * h], kj[2 * h], q4[2 * h + 1] * kj[2 * h + 1]) - mx[h]) * li[h];
      const float ds = p * (fmaf(dO[2 * h], vj[2 * h], dO[2 * h + 1] * vj[2 * h + 1]) - dd[h]);
      dq[2 * h] = fmaf(ds, kj[2 * h], dq[2 * h]);
      dq[2 * h + 1] = fmaf(ds, kj[2 * h + 1], dq[2 * h + 1]);
    }
  }
#pragma unroll
  for (int r = 0; r < S; ++r) dq[r] *= MHA_SCALE;
  Vec<S> dk = vzero<S>(), dv = vzero<S>();  // phase B: lane = key
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    const Vec<S> qi = ldv<S>(Qb + i * 16 + c0);
    const Vec<S> doi = ldv<S>(DOb + i * 16 + c0);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      const float mi = Mb[i * 8 + HP * w + h], l_ = Lb[i * 8 + HP * w + h], di = Db[i * 8 + HP * w + h];
      const float p = __expf(fmaf(qi[2 * h], k4[2 * h], qi[2 * h + 1] * k4[2 * h + 1]) - mi) * l_;
      dv[2 * h] = fmaf(p, doi[2 * h], dv[2 * h]);
      dv[2 * h + 1] = fmaf(p, doi[2 * h + 1], dv[2 * h + 1]);
      const float ds = p * (fmaf(doi[2 * h], v4[2 * h], doi[2 * h + 1] * v4[2 * h + 1]) - di);
      dk[2 * h] = fmaf(ds, qi[2 * h], dk[2 * h]);
      dk[2 * h + 1] = fmaf(ds, qi[2 * h + 1], dk[2 * h + 1]);
    }
  }
  if (!active) {
    dq = vzero<S>();
    dk = vzero<S>();
    dv = vzero<S>();
  }
  __syncthreads();  // every wave is done with Qb/DOb (DQb aliases H1b, DKb aliases DR1b, DVb aliases DF1b: all dead)
  stv<S>(DQb + lane * 16 + c0, dq);
  stv<S>(DKb + lane * 16 + c0, dk);
  stv<S>(DVb + lane * 16 + c0, dv);
  __syncthreads();
  // ---- in-projection: [q;k;v] = Win x + bin ----
  if (S == 4) {  // one matrix per wave on the matrix cores (large batch); the 8-wave form keeps the row slices (latency)
    if (w == 0) wgrad_mfma(DQb, Xb, lane, N, gp + OFF_WIN);
    if (w == 1) wgrad_mfma(DKb, Xb, lane, N, gp + OFF_WIN + 256);
    if (w == 2) wgrad_mfma(DVb, Xb, lane, N, gp + OFF_WIN + 512);
  } else {
    wgrad_slice<S>(DQb, Xb, c0, lane, N, gp + OFF_WIN);
    wgrad_slice<S>(DKb, Xb, c0, lane, N, gp + OFF_WIN + 256);
    wgrad_slice<S>(DVb, Xb, c0, lane, N, gp + OFF_WIN + 512);
  }
  bgrad_slice<S>(dq, c0, lane, gp + OFF_BIN);
  bgrad_slice<S>(dk, c0, lane, gp + OFF_BIN + 16);
  bgrad_slice<S>(dv, c0, lane, gp + OFF_BIN + 32);
  Vec<S> dx = dr1;
  ld_row(DQb + lane * 16, row);
  mvt_slice_acc<S>(Wsh + OFF_WIN, c0, row, dx);
  ld_row(DKb + lane * 16, row);
  mvt_slice_acc<S>(Wsh + OFF_WIN + 256, c0, row, dx);
  ld_row(DVb + lane * 16, row);
  mvt_slice_acc<S>(Wsh + OFF_WIN + 512, c0, row, dx);
  if (active) stv<S>(d.dx + (long)b * d.ldx + lane * 16 + c0, dx);
}

int launch_mha(hipStream_t st, const nasrec_mha_desc_t* d) {
  if (d->N < 1 || d->N > MHA_N) return nasrec_set_error(-2, "mha: N=%d out of range [1,%d]", d->N, MHA_N);
  if (d->B == 0) return 0;
  if (d->kind == NASREC_OP_MHA_FWD) {
    hipLaunchKernelGGL(mha_fwd_kernel<MHA_SLICE_FWD>, dim3(d->B), dim3(1024 / MHA_SLICE_FWD), 0, st, *d);
  } else {
    if (d->saved == nullptr) return nasrec_set_error(-2, "mha backward needs the state saved by the forward launch (desc.saved)");
    // 8 waves per sample win where latency counts (batch 256: 23.5 against 26.7 us); at large batch the 4-wave form does the
    // same work with fewer wave-instructions per sample (B = 4096: 269 against 286 us)
    if (d->B >= 1024)
      hipLaunchKernelGGL(mha_bwd_kernel<4>, dim3(d->B), dim3(256), 0, st, *d);
    else
      hipLaunchKernelGGL(mha_bwd_kernel<MHA_SLICE_BWD>, dim3(d->B), dim3(1024 / MHA_SLICE_BWD), 0, st, *d);
  }
  return nasrec_check_launch("mha");
}
