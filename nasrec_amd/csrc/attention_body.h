// Helpers and the forward body of the fused Transformer kernels (see attention.hip), shared with chain.hip.
#pragma once
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

// Forward state kept for the backward: N * NASREC_MHA_SAVED floats per sample, as PLANES [token][16] (one per vector) so that a
// plane is 64 N contiguous bytes: the forward copies each plane out of LDS with one fully coalesced 16-byte store per thread
// (token records of 148 floats made every 16-byte piece of a wave's store land in a different cache line: +34 us per launch at
// B = 4096, +4 us at B = 256), and as early as the plane is final, so the store latency hides under the rest of the kernel.
#define SV_Q 0      // plane index: scaled query
#define SV_K 1
#define SV_V 2
#define SV_O 3      // attention output (before the out-projection)
#define SV_H1 4     // LayerNorm-1 output
#define SV_XH1 5    // LayerNorm-1 x-hat
#define SV_F1 6     // FFN hidden (post-ReLU)
#define SV_XH2 7    // LayerNorm-2 x-hat
#define SV_M 8      // per-head softmax max (8) and 1/sum (8)
#define SV_RSTD 9   // [token][4]: 1/std of both LayerNorms, 2 unused
__device__ __forceinline__ float* sv_plane(float* saved, int b, int N, int p) { return saved + ((long)b * N) * NASREC_MHA_SAVED + (long)p * N * 16; }
__device__ __forceinline__ const float* sv_plane(const float* saved, int b, int N, int p) {
  return saved + ((long)b * N) * NASREC_MHA_SAVED + (long)p * N * 16;
}
// LDS plane [token][16] -> global plane, all NT threads, 16 bytes each, contiguous
template <int NT>
__device__ __forceinline__ void sv_copy_out(float* dst, const float* lds, int N, int tid) {
  for (int t = tid; t < N * 4; t += NT) *reinterpret_cast<f32x4*>(dst + 4 * t) = *reinterpret_cast<const f32x4*>(lds + 4 * t);
}

// All 1696 parameters of the node are staged once per workgroup into LDS (6.8 KB) and read back with
// wave-uniform (broadcast) ds_reads: keeping them in SGPRs instead blows the scalar register file.
static __device__ const int kParamOff[12] = {OFF_WIN, OFF_BIN, OFF_WOUT, OFF_BOUT, OFF_L1W, OFF_L1B,
                                             OFF_W1,  OFF_C1,  OFF_W2,   OFF_C2,   OFF_L2W, OFF_L2B};
static __device__ const int kParamLen[12] = {768, 48, 256, 16, 16, 16, 256, 16, 256, 16, 16, 16};

// measured per launch on the bench step (N = 64 / 8 / 48 tokens): forward 15.5 / 8.7 / 13.3 us with 4 waves against
// 17.6 / 9.0 / 14.8 us with 8; backward 23.9 / 27.3 us with 4 waves against 21.1 / 24.7 us with 8
#ifndef MHA_SLICE_FWD
#define MHA_SLICE_FWD 4
#endif
#ifndef MHA_SLICE_BWD
#define MHA_SLICE_BWD 2
#endif

template <int NT>
__device__ __forceinline__ void stage_params(const nasrec_mha_desc_t& d, float* Wsh, int tid) {
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const float* src = d.params[q];
    for (int i = tid; i < kParamLen[q]; i += NT) Wsh[kParamOff[q] + i] = src[i];
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int S>
struct Vec {
  float v[S];
  __device__ __forceinline__ float& operator[](int i) { return v[i]; }
  __device__ __forceinline__ const float& operator[](int i) const { return v[i]; }
};
template <int S>
__device__ __forceinline__ Vec<S> vzero() {
  Vec<S> r;
#pragma unroll
  for (int i = 0; i < S; ++i) r[i] = 0.f;
  return r;
}
// S contiguous floats, S*4-byte aligned (LDS or global)
template <int S>
__device__ __forceinline__ Vec<S> ldv(const float* p) {
  Vec<S> r;
  if (S == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r[0] = t[0]; r[1] = t[1]; r[2 % S] = t[2]; r[3 % S] = t[3];
  } else {
    const f32x2 t = *reinterpret_cast<const f32x2*>(p);
    r[0] = t[0]; r[1] = t[1];
  }
  return r;
}
template <int S>
__device__ __forceinline__ void stv(float* p, const Vec<S>& v) {
  if (S == 4) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2 % S], v[3 % S]};
  } else {
    *reinterpret_cast<f32x2*>(p) = (f32x2){v[0], v[1]};
  }
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

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

// y[r] = b[c0+r] + sum_i W[(c0+r)*16 + i] * x[i], r < S  (W, b in LDS; x = full 16-vector in registers)
template <int S>
__device__ __forceinline__ Vec<S> mv_slice(const float* W, const float* b, int c0, const float* x) {
  Vec<S> y;
#pragma unroll
  for (int r = 0; r < S; ++r) {
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

// y[ii] += sum_o W[o*16 + c0+ii] * g[o], ii < S  (transposed product restricted to the wave's columns)
template <int S>
__device__ __forceinline__ void mvt_slice_acc(const float* W, int c0, const float* g, Vec<S>& y) {
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    const Vec<S> w = ldv<S>(W + o * 16 + c0);
#pragma unroll
    for (int ii = 0; ii < S; ++ii) y[ii] = fmaf(w[ii], g[o], y[ii]);
  }
}

template <int S>
__device__ __forceinline__ float sumv(const Vec<S>& v) {
  if (S == 4) return (v[0] + v[1]) + (v[2 % S] + v[3 % S]);
  return v[0] + v[1];
}

// sum over the NW wave slices of one value per lane (fixed order)
template <int NW>
__device__ __forceinline__ float slice_sum(const float* red, int lane) {
  float s = 0.f;
  if (NW == 4) {
    s = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
  } else {
    s = ((red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane])) +
        ((red[256 + lane] + red[320 + lane]) + (red[384 + lane] + red[448 + lane]));
  }
  return s;
}

// mean and 1/std of a 16-vector whose slices live in the NW waves (lane = token); two LDS exchanges
template <int S>
__device__ __forceinline__ void ln_stats(const Vec<S>& v, float* redA, float* redB, int w, int lane, float& mu, float& rstd) {
  constexpr int NW = 16 / S;
  redA[w * 64 + lane] = sumv<S>(v);
  __syncthreads();
  mu = slice_sum<NW>(redA, lane) * (1.f / 16.f);
  float q = 0.f;
#pragma unroll
  for (int r = 0; r < S; ++r) q += (v[r] - mu) * (v[r] - mu);
  redB[w * 64 + lane] = q;
  __syncthreads();
  const float var = slice_sum<NW>(redB, lane) * (1.f / 16.f);
  rstd = 1.f / sqrtf(var + 1e-5f);
}

// forward of one sample b by one workgroup of 1024 / S threads (called by mha_fwd_kernel and by the per-sample chain kernel)
template <int S>
__device__ __forceinline__ void mha_fwd_sample(const nasrec_mha_desc_t& d, const int b) {
  constexpr int NW = 16 / S, NT = 64 * NW, HP = S / 2;
  __shared__ __attribute__((aligned(16))) float Wsh[NASREC_MHA_PARAMS];
  __shared__ __attribute__((aligned(16))) float Ks[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Vs[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Ob[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Hb[MHA_N * 16];
  __shared__ __attribute__((aligned(16))) float Fb[MHA_N * 16];
  __shared__ float red[4][NT];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c0 = S * w;
  const int N = d.N;
  const bool active = lane < N;
  const bool saving = d.saved != nullptr;  // training: keep what the backward needs instead of recomputing it there
  stage_params<NT>(d, Wsh, tid);
  float x[16];
  if (active) {
    ld_row(d.x + (long)b * d.ldx + lane * 16, x);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
  }
  __syncthreads();
  // in-projection, the wave's S columns of q, k, v
  Vec<S> q4 = mv_slice<S>(Wsh + OFF_WIN, Wsh + OFF_BIN, c0, x);
#pragma unroll
  for (int r = 0; r < S; ++r) q4[r] *= MHA_SCALE;
  const Vec<S> k4 = mv_slice<S>(Wsh + OFF_WIN + 256, Wsh + OFF_BIN + 16, c0, x);
  const Vec<S> v4 = mv_slice<S>(Wsh + OFF_WIN + 512, Wsh + OFF_BIN + 32, c0, x);
  float mx[HP], ls[HP], li[HP];
  Vec<S> o4 = vzero<S>();
  if (S == 4) {
    // Two heads per wave = one packed-fp32 lane pair: K and V rows are parked as (h0c0, h1c0, h0c1, h1c1), so one ds_read_b128
    // yields the operand pairs of v_pk_mul / v_pk_fma (2 heads per instruction; the attention core is ~3/4 of this kernel's
    // vector instructions).  Scores are kept in log2 units (q pre-scaled by log2 e): exp(s - max) = v_exp_f32(s' - max').
    constexpr float LOG2E = 1.44269504088896340736f;
    *reinterpret_cast<f32x4*>(Ks + lane * 16 + c0) = (f32x4){k4[0], k4[2 % S], k4[1], k4[3 % S]};
    *reinterpret_cast<f32x4*>(Vs + lane * 16 + c0) = (f32x4){v4[0], v4[2 % S], v4[1], v4[3 % S]};
    if (saving) {  // q, k, v planes in natural column order (Hb, Fb, Ob are free until later stages)
      stv<S>(Hb + lane * 16 + c0, q4);
      stv<S>(Fb + lane * 16 + c0, k4);
      stv<S>(Ob + lane * 16 + c0, v4);
    }
    __syncthreads();
    if (saving) {
      sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_Q), Hb, N, tid);
      sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_K), Fb, N, tid);
      sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_V), Ob, N, tid);
      __syncthreads();  // Ob is rewritten right after the attention loop (another wave may get there first)
    }
    const f32x2 qa = {q4[0] * LOG2E, q4[2 % S] * LOG2E}, qb = {q4[1] * LOG2E, q4[3 % S] * LOG2E};
    f32x2 m2 = {-INFINITY, -INFINITY};
#pragma unroll 8
    for (int j = 0; j < N; ++j) {
      const f32x4 kj = ld4(Ks + j * 16 + c0);
      const f32x2 s2 = qa * (f32x2){kj[0], kj[1]} + qb * (f32x2){kj[2], kj[3]};
      m2[0] = fmaxf(m2[0], s2[0]);
      m2[1] = fmaxf(m2[1], s2[1]);
    }
    f32x2 l2 = {0.f, 0.f}, oa = {0.f, 0.f}, ob = {0.f, 0.f};
#pragma unroll 8
    for (int j = 0; j < N; ++j) {
      const f32x4 kj = ld4(Ks + j * 16 + c0);
      const f32x4 vj = ld4(Vs + j * 16 + c0);
      const f32x2 t2 = qa * (f32x2){kj[0], kj[1]} + qb * (f32x2){kj[2], kj[3]} - m2;
      const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
      l2 = l2 + p2;
      oa = p2 * (f32x2){vj[0], vj[1]} + oa;
      ob = p2 * (f32x2){vj[2], vj[3]} + ob;
    }
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      mx[h] = m2[h] * (1.f / LOG2E);  // the backward works in natural units
      ls[h] = l2[h];
      li[h] = 1.f / ls[h];
    }
    o4[0] = oa[0] * li[0];
    o4[1] = ob[0] * li[0];
    o4[2 % S] = oa[1] * li[1 % HP];
    o4[3 % S] = ob[1] * li[1 % HP];
  } else {
  stv<S>(Ks + lane * 16 + c0, k4);
  stv<S>(Vs + lane * 16 + c0, v4);
  if (saving) stv<S>(Hb + lane * 16 + c0, q4);
  __syncthreads();
  if (saving) {
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_Q), Hb, N, tid);
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_K), Ks, N, tid);
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_V), Vs, N, tid);
  }
  // attention, the wave's HP heads (head h = columns c0+2h, c0+2h+1)
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    mx[h] = -INFINITY;
    ls[h] = 0.f;
  }
#pragma unroll 4
  for (int j = 0; j < N; ++j) {
    const Vec<S> kj = ldv<S>(Ks + j * 16 + c0);
#pragma unroll
    for (int h = 0; h < HP; ++h) mx[h] = fmaxf(mx[h], fmaf(q4[2 * h], kj[2 * h], q4[2 * h + 1] * kj[2 * h + 1]));
  }
#pragma unroll 4
  for (int j = 0; j < N; ++j) {
    const Vec<S> kj = ldv<S>(Ks + j * 16 + c0);
    const Vec<S> vj = ldv<S>(Vs + j * 16 + c0);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      const float p = __expf(fmaf(q4[2 * h], kj[2 * h], q4[2 * h + 1] * kj[2 * h + 1]) - mx[h]);
      ls[h] += p;
      o4[2 * h] = fmaf(p, vj[2 * h], o4[2 * h]);
      o4[2 * h + 1] = fmaf(p, vj[2 * h + 1], o4[2 * h + 1]);
    }
  }
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    li[h] = 1.f / ls[h];
    o4[2 * h] *= li[h];
    o4[2 * h + 1] *= li[h];
  }
  }
  stv<S>(Ob + lane * 16 + c0, o4);
  __syncthreads();
  if (saving) sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_O), Ob, N, tid);
  // out-projection + residual + LayerNorm 1
  float row[16];
  ld_row(Ob + lane * 16, row);
  Vec<S> r1 = mv_slice<S>(Wsh + OFF_WOUT, Wsh + OFF_BOUT, c0, row);
#pragma unroll
  for (int r = 0; r < S; ++r) r1[r] += x[c0 + r];
  float mu1, rstd1;
  ln_stats<S>(r1, red[0], red[1], w, lane, mu1, rstd1);
  Vec<S> xh1, h1;
#pragma unroll
  for (int r = 0; r < S; ++r) {
    xh1[r] = (r1[r] - mu1) * rstd1;
    h1[r] = xh1[r] * Wsh[OFF_L1W + c0 + r] + Wsh[OFF_L1B + c0 + r];
  }
  stv<S>(Hb + lane * 16 + c0, h1);
  if (saving) stv<S>(Ks + lane * 16 + c0, xh1);  // (K rows are dead since the barrier behind the attention loops)
  __syncthreads();
  if (saving) {
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_H1), Hb, N, tid);
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_XH1), Ks, N, tid);
  }
  // FFN
  ld_row(Hb + lane * 16, row);
  Vec<S> f1 = mv_slice<S>(Wsh + OFF_W1, Wsh + OFF_C1, c0, row);
#pragma unroll
  for (int r = 0; r < S; ++r) f1[r] = fmaxf(f1[r], 0.f);
  stv<S>(Fb + lane * 16 + c0, f1);
  __syncthreads();
  if (saving) sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_F1), Fb, N, tid);
  ld_row(Fb + lane * 16, row);
  Vec<S> r2 = mv_slice<S>(Wsh + OFF_W2, Wsh + OFF_C2, c0, row);
#pragma unroll
  for (int r = 0; r < S; ++r) r2[r] += h1[r];
  float mu2, rstd2;
  ln_stats<S>(r2, red[2], red[3], w, lane, mu2, rstd2);
  Vec<S> xh2, out;
  const bool masked = d.dims_in_use >= 0 && lane >= d.dims_in_use;
#pragma unroll
  for (int r = 0; r < S; ++r) {
    xh2[r] = (r2[r] - mu2) * rstd2;
    out[r] = masked ? 0.f : xh2[r] * Wsh[OFF_L2W + c0 + r] + Wsh[OFF_L2B + c0 + r];
  }
  if (active) stv<S>(d.out + (long)b * d.ldo + lane * 16 + c0, out);
  if (saving) {
    // x-hat of LayerNorm 2 and the softmax statistics: through the dead V / O rows (every wave is past its reads of them: the
    // LayerNorm barriers above), then planes like the rest
    stv<S>(Vs + lane * 16 + c0, xh2);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      Ob[lane * 16 + HP * w + h] = mx[h];
      Ob[lane * 16 + 8 + HP * w + h] = li[h];
    }
    if (w == 0 && active) *reinterpret_cast<f32x4*>(sv_plane(d.saved, b, N, SV_RSTD) + lane * 4) = (f32x4){rstd1, rstd2, 0.f, 0.f};
    __syncthreads();
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_XH2), Vs, N, tid);
    sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_M), Ob, N, tid);
  }
}

