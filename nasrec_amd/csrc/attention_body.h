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

// All loads first, all LDS stores after: a loop per parameter array compiles to load -> wait -> store per array, i.e. twelve
// dependent memory round trips before the kernel does anything (0.5 - 2 us each on the cold L2 of a batch-256 step).  Every array is
// a multiple of 16 floats and 16-byte aligned (the parameter arena aligns each parameter), so the 1696 floats are 424 16-byte
// pieces, at most two per thread: the piece's array is found by compares on its offset.
template <int NT>
struct ParamPieces {
  static constexpr int PIECES = NASREC_MHA_PARAMS / 4, PER = (PIECES + NT - 1) / NT;
  f32x4 v[PER];
};
// (two halves, so that a caller can put its own loads between them: everything the kernel needs first is then ONE round trip)
template <int NT>
__device__ __forceinline__ void stage_params_load(const nasrec_mha_desc_t& d, int tid, ParamPieces<NT>& pp) {
  static_assert(NASREC_MHA_PARAMS % 4 == 0, "16-byte pieces");
#pragma unroll
  for (int k = 0; k < ParamPieces<NT>::PER; ++k) {
    const int off = 4 * min(tid + k * NT, ParamPieces<NT>::PIECES - 1);  // float offset of the piece inside the 1696-float record (clamped: the store is skipped)
    // (the pointers go through readfirstlane: left as a select over d.params[q] the compiler turns the chain into a per-lane LOAD of
    // the pointer from the argument array — one more dependent round trip)
    unsigned lo = 0, hi = 0;
    int base = 0;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const unsigned long long pq = (unsigned long long)d.params[q];
      const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pq), phi = __builtin_amdgcn_readfirstlane((unsigned)(pq >> 32));
      const bool in = off >= kParamOff[q];
      lo = in ? plo : lo;
      hi = in ? phi : hi;
      base = in ? kParamOff[q] : base;
    }
    typedef __attribute__((address_space(1))) const f32x4 gf32x4;  // (global, not flat: the integer round trip loses the address space)
    pp.v[k] = *reinterpret_cast<gf32x4*>((((unsigned long long)hi << 32) | lo) + 4ull * (unsigned)(off - base));
  }
}
template <int NT>
__device__ __forceinline__ void stage_params_store(float* Wsh, int tid, const ParamPieces<NT>& pp) {
#pragma unroll
  for (int k = 0; k < ParamPieces<NT>::PER; ++k)
    if (tid + k * NT < ParamPieces<NT>::PIECES) *reinterpret_cast<f32x4*>(Wsh + 4 * (tid + k * NT)) = pp.v[k];
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
// LDS floats of the forward / backward bodies (the caller owns the buffer: a kernel of its own, or the shared buffer of a worklist launch)
#define MHA_FWD_LDS_FLOATS(S) (NASREC_MHA_PARAMS + 5 * MHA_N * 16 + 4 * 64 * (16 / (S)))
#define MHA_BWD_LDS_FLOATS(S) (NASREC_MHA_PARAMS + 9 * MHA_N * 16 + 3 * MHA_N * 8 + 2 * 64 * (16 / (S)))

#ifdef MHA_STAMPS  // timing-only build (tools/mha_stamps.py): wave 0 keeps the clock at the stage boundaries and overwrites the first words of its parameter-gradient partial (backward) / output row (forward) with it
#define MHA_STAMP(i) mha_st[i] = (unsigned)__builtin_readcyclecounter()
#else
#define MHA_STAMP(i)
#endif
template <int S>
__device__ __forceinline__ void mha_fwd_sample(const nasrec_mha_desc_t& d, const int b, float* lds) {
  constexpr int NW = 16 / S, NT = 64 * NW, HP = S / 2;
  static_assert(NASREC_MHA_PARAMS % 4 == 0, "16-byte aligned carve");
  float* Wsh = lds;
  float* Ks = Wsh + NASREC_MHA_PARAMS;
  float* Vs = Ks + MHA_N * 16;
  float* Ob = Vs + MHA_N * 16;
  float* Hb = Ob + MHA_N * 16;
  float* Fb = Hb + MHA_N * 16;
  float(*red)[NT] = reinterpret_cast<float(*)[NT]>(Fb + MHA_N * 16);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c0 = S * w;
  const int N = d.N;
  const bool active = lane < N;
  const bool saving = d.saved != nullptr;  // training: keep what the backward needs instead of recomputing it there
#ifdef MHA_STAMPS
  unsigned mha_st[16];
#endif
  MHA_STAMP(0);
  float x[16];  // (issued before the parameters are parked: one round trip for both)
  ParamPieces<NT> pp;
  stage_params_load<NT>(d, tid, pp);
  ld_row(d.x + (long)b * d.ldx + min(lane, N - 1) * 16, x);
  stage_params_store<NT>(Wsh, tid, pp);
  if (!active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
  }
  __syncthreads();
  MHA_STAMP(1);
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
    MHA_STAMP(2);
    const f32x2 qa = {q4[0] * LOG2E, q4[2 % S] * LOG2E}, qb = {q4[1] * LOG2E, q4[3 % S] * LOG2E};
    f32x2 m2 = {-INFINITY, -INFINITY};
#pragma unroll 8
    for (int j = 0; j < N; ++j) {
      const f32x4 kj = ld4(Ks + j * 16 + c0);
      const f32x2 s2 = qa * (f32x2){kj[0], kj[1]} + qb * (f32x2){kj[2], kj[3]};
      m2[0] = fmaxf(m2[0], s2[0]);
      m2[1] = fmaxf(m2[1], s2[1]);
    }
    MHA_STAMP(3);
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
  MHA_STAMP(4);
  stv<S>(Ob + lane * 16 + c0, o4);
  __syncthreads();
  if (saving) sv_copy_out<NT>(sv_plane(d.saved, b, N, SV_O), Ob, N, tid);
  MHA_STAMP(5);
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
  MHA_STAMP(6);
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
  MHA_STAMP(7);
  float mu2, rstd2;
  ln_stats<S>(r2, red[2], red[3], w, lane, mu2, rstd2);
  MHA_STAMP(8);
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
#ifdef MHA_STAMPS
  MHA_STAMP(9);
  __syncthreads();
  if (tid == 0)
    for (int i = 0; i < 10; ++i) d.out[(long)b * d.ldo + i] = __builtin_bit_cast(float, mha_st[i]);
#endif
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
  // every LDS read of the product is issued before the first MFMA (a loop of read, wait, multiply was 16 dependent LDS round trips
  // on one wave while the others wait at the stage's barrier: tools/mha_stamps.py); one accumulator, tokens in order as before.
  // (Running it on EVERY wave of the one-matrix stages — each on its own SIMD's idle matrix pipe, one storing — so that the vector
  // work of the stage could sit between the MFMAs moved time between the stages and left the body where it was: 445 -> 435.)
  float a[MHA_N / 4], b[MHA_N / 4];
#pragma unroll
  for (int s = 0; s < MHA_N / 4; ++s) {
    const int t = 4 * s + g;
    const int tt = t < N ? t : 0;
    a[s] = G[tt * 16 + r];
    b[s] = V[tt * 16 + r];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < MHA_N / 4; ++s) {
    if (4 * s < N) {  // (uniform)
      const bool ok = 4 * s + g < N;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? a[s] : 0.f, ok ? b[s] : 0.f, acc, 0, 0, 0);
    }
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

// backward of one sample b by one workgroup of 1024 / S threads (mha_bwd_kernel, and the worklist launches of csrc/worklist.hip)
template <int S>
__device__ __forceinline__ void mha_bwd_sample(const nasrec_mha_desc_t& d, const int b, float* lds) {
  constexpr int NW = 16 / S, NT = 64 * NW, HP = S / 2;
  float* Wsh = lds;
  float(*Bf)[MHA_N * 16] = reinterpret_cast<float(*)[MHA_N * 16]>(Wsh + NASREC_MHA_PARAMS);
  float* Mb = Wsh + NASREC_MHA_PARAMS + 9 * MHA_N * 16;
  float* Lb = Mb + MHA_N * 8;
  float* Db = Lb + MHA_N * 8;
  float(*red)[NT] = reinterpret_cast<float(*)[NT]>(Db + MHA_N * 8);  // both LayerNorm stages (barriers separate their uses)
  // LDS rows [token][16]; buffers are re-used once their previous content is dead (a barrier separates the uses)
  float* Xb = Bf[0];
  float* Qb = Bf[1];
  float* Kb = Bf[2];
  float* Vb = Bf[3];
  float* Ob = Bf[4];
  float* F1b = Bf[5];
  float* DOb = Bf[5];   // after the FFN-2 stage
  float* H1b = Bf[6];
  float* DR2b = Bf[7];
  float* DR1b = Bf[7];  // after the FFN-1 stage
  float* DF1b = Bf[8];
  // dq / dk / dv take the rows of q / k / v once the attention loops are done with them, so that DR1b and Ob (the operands of the
  // out-projection's weight gradient) live to the last stage, where a wave is free for it
  float* DQb = Bf[1];
  float* DKb = Bf[2];
  float* DVb = Bf[3];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c0 = S * w;
  const int N = d.N;
  const bool active = lane < N;
  float* gp = d.dparams_partial + (long)b * (d.partial_ld > 0 ? d.partial_ld : NASREC_MHA_PARAMS);
  // the lane's own global operands (token = lane): issued with the planes, used after the first barrier
#ifdef MHA_STAMPS
  unsigned mha_st[16];
#endif
  MHA_STAMP(0);
  const int tl = min(lane, N - 1);
  const f32x2 rs2 = *reinterpret_cast<const f32x2*>(sv_plane(d.saved, b, N, SV_RSTD) + tl * 4);
  Vec<S> dout_in = ldv<S>(d.dout + (long)b * d.ldo + tl * 16 + c0);
  ParamPieces<NT> pp;
  stage_params_load<NT>(d, tid, pp);
  // ---- the sample's planes: global -> LDS with contiguous 16-byte accesses (x and the forward state are [token][16] planes).  All ten
  // loads are issued before the first store: plane after plane (load, wait, store) was ten dependent round trips, about half of this
  // kernel's time at batch 256 (one workgroup per CU, nothing else to hide them) ----------------------------------------------------
  {
    const int n4 = N * 4;  // 16-byte pieces per plane; N <= MHA_N = 64 tokens: at most one per thread
    static_assert(4 * MHA_N <= NT || NT >= 256, "one piece per thread and plane");
    const int t = min(tid, n4 - 1);
    const float* src[10] = {d.x + (long)b * d.ldx,          sv_plane(d.saved, b, N, SV_Q),   sv_plane(d.saved, b, N, SV_K),   sv_plane(d.saved, b, N, SV_V),
                            sv_plane(d.saved, b, N, SV_O),  sv_plane(d.saved, b, N, SV_H1),  sv_plane(d.saved, b, N, SV_F1),  sv_plane(d.saved, b, N, SV_XH1),
                            sv_plane(d.saved, b, N, SV_XH2), sv_plane(d.saved, b, N, SV_M)};
    float* dst[9] = {Xb, Qb, Kb, Vb, Ob, H1b, F1b, Bf[7], Bf[8]};  // (x-hats: only on their way to registers)
    f32x4 v[10];
#pragma unroll
    for (int p = 0; p < 10; ++p) v[p] = *reinterpret_cast<const f32x4*>(src[p] + 4 * t);
    stage_params_store<NT>(Wsh, tid, pp);
    if (tid < n4) {
      if (S == 4) {  // q, k, v rows as (h0c0, h1c0, h0c1, h1c1): the operand pairs of the packed attention loops below
#pragma unroll
        for (int p = 1; p < 4; ++p) v[p] = (f32x4){v[p][0], v[p][2], v[p][1], v[p][3]};
      }
#pragma unroll
      for (int p = 0; p < 9; ++p) *reinterpret_cast<f32x4*>(dst[p] + 4 * tid) = v[p];
      // [token][8 max | 8 1/sum] -> Mb, Lb
      *reinterpret_cast<f32x4*>(((tid & 2) ? Lb : Mb) + (tid >> 2) * 8 + (tid & 1) * 4) = v[9];
    }
  }
  __syncthreads();  // parameters and the token rows are in LDS
  MHA_STAMP(1);
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
    rstd1 = rs2[0];
    rstd2 = rs2[1];
    if (!(d.dims_in_use >= 0 && lane >= d.dims_in_use)) dout = dout_in;
  }
  // The attention loops recompute p = exp(s - max) / sum as ONE exponential: exp2(s' - m'), s' = (q log2 e) . k and m' = (max + ln sum) log2 e
  // (round 5: the loops are 42 % of this body and issue-bound; the multiply by 1 / sum and the log2 e scaling inside expf were three
  // of ~15 vector instructions per key and head pair).  mq: the lane's own rows as queries; Mb below: every row, for the lanes as keys.
  constexpr float LOG2E = 1.44269504088896340736f;
  float mq[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) mq[h] = (mx[h] - __logf(li[h])) * LOG2E;
  MHA_STAMP(2);
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
  __syncthreads();  // (also: every lane has its own max / 1 / sum in registers)
  for (int i = tid; i < N * 8; i += NT) Mb[i] = (Mb[i] - __logf(Lb[i])) * LOG2E;  // m' per (row, head): read by the key-side loop, several barriers on
  float c1 = slice_sum<NW>(red[0], lane) * (1.f / 16.f);
  float c2 = slice_sum<NW>(red[1], lane) * (1.f / 16.f);
  Vec<S> dr2;
#pragma unroll
  for (int r = 0; r < S; ++r) dr2[r] = (gw[r] - c1 - xh2[r] * c2) * rstd2;
  stv<S>(DR2b + lane * 16 + c0, dr2);
  __syncthreads();
  MHA_STAMP(3);
  // ---- FFN 2: f2 = W2 f1 + c2 ----
  // (4-wave form: a 16 x 16 weight gradient is one wave's MFMA chain while the others wait at the stage's barrier, and nothing inside the
  // kernel reads it — so the six chains sit in TWO stages, on different waves, instead of one in each of four: W2 and W1 in the FFN-1
  // stage (their operands live until LayerNorm 1 writes DR1b), Wout with the three in-projection matrices in the last.  Same operands,
  // same chains: same bits.)
  if (S != 4) wgrad_slice<S>(DR2b, F1b, c0, lane, N, gp + OFF_W2);
  bgrad_slice<S>(dr2, c0, lane, gp + OFF_C2);
  float row[16];
  ld_row(DR2b + lane * 16, row);
  Vec<S> df1 = vzero<S>();
  mvt_slice_acc<S>(Wsh + OFF_W2, c0, row, df1);
#pragma unroll
  for (int r = 0; r < S; ++r) df1[r] = f1[r] > 0.f ? df1[r] : 0.f;
  stv<S>(DF1b + lane * 16 + c0, df1);
  __syncthreads();
  MHA_STAMP(4);
  // ---- FFN 1: f1 = relu(W1 h1 + c1) ----
  if (S == 4) {
    if (w == 0) wgrad_mfma(DR2b, F1b, lane, N, gp + OFF_W2);
    if (w == 1) wgrad_mfma(DF1b, H1b, lane, N, gp + OFF_W1);
  } else {
    wgrad_slice<S>(DF1b, H1b, c0, lane, N, gp + OFF_W1);
  }
  bgrad_slice<S>(df1, c0, lane, gp + OFF_C1);
  ld_row(DF1b + lane * 16, row);
  Vec<S> dh1 = dr2;
  mvt_slice_acc<S>(Wsh + OFF_W1, c0, row, dh1);
  MHA_STAMP(5);
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
  MHA_STAMP(6);
  // ---- out-projection: a = Wout o + bout ----
  if (S != 4) wgrad_slice<S>(DR1b, Ob, c0, lane, N, gp + OFF_WOUT);
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
  if (S == 4) {
    *reinterpret_cast<f32x4*>(DOb + lane * 16 + c0) = (f32x4){dO[0], dO[2 % S], dO[1], dO[3 % S]};  // (pairs, like q / k / v)
  } else {
    stv<S>(DOb + lane * 16 + c0, dO);
  }
  __syncthreads();  // also: every wave is done reading DR1b
  MHA_STAMP(7);
  // ---- attention backward, the wave's HP heads ----
  Vec<S> dq = vzero<S>(), dk = vzero<S>(), dv = vzero<S>();
  if (S == 4) {
    // Two heads per wave = one packed-fp32 lane pair (v_pk_mul / v_pk_fma: 2 heads per instruction), on rows parked as (h0c0, h1c0,
    // h0c1, h1c1); every element sees the operations of the scalar form below in the same order (same bits).  q4 / k4 / v4 hold
    // their rows in that pair order here.
    const f32x2 qa = {q4[0] * LOG2E, q4[1] * LOG2E}, qb = {q4[2 % S] * LOG2E, q4[3 % S] * LOG2E}, doa = {dO[0], dO[2 % S]}, dob = {dO[1], dO[3 % S]};
    const f32x2 m2 = {mq[0], mq[1 % HP]}, dd2 = {dd[0], dd[1 % HP]};
    f32x2 dqa = {0.f, 0.f}, dqb = {0.f, 0.f};  // phase A: lane = query
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
      const f32x4 kj = ld4(Kb + j * 16 + c0), vj = ld4(Vb + j * 16 + c0);
      const f32x2 ka = {kj[0], kj[1]}, kb = {kj[2], kj[3]}, va = {vj[0], vj[1]}, vb = {vj[2], vj[3]};
      const f32x2 t = __builtin_elementwise_fma(qa, ka, qb * kb) - m2;
      const f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      const f32x2 ds = p * (__builtin_elementwise_fma(doa, va, dob * vb) - dd2);
      dqa = __builtin_elementwise_fma(ds, ka, dqa);
      dqb = __builtin_elementwise_fma(ds, kb, dqb);
    }
    dq[0] = dqa[0] * MHA_SCALE;
    dq[1] = dqb[0] * MHA_SCALE;
    dq[2 % S] = dqa[1] * MHA_SCALE;
    dq[3 % S] = dqb[1] * MHA_SCALE;
    MHA_STAMP(8);
    const f32x2 ka = {k4[0] * LOG2E, k4[1] * LOG2E}, kb = {k4[2 % S] * LOG2E, k4[3 % S] * LOG2E}, va = {v4[0], v4[1]}, vb = {v4[2 % S], v4[3 % S]};
    f32x2 dka = {0.f, 0.f}, dkb = {0.f, 0.f}, dva = {0.f, 0.f}, dvb = {0.f, 0.f};  // phase B: lane = key (ka / kb: the scores' side only)
#pragma unroll 4
    for (int i = 0; i < N; ++i) {
      const f32x4 qi = ld4(Qb + i * 16 + c0), doi = ld4(DOb + i * 16 + c0);
      const f32x2 qia = {qi[0], qi[1]}, qib = {qi[2], qi[3]}, da = {doi[0], doi[1]}, db = {doi[2], doi[3]};
      const f32x2 mi = *reinterpret_cast<const f32x2*>(Mb + i * 8 + HP * w), di = *reinterpret_cast<const f32x2*>(Db + i * 8 + HP * w);
      const f32x2 t = __builtin_elementwise_fma(qia, ka, qib * kb) - mi;
      const f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      dva = __builtin_elementwise_fma(p, da, dva);
      dvb = __builtin_elementwise_fma(p, db, dvb);
      const f32x2 ds = p * (__builtin_elementwise_fma(da, va, db * vb) - di);
      dka = __builtin_elementwise_fma(ds, qia, dka);
      dkb = __builtin_elementwise_fma(ds, qib, dkb);
    }
    dk[0] = dka[0]; dk[1] = dkb[0]; dk[2 % S] = dka[1]; dk[3 % S] = dkb[1];
    dv[0] = dva[0]; dv[1] = dvb[0]; dv[2 % S] = dva[1]; dv[3 % S] = dvb[1];
  } else {
#pragma unroll 4
  for (int j = 0; j < N; ++j) {  // phase A: lane = query
    const Vec<S> kj = ldv<S>(Kb + j * 16 + c0);
    const Vec<S> vj = ldv<S>(Vb + j * 16 + c0);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      const float p = __builtin_amdgcn_exp2f(fmaf(q4[2 * h] * LOG2E, kj[2 * h], (q4[2 * h + 1] * LOG2E) * kj[2 * h + 1]) - mq[h]);
      const float ds = p * (fmaf(dO[2 * h], vj[2 * h], dO[2 * h + 1] * vj[2 * h + 1]) - dd[h]);
      dq[2 * h] = fmaf(ds, kj[2 * h], dq[2 * h]);
      dq[2 * h + 1] = fmaf(ds, kj[2 * h + 1], dq[2 * h + 1]);
    }
  }
#pragma unroll
  for (int r = 0; r < S; ++r) dq[r] *= MHA_SCALE;
  MHA_STAMP(8);
#pragma unroll 4
  for (int i = 0; i < N; ++i) {  // phase B: lane = key
    const Vec<S> qi = ldv<S>(Qb + i * 16 + c0);
    const Vec<S> doi = ldv<S>(DOb + i * 16 + c0);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      const float mi = Mb[i * 8 + HP * w + h], di = Db[i * 8 + HP * w + h];
      const float p = __builtin_amdgcn_exp2f(fmaf(qi[2 * h], k4[2 * h] * LOG2E, qi[2 * h + 1] * (k4[2 * h + 1] * LOG2E)) - mi);
      dv[2 * h] = fmaf(p, doi[2 * h], dv[2 * h]);
      dv[2 * h + 1] = fmaf(p, doi[2 * h + 1], dv[2 * h + 1]);
      const float ds = p * (fmaf(doi[2 * h], v4[2 * h], doi[2 * h + 1] * v4[2 * h + 1]) - di);
      dk[2 * h] = fmaf(ds, qi[2 * h], dk[2 * h]);
      dk[2 * h + 1] = fmaf(ds, qi[2 * h + 1], dk[2 * h + 1]);
    }
  }
  }
  MHA_STAMP(9);
  if (!active) {
    dq = vzero<S>();
    dk = vzero<S>();
    dv = vzero<S>();
  }
  __syncthreads();  // every wave is done with Qb / Kb / Vb / DOb
  stv<S>(DQb + lane * 16 + c0, dq);
  stv<S>(DKb + lane * 16 + c0, dk);
  stv<S>(DVb + lane * 16 + c0, dv);
  __syncthreads();
  MHA_STAMP(10);
  // ---- in-projection: [q;k;v] = Win x + bin ----
  if (S == 4) {  // one matrix per wave on the matrix cores (large batch); the 8-wave form keeps the row slices (latency)
    if (w == 0) wgrad_mfma(DQb, Xb, lane, N, gp + OFF_WIN);
    if (w == 1) wgrad_mfma(DKb, Xb, lane, N, gp + OFF_WIN + 256);
    if (w == 2) wgrad_mfma(DVb, Xb, lane, N, gp + OFF_WIN + 512);
    if (w == 3) wgrad_mfma(DR1b, Ob, lane, N, gp + OFF_WOUT);
  } else {
    wgrad_slice<S>(DQb, Xb, c0, lane, N, gp + OFF_WIN);
    wgrad_slice<S>(DKb, Xb, c0, lane, N, gp + OFF_WIN + 256);
    wgrad_slice<S>(DVb, Xb, c0, lane, N, gp + OFF_WIN + 512);
  }
  bgrad_slice<S>(dq, c0, lane, gp + OFF_BIN);
  bgrad_slice<S>(dk, c0, lane, gp + OFF_BIN + 16);
  bgrad_slice<S>(dv, c0, lane, gp + OFF_BIN + 32);
  MHA_STAMP(11);
  Vec<S> dx = dr1;
  ld_row(DQb + lane * 16, row);
  mvt_slice_acc<S>(Wsh + OFF_WIN, c0, row, dx);
  ld_row(DKb + lane * 16, row);
  mvt_slice_acc<S>(Wsh + OFF_WIN + 256, c0, row, dx);
  ld_row(DVb + lane * 16, row);
  mvt_slice_acc<S>(Wsh + OFF_WIN + 512, c0, row, dx);
  if (active) stv<S>(d.dx + (long)b * d.ldx + lane * 16 + c0, dx);
#ifdef MHA_STAMPS
  MHA_STAMP(12);
  __syncthreads();
  if (tid == 0)
    for (int i = 0; i < 13; ++i) gp[i] = __builtin_bit_cast(float, mha_st[i]);
#endif
}

