// Token-major bodies of the fused Transformer (256 threads = 4 waves per sample), round 6 (modules.py:648-686):
//
//   lane (r = lane & 15, g = lane >> 4) of wave w owns ONE token — 16 blk + r, blk = the wave's block of 16 tokens — and the columns
//   4 g .. 4 g + 3 of every 16-wide vector of that token (= heads 2 g and 2 g + 1 of the attention), in registers, from the first load to
//   the last store.
//
// * Every per-token 16 x 16 product (in-projection x 3, out-projection, the two FFN layers, their transposes in the backward) is FOUR
//   v_mfma_f32_16x16x4_f32 of the wave on its own 16 tokens, computed transposed: D[o][token] = sum_k W[o][k] Y[token][k] with A = W and
//   B = Y^T.  The k index of step s in lane group g is 4 g + s — exactly the column the lane holds in register s — and D leaves lane
//   (r, g) with rows o = 4 g .. 4 g + 3 of token r: the output is in the input's layout.  One LDS read of the weights per product instead
//   of sixteen broadcast reads and 64 FMAs; no LDS exchange of the vectors, no barrier between the stages.
//   (Rounds 3 - 5 ran a column-slice mapping — lane = token, wave = 2 or 4 columns of every vector — that passed every intermediate
//   vector through LDS and a workgroup barrier because each wave needed the other waves' columns: twelve barrier-separated stages whose
//   latency, not their arithmetic, was 60 % of the backward at batch 256.)
// * LayerNorm sums over a token's 16 columns = 4 in the lane + the 4 lanes (r, 0..3): v_permlane16_swap / v_permlane32_swap (gfx950), two
//   instructions and two adds, every lane of the token gets the same bits.
// * Attention (head_dim 2: nothing for the matrix cores): K / V rows (backward: Q, dO, (m', D) too) in LDS in natural order; inside the
//   loops a lane pair (r, r ^ 1) trades heads so that each lane runs ONE head of TWO tokens as packed fp32 (TokPairs).
// * Backward: q, k, v, both LayerNorm x-hats, the LayerNorm-1 output and the FFN hidden layer are rebuilt from x, the saved attention
//   output and the saved LayerNorm statistics — six more products per token, the forward's own instructions on the same operands (the
//   same bits) instead of 112 more floats per token through HBM in both directions.  Weight gradients: 4 MFMAs per matrix with k = the
//   wave's own tokens (tok_wgrad), bias-like gradients: DPP sums over the 16 lanes of a row; per-BLOCK partials in LDS, summed in block
//   order at the end (deterministic whatever the rotation of blocks over waves).
// * A wave whose block holds no token (N <= 48) skips everything but the parameter staging and the barriers.
// Barriers: forward 2, backward 4.  LDS: forward 15 KB, backward 39.5 KB.
#pragma once
#include "attention_body.h"

__device__ __forceinline__ float tok_row16_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane of the row
  v += dpp_mov<0xb1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4e>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8
  return v;
}
__device__ __forceinline__ float tok_rows4_sum(float v) {  // sum over lanes r, r + 16, r + 32, r + 48, in each of them (same bits)
  // (inline asm: handed the same value twice, the builtin is emitted with ONE register as both operands — a swap of the register with
  // itself, rows exchanged in place — instead of a copy; two "+v" operands are two registers.  tools/micro/permlane_probe.hip.)
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // a = [row0, row0, row2, row2], b = [row1, row1, row3, row3]
  a += b;
  b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // a = [lower, lower], b = [upper, upper]
  return a + b;
}
// out[token r][4 g + i] = acc[i] + sum_k Wl[(4 g + i) * 16 + k] * y[token r][k]   (Wl: LDS, row = output; y: the lane's 4 columns)
// (two accumulators — k = 4 g + {0, 1} on top of `acc`, {2, 3} from zero — so that a product is two dependent MFMAs and an add deep instead
// of four: these products sit in dependent chains of three to five per body; forward and recomputation share this one function)
__device__ __forceinline__ f32x4 tok_mma(const f32x4 a, const f32x4 y, f32x4 acc) {  // the lane's piece of the matrix in registers
  f32x4 hi = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], y[0], acc, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], y[2], hi, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], y[1], acc, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], y[3], hi, 0, 0, 0);
  return acc + hi;
}
__device__ __forceinline__ f32x4 tok_mm(const float* Wl, int r, int c0, const f32x4 y, f32x4 acc) { return tok_mma(ld4(Wl + r * 16 + c0), y, acc); }
__device__ __forceinline__ float sum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// The attention loops run on PAIRS OF TOKENS: the two lanes (r, g), (r ^ 1, g) own tokens t0 = r & ~1, t1 = t0 + 1 and heads 2 g, 2 g + 1;
// inside the loops lane r & 1 = hs takes head 2 g + hs of BOTH tokens — one packed-fp32 pair per component — so that a key (or query) row
// costs the lane 8 bytes of LDS instead of 16 for the same two (token, head) products.  With the row of both heads per lane the loops
// ran at the LDS port's 128 bytes per clock (four waves x 64 lanes x 16 or 32 bytes per key: 40 / 69 clocks per key measured in the two
// forward passes against 16 / 56 of vector issue).  Exchange = one quad_perm DPP move per float, in and out.
struct TokPairs { f32x2 c0, c1; };  // components 0 and 1 of one head, tokens (t0, t1)
__device__ __forceinline__ TokPairs tok_to_pairs(const f32x4 v, bool hs) {  // v: (h0c0, h0c1, h1c0, h1c1) of the lane's own token
  const float k0 = hs ? v[2] : v[0], k1 = hs ? v[3] : v[1];  // own token, head hs
  const float r0 = dpp_mov<0xb1>(hs ? v[0] : v[2]), r1 = dpp_mov<0xb1>(hs ? v[1] : v[3]);  // the partner's token, head hs
  TokPairs p;
  p.c0 = hs ? (f32x2){r0, k0} : (f32x2){k0, r0};
  p.c1 = hs ? (f32x2){r1, k1} : (f32x2){k1, r1};
  return p;
}
__device__ __forceinline__ f32x4 tok_from_pairs(const f32x2 c0, const f32x2 c1, bool hs) {  // back: both heads of the lane's own token
  const float k0 = hs ? c0[1] : c0[0], k1 = hs ? c1[1] : c1[0];
  const float r0 = dpp_mov<0xb1>(hs ? c0[0] : c0[1]), r1 = dpp_mov<0xb1>(hs ? c1[0] : c1[1]);
  return hs ? (f32x4){r0, r1, k0, k1} : (f32x4){k0, k1, r0, r1};
}
__device__ __forceinline__ f32x2 tok_swap_pair(const f32x2 v, bool hs) {  // (head 2 g, head 2 g + 1) of the own token <-> (t0, t1) of head hs: an involution
  const float k = hs ? v[1] : v[0], r = dpp_mov<0xb1>(hs ? v[0] : v[1]);
  return hs ? (f32x2){r, k} : (f32x2){k, r};
}

__device__ __forceinline__ void tok_ln_stats(const f32x4 v, float& mu, float& rstd) {
  mu = tok_rows4_sum(sum4(v)) * (1.f / 16.f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) q += (v[e] - mu) * (v[e] - mu);
  rstd = 1.f / sqrtf(tok_rows4_sum(q) * (1.f / 16.f) + 1e-5f);
}

#define MHA_TOK_FWD_LDS_FLOATS (NASREC_MHA_PARAMS + 2 * MHA_N * 16)
// Which block of 16 tokens wave w of sample b owns: rotated by the sample index, so that with N <= 48 the waves WITHOUT tokens (they only
// help park the parameters and meet the barriers) are not the same wave slot — hence the same SIMD — in every workgroup of the launch.
__device__ __forceinline__ int tok_block(int w, int b) { return (w + b) & 3; }

__device__ __forceinline__ void mha_fwd_tok(const nasrec_mha_desc_t& d, const int b, float* lds) {
  constexpr int NT = 256;
  float* Wsh = lds;
  float* Ks = Wsh + NASREC_MHA_PARAMS;
  float* Vs = Ks + MHA_N * 16;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4, c0 = 4 * g, tok = 16 * tok_block(w, b) + r;
  const int N = d.N;
  const bool active = tok < N, wave_active = tok - r < N;  // (wave-uniform: the wave's block holds tokens)
  // supernet token mask (modules.py:678-686): output tokens >= dims_in_use are zero.  They still act as KEYS (their k / v rows come from
  // whatever x holds there), but nothing of them as queries survives: a wave whose whole block is masked parks its k / v rows and stores zeros
  const int qn = d.dims_in_use >= 0 ? min(N, d.dims_in_use) : N;
  const bool q_active = tok - r < qn;
  const bool saving = d.saved != nullptr;
#ifdef MHA_STAMPS
  unsigned mha_st[16];
#endif
  MHA_STAMP(0);
  ParamPieces<NT> pp;
  stage_params_load<NT>(d, tid, pp);
  f32x4 x4 = {0.f, 0.f, 0.f, 0.f};
  if (wave_active) x4 = ld4(d.x + (long)b * d.ldx + min(tok, N - 1) * 16 + c0);
  stage_params_store<NT>(Wsh, tid, pp);
  if (!active) x4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  MHA_STAMP(1);
  // in-projection
  f32x4 q4 = x4;
  if (wave_active) {
    q4 = tok_mm(Wsh + OFF_WIN, r, c0, x4, ld4(Wsh + OFF_BIN + c0)) * MHA_SCALE;
    const f32x4 k4 = tok_mm(Wsh + OFF_WIN + 256, r, c0, x4, ld4(Wsh + OFF_BIN + 16 + c0));
    const f32x4 v4 = tok_mm(Wsh + OFF_WIN + 512, r, c0, x4, ld4(Wsh + OFF_BIN + 32 + c0));
    *reinterpret_cast<f32x4*>(Ks + tok * 16 + c0) = k4;
    *reinterpret_cast<f32x4*>(Vs + tok * 16 + c0) = v4;
  }
  __syncthreads();
  MHA_STAMP(2);
  if (wave_active && !q_active) {
    if (active) *reinterpret_cast<f32x4*>(d.out + (long)b * d.ldo + (long)tok * 16 + c0) = (f32x4){0.f, 0.f, 0.f, 0.f};
  } else if (wave_active) {
    const long po = (long)tok * 16 + c0;  // the lane's 16-byte piece of a [token][16] plane
    // attention: head 2 g + hs of the lane pair's two tokens as queries (see TokPairs); scores in log2 units
    constexpr float LOG2E = 1.44269504088896340736f;
    const bool hs = r & 1;
    const int h2 = c0 + 2 * (r & 1);  // the head's two columns in a row
    const TokPairs Q = tok_to_pairs(q4 * LOG2E, hs);
    f32x2 m2 = {-INFINITY, -INFINITY};
#pragma unroll 8
    for (int j = 0; j < N; ++j) {
      const f32x2 kj = *reinterpret_cast<const f32x2*>(Ks + j * 16 + h2);
      const f32x2 s2 = Q.c0 * kj[0] + Q.c1 * kj[1];
      m2[0] = fmaxf(m2[0], s2[0]);
      m2[1] = fmaxf(m2[1], s2[1]);
    }
    MHA_STAMP(3);
    f32x2 l2 = {0.f, 0.f}, oa = {0.f, 0.f}, ob = {0.f, 0.f};
#pragma unroll 8
    for (int j = 0; j < N; ++j) {
      const f32x2 kj = *reinterpret_cast<const f32x2*>(Ks + j * 16 + h2);
      const f32x2 vj = *reinterpret_cast<const f32x2*>(Vs + j * 16 + h2);
      const f32x2 t2 = Q.c0 * kj[0] + Q.c1 * kj[1] - m2;
      const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
      l2 = l2 + p2;
      oa = p2 * vj[0] + oa;
      ob = p2 * vj[1] + ob;
    }
    const f32x2 lip = {1.f / l2[0], 1.f / l2[1]};
    const f32x2 mx = tok_swap_pair(m2 * (1.f / LOG2E), hs), li = tok_swap_pair(lip, hs);  // per head of the own token; the backward works in natural units
    const f32x4 o4 = tok_from_pairs(oa * lip, ob * lip, hs);
    MHA_STAMP(4);
    if (saving && active) {
      *reinterpret_cast<f32x4*>(sv_plane(d.saved, b, N, SV_O) + po) = o4;
      float* mp = sv_plane(d.saved, b, N, SV_M) + (long)tok * 16 + 2 * g;
      *reinterpret_cast<f32x2*>(mp) = mx;
      *reinterpret_cast<f32x2*>(mp + 8) = li;
    }
    MHA_STAMP(5);
    // out-projection + residual + LayerNorm 1
    const f32x4 r1 = tok_mm(Wsh + OFF_WOUT, r, c0, o4, ld4(Wsh + OFF_BOUT + c0)) + x4;
    float mu1, rstd1;
    tok_ln_stats(r1, mu1, rstd1);
    const f32x4 xh1 = (r1 - mu1) * rstd1;
    const f32x4 h1 = xh1 * ld4(Wsh + OFF_L1W + c0) + ld4(Wsh + OFF_L1B + c0);
    MHA_STAMP(6);
    // FFN
    f32x4 f1 = tok_mm(Wsh + OFF_W1, r, c0, h1, ld4(Wsh + OFF_C1 + c0));
#pragma unroll
    for (int e = 0; e < 4; ++e) f1[e] = fmaxf(f1[e], 0.f);
    const f32x4 r2 = tok_mm(Wsh + OFF_W2, r, c0, f1, ld4(Wsh + OFF_C2 + c0)) + h1;
    MHA_STAMP(7);
    float mu2, rstd2;
    tok_ln_stats(r2, mu2, rstd2);
    MHA_STAMP(8);
    const f32x4 xh2 = (r2 - mu2) * rstd2;
    f32x4 out = xh2 * ld4(Wsh + OFF_L2W + c0) + ld4(Wsh + OFF_L2B + c0);
    if (d.dims_in_use >= 0 && tok >= d.dims_in_use) out = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (active) {
      *reinterpret_cast<f32x4*>(d.out + (long)b * d.ldo + po) = out;
      if (saving && g == 0) *reinterpret_cast<f32x4*>(sv_plane(d.saved, b, N, SV_STAT) + tok * 4) = (f32x4){rstd1, rstd2, mu1, mu2};
    }
  }
#ifdef MHA_STAMPS
  MHA_STAMP(9);
  __syncthreads();
  if (tid == 0)  // (wave 0 owns block b & 3: tools/mha_stamps.py reads the samples where that block holds tokens)
    for (int i = 0; i < 10; ++i) d.out[(long)b * d.ldo + i] = __builtin_bit_cast(float, mha_st[i]);
#endif
}

// ---- backward -------------------------------------------------------------------------------------------------------------------------------
// LDS floats: transposed matrices 6 x 16 rows of 20 | the twelve-minus-matrices parameter vectors 160 | K V Q dO rows 4 x 1024 | m' and D per (token, head) 2 x 512 |
// per-wave weight-gradient operand planes 4 x 2 x 256 | per-block token sums 4 x 160
#define MHA_TOK_WT 0
#define MHA_TOK_WLD 20   // row stride of a parked matrix: rows 16 banks apart for the column reads, 16-byte aligned for the row reads
#define MHA_TOK_WSZ (16 * MHA_TOK_WLD)
#define MHA_TOK_VEC (6 * MHA_TOK_WSZ)  // bin 48 | bout 16 | l1w 16 | l1b 16 | c1 16 | c2 16 | l2w 16 | l2b 16
#define MHA_TOK_ROWS (MHA_TOK_VEC + 160)
#define MHA_TOK_MD (MHA_TOK_ROWS + 4 * MHA_N * 16)
#define MHA_TOK_SCR (MHA_TOK_MD + 2 * MHA_N * 8)  // per wave: two operand planes [16 tokens][16] of its weight-gradient products
// (after the loops, rows + m' / D + the first half of these planes = the 4 x 6 x 256 per-block weight-gradient partials; 39.5 KB in all:
// four workgroups per CU)
#define MHA_TOK_BP (MHA_TOK_SCR + 4 * 512)
#define MHA_TOK_BWD_LDS_FLOATS (MHA_TOK_BP + 4 * 160)

// The six 16 x 16 matrices parked TRANSPOSED (Wt[m][i][o] = W_m[o][i], rows of MHA_TOK_WLD floats): a product with W^T reads the
// lane's operand as one 16-byte row piece (tok_mm), a product with W as four dwords down a column (tok_mm_col) — 16 consecutive floats
// per lane group, the groups 16 banks apart: conflict-free both ways, as is this store.  The vectors go to their own 160 floats.
template <int NT>
__device__ __forceinline__ void stage_params_store_t(float* lds, int tid, const ParamPieces<NT>& pp) {
#pragma unroll
  for (int k = 0; k < ParamPieces<NT>::PER; ++k) {
    const int piece = tid + k * NT;
    if (piece >= ParamPieces<NT>::PIECES) continue;
    const int off = 4 * piece;
    int m = -1, base = 0, vec = -1;
    if (off < OFF_BIN) {
      m = off >> 8;
      base = m << 8;
    } else if (off < OFF_WOUT) {
      vec = off - OFF_BIN;
    } else if (off < OFF_BOUT) {
      m = 3;
      base = OFF_WOUT;
    } else if (off < OFF_W1) {
      vec = 48 + (off - OFF_BOUT);  // bout, l1w, l1b
    } else if (off < OFF_C1) {
      m = 4;
      base = OFF_W1;
    } else if (off < OFF_W2) {
      vec = 96 + (off - OFF_C1);  // c1
    } else if (off < OFF_C2) {
      m = 5;
      base = OFF_W2;
    } else {
      vec = 112 + (off - OFF_C2);  // c2, l2w, l2b
    }
    if (m >= 0) {
      const int o = (off - base) >> 4, i0 = (off - base) & 15;
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[MHA_TOK_WT + m * MHA_TOK_WSZ + (i0 + e) * MHA_TOK_WLD + o] = pp.v[k][e];
    } else {
      *reinterpret_cast<f32x4*>(lds + MHA_TOK_VEC + vec) = pp.v[k];
    }
  }
}
// products against a parked matrix: with W^T (the backward's) and with W (the recomputed forward)
__device__ __forceinline__ f32x4 tok_mm_t(const float* Wt, int r, int c0, const f32x4 y, f32x4 acc) {
  return tok_mma(ld4(Wt + r * MHA_TOK_WLD + c0), y, acc);
}
__device__ __forceinline__ f32x4 tok_mm_col(const float* Wt, int r, int c0, const f32x4 y, f32x4 acc) {
  const float* p = Wt + c0 * MHA_TOK_WLD + r;
  return tok_mma((f32x4){p[0], p[MHA_TOK_WLD], p[2 * MHA_TOK_WLD], p[3 * MHA_TOK_WLD]}, y, acc);
}

// per-wave sum over the wave's 16 tokens of the lane's 4 columns -> BP[w][vec][16]
__device__ __forceinline__ void tok_bias_partial(float* bp, int vec, int r, int c0, const f32x4 v) {
  f32x4 s;
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = tok_row16_sum(v[e]);
  if (r == 0) *reinterpret_cast<f32x4*>(bp + vec * 16 + c0) = s;
}

// The wave's share of a weight gradient dW[o][i] = sum_tok G[tok][o] V[tok][i]: its own 16 tokens, 4 MFMAs with k = token.  The lane's
// pieces of G and V (token r, columns 4 g ..) go to the wave's two scratch planes and come back transposed — lane (r, g) feeds
// A(o = r, k = 4 s + g) = G[4 s + g][r] and B(k, i = r) = V[4 s + g][r]: 16 consecutive floats per lane group, the groups 16 banks apart.
// LDS operations of one wave execute in order: no barrier.  Result: D[o = 4 g + e][i = r] in register e.  Rows of tokens >= N are zero
// in G (every gradient of an inactive lane is zero), so they add nothing.
__device__ __forceinline__ f32x4 tok_wgrad(float* Gs, float* Vs, int r, int g, const f32x4 gv, const f32x4 vv, bool write_v) {
  *reinterpret_cast<f32x4*>(Gs + r * 16 + 4 * g) = gv;
  if (write_v) *reinterpret_cast<f32x4*>(Vs + r * 16 + 4 * g) = vv;
  f32x4 a, b;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a[s] = Gs[(4 * s + g) * 16 + r];
    b[s] = Vs[(4 * s + g) * 16 + r];
  }
  return tok_mma(a, b, (f32x4){0.f, 0.f, 0.f, 0.f});
}

__device__ __forceinline__ void mha_bwd_tok(const nasrec_mha_desc_t& d, const int b, float* lds) {
  constexpr int NT = 256;
  float* Wt = lds + MHA_TOK_WT;
  float* Kb = lds + MHA_TOK_ROWS;
  float* Vb = Kb + MHA_N * 16;
  float* Qb = Vb + MHA_N * 16;
  float* DOb = Qb + MHA_N * 16;
  float* MDb = lds + MHA_TOK_MD;  // [token][16]: (m', D) of head h at columns 2 h, 2 h + 1
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4, c0 = 4 * g, blk = tok_block(w, b), tok = 16 * blk + r;
  float* bp = lds + MHA_TOK_BP + blk * 160;   // per-BLOCK partials (summed in block order below: the same bits whatever the rotation)
  float* Gs = lds + MHA_TOK_SCR + w * 512;    // the wave's operand planes of its weight-gradient products
  float* Vs = Gs + 256;
  const int N = d.N;
  const bool active = tok < N, wave_active = 16 * blk < N;
  // supernet token mask: d out is zero for tokens >= dims_in_use, hence every gradient of theirs AS QUERIES is exactly zero (d r2, d f1, d h1,
  // d r1, d O, D, d q); as KEYS they still receive d k / d v from the unmasked queries.  A wave whose whole block is masked recomputes its k / v
  // rows, runs the key-side loop only, and that loop walks the unmasked queries only (the others would add p * 0).
  const int qn = d.dims_in_use >= 0 ? min(N, d.dims_in_use) : N;
  const bool q_active = 16 * blk < qn;
  float* gp = d.dparams_partial + (long)b * (d.partial_ld > 0 ? d.partial_ld : NASREC_MHA_PARAMS);
#ifdef MHA_STAMPS
  unsigned mha_st[16];
#endif
  MHA_STAMP(0);
  // ---- everything the lane needs of its token, one round trip: x, dout, attention output and statistics (and its share of the parameters) ----
  const int tl = min(tok, N - 1);
  const long po = (long)tl * 16 + c0;
  constexpr float LOG2E = 1.44269504088896340736f;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  ParamPieces<NT> pp;
  stage_params_load<NT>(d, tid, pp);
  f32x4 st4 = z4, dout = z4, x4 = z4, o4 = z4;
  f32x2 mq = {0.f, 0.f};
  if (wave_active && !q_active) x4 = ld4(d.x + (long)b * d.ldx + po);
  if (q_active) {
    st4 = ld4(sv_plane(d.saved, b, N, SV_STAT) + tl * 4);
    const float* mp = sv_plane(d.saved, b, N, SV_M) + (long)tl * 16 + 2 * g;
    const f32x2 mx = *reinterpret_cast<const f32x2*>(mp), li = *reinterpret_cast<const f32x2*>(mp + 8);
    dout = ld4(d.dout + (long)b * d.ldo + po);
    x4 = ld4(d.x + (long)b * d.ldx + po);
    o4 = ld4(sv_plane(d.saved, b, N, SV_O) + po);
    mq = (f32x2){(mx[0] - __logf(li[0])) * LOG2E, (mx[1] - __logf(li[1])) * LOG2E};  // m' = (max + ln sum) log2 e per (token, head)
  }
  stage_params_store_t<NT>(lds, tid, pp);
  if (!active) x4 = o4 = dout = z4;
  if (d.dims_in_use >= 0 && tok >= d.dims_in_use) dout = z4;
  __syncthreads();  // the parameters are parked
  MHA_STAMP(1);
  f32x4 q4 = z4, k4 = z4, v4 = z4, dr1 = z4, dO = z4, gW2 = z4, gW1 = z4, gWo = z4, gWq = z4, gWk = z4, gWv = z4;
  f32x2 dd2 = {0.f, 0.f};
  if (wave_active && !q_active) {  // a masked block: its k / v rows (it is a set of keys), nothing else
    const float* VEC = lds + MHA_TOK_VEC;
    k4 = tok_mm_col(Wt + MHA_TOK_WSZ, r, c0, x4, ld4(VEC + 16 + c0));
    v4 = tok_mm_col(Wt + 2 * MHA_TOK_WSZ, r, c0, x4, ld4(VEC + 32 + c0));
    *reinterpret_cast<f32x4*>(Kb + tok * 16 + c0) = k4;
    *reinterpret_cast<f32x4*>(Vb + tok * 16 + c0) = v4;
#pragma unroll
    for (int vec = 0; vec < 7; ++vec)
      if (r == 0) *reinterpret_cast<f32x4*>(bp + vec * 16 + c0) = z4;  // (the block's token sums of everything but d q / d k / d v: zeros)
  }
  if (q_active) {
    // ---- the forward again, from x and o (mha_fwd_tok's instructions on the same operands) ----
    const float* VEC = lds + MHA_TOK_VEC;
    const float rstd1 = st4[0], rstd2 = st4[1];
    const f32x4 l1w = ld4(VEC + 64 + c0), l2w = ld4(VEC + 128 + c0);
    q4 = tok_mm_col(Wt, r, c0, x4, ld4(VEC + c0)) * MHA_SCALE;
    k4 = tok_mm_col(Wt + MHA_TOK_WSZ, r, c0, x4, ld4(VEC + 16 + c0));
    v4 = tok_mm_col(Wt + 2 * MHA_TOK_WSZ, r, c0, x4, ld4(VEC + 32 + c0));
    const f32x4 xh1 = (tok_mm_col(Wt + 3 * MHA_TOK_WSZ, r, c0, o4, ld4(VEC + 48 + c0)) + x4 - st4[2]) * rstd1;
    const f32x4 h1 = xh1 * l1w + ld4(VEC + 80 + c0);
    f32x4 f1 = tok_mm_col(Wt + 4 * MHA_TOK_WSZ, r, c0, h1, ld4(VEC + 96 + c0));
#pragma unroll
    for (int e = 0; e < 4; ++e) f1[e] = fmaxf(f1[e], 0.f);
    const f32x4 xh2 = (tok_mm_col(Wt + 5 * MHA_TOK_WSZ, r, c0, f1, ld4(VEC + 112 + c0)) + h1 - st4[3]) * rstd2;
    *reinterpret_cast<f32x4*>(Qb + tok * 16 + c0) = q4;
    *reinterpret_cast<f32x4*>(Kb + tok * 16 + c0) = k4;
    *reinterpret_cast<f32x4*>(Vb + tok * 16 + c0) = v4;
    MHA_STAMP(2);
    // ---- LayerNorm 2 ----
    tok_bias_partial(bp, 0, r, c0, dout * xh2);
    tok_bias_partial(bp, 1, r, c0, dout);
    f32x4 gw = dout * l2w;
    float ca = tok_rows4_sum(sum4(gw)) * (1.f / 16.f), cb = tok_rows4_sum(sum4(gw * xh2)) * (1.f / 16.f);
    const f32x4 dr2 = (gw - ca - xh2 * cb) * rstd2;
    gW2 = tok_wgrad(Gs, Vs, r, g, dr2, f1, true);
    tok_bias_partial(bp, 2, r, c0, dr2);
    MHA_STAMP(3);
    // ---- FFN 2, FFN 1 ----
    f32x4 df1 = tok_mm_t(Wt + 5 * MHA_TOK_WSZ, r, c0, dr2, z4);
#pragma unroll
    for (int e = 0; e < 4; ++e) df1[e] = f1[e] > 0.f ? df1[e] : 0.f;
    gW1 = tok_wgrad(Gs, Vs, r, g, df1, h1, true);
    tok_bias_partial(bp, 3, r, c0, df1);
    MHA_STAMP(4);
    const f32x4 dh1 = tok_mm_t(Wt + 4 * MHA_TOK_WSZ, r, c0, df1, dr2);
    MHA_STAMP(5);
    // ---- LayerNorm 1 ----
    tok_bias_partial(bp, 4, r, c0, dh1 * xh1);
    tok_bias_partial(bp, 5, r, c0, dh1);
    gw = dh1 * l1w;
    ca = tok_rows4_sum(sum4(gw)) * (1.f / 16.f);
    cb = tok_rows4_sum(sum4(gw * xh1)) * (1.f / 16.f);
    dr1 = (gw - ca - xh1 * cb) * rstd1;
    gWo = tok_wgrad(Gs, Vs, r, g, dr1, o4, true);
    tok_bias_partial(bp, 6, r, c0, dr1);
    MHA_STAMP(6);
    // ---- out-projection ----
    dO = tok_mm_t(Wt + 3 * MHA_TOK_WSZ, r, c0, dr1, z4);
    dd2 = (f32x2){fmaf(dO[0], o4[0], dO[1] * o4[1]), fmaf(dO[2], o4[2], dO[3] * o4[3])};
    *reinterpret_cast<f32x4*>(MDb + tok * 16 + c0) = (f32x4){mq[0], dd2[0], mq[1], dd2[1]};  // (m', D) per head
    *reinterpret_cast<f32x4*>(DOb + tok * 16 + c0) = dO;
  }
  __syncthreads();  // every token's K / V / Q / dO rows, m' and D are in LDS
  MHA_STAMP(7);
  if (wave_active) {
    // ---- attention backward on token pairs (see TokPairs): head 2 g + hs of tokens t0, t1; one exponential per probability (exp2(s' - m')) ----
    f32x4 dq = z4, dk, dv;
    const bool hs = r & 1;
    const int h2 = c0 + 2 * (r & 1);
    if (q_active) {
      const TokPairs Q = tok_to_pairs(q4 * LOG2E, hs), DO = tok_to_pairs(dO, hs);
      const f32x2 M2 = tok_swap_pair(mq, hs), DD = tok_swap_pair(dd2, hs);
      f32x2 dq0 = {0.f, 0.f}, dq1 = {0.f, 0.f};  // phase A: the pair's tokens as queries
#pragma unroll 4
      for (int j = 0; j < N; ++j) {
        const f32x2 kj = *reinterpret_cast<const f32x2*>(Kb + j * 16 + h2), vj = *reinterpret_cast<const f32x2*>(Vb + j * 16 + h2);
        const f32x2 t = Q.c0 * kj[0] + Q.c1 * kj[1] - M2;
        const f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        const f32x2 ds = p * (DO.c0 * vj[0] + DO.c1 * vj[1] - DD);
        dq0 = ds * kj[0] + dq0;
        dq1 = ds * kj[1] + dq1;
      }
      dq = tok_from_pairs(dq0 * MHA_SCALE, dq1 * MHA_SCALE, hs);
    }
    MHA_STAMP(8);
    {
      const TokPairs K = tok_to_pairs(k4 * LOG2E, hs), V = tok_to_pairs(v4, hs);
      f32x2 dk0 = {0.f, 0.f}, dk1 = {0.f, 0.f}, dv0 = {0.f, 0.f}, dv1 = {0.f, 0.f};  // phase B: the pair's tokens as keys
#pragma unroll 4
      for (int i = 0; i < qn; ++i) {  // (queries behind the token mask have d O = D = 0: they would add nothing)
        const f32x2 qi = *reinterpret_cast<const f32x2*>(Qb + i * 16 + h2), doi = *reinterpret_cast<const f32x2*>(DOb + i * 16 + h2);
        const f32x2 md = *reinterpret_cast<const f32x2*>(MDb + i * 16 + h2);
        const f32x2 t = K.c0 * qi[0] + K.c1 * qi[1] - md[0];
        const f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        dv0 = p * doi[0] + dv0;
        dv1 = p * doi[1] + dv1;
        const f32x2 ds = p * (V.c0 * doi[0] + V.c1 * doi[1] - md[1]);
        dk0 = ds * qi[0] + dk0;
        dk1 = ds * qi[1] + dk1;
      }
      dk = tok_from_pairs(dk0, dk1, hs);
      dv = tok_from_pairs(dv0, dv1, hs);
    }
    MHA_STAMP(9);
    if (!active) dq = dk = dv = z4;
    tok_bias_partial(bp, 7, r, c0, dq);
    tok_bias_partial(bp, 8, r, c0, dk);
    tok_bias_partial(bp, 9, r, c0, dv);
    // ---- in-projection: weight gradients (the wave's tokens), dx = dr1 + Win^T [dq; dk; dv] ----
    gWq = tok_wgrad(Gs, Vs, r, g, dq, x4, true);
    gWk = tok_wgrad(Gs, Vs, r, g, dk, x4, false);
    gWv = tok_wgrad(Gs, Vs, r, g, dv, x4, false);
    const f32x4 dx = (tok_mm_t(Wt, r, c0, dq, dr1) + tok_mm_t(Wt + MHA_TOK_WSZ, r, c0, dk, z4)) + tok_mm_t(Wt + 2 * MHA_TOK_WSZ, r, c0, dv, z4);
    if (active) *reinterpret_cast<f32x4*>(d.dx + (long)b * d.ldx + (long)tok * 16 + c0) = dx;
  }
  MHA_STAMP(10);
  __syncthreads();  // every wave is done with the K / V / Q / dO rows, m' and D: their LDS takes the per-block partials [block][matrix][i][o]
  if (wave_active) {
    float* P = lds + MHA_TOK_ROWS + blk * 1536 + r * 16 + c0;
    *reinterpret_cast<f32x4*>(P) = gWq;
    *reinterpret_cast<f32x4*>(P + 256) = gWk;
    *reinterpret_cast<f32x4*>(P + 512) = gWv;
    *reinterpret_cast<f32x4*>(P + 768) = gWo;
    *reinterpret_cast<f32x4*>(P + 1024) = gW1;
    *reinterpret_cast<f32x4*>(P + 1280) = gW2;
  }
  __syncthreads();
  MHA_STAMP(11);
  // ---- the sample's parameter gradients: the blocks' partials in block order; thread t owns entry (o = t / 16, i = t % 16) of every matrix ----
  {
    const float* P = lds + MHA_TOK_ROWS + (tid & 15) * 16 + (tid >> 4);
    const int nb = (N + 15) >> 4;  // blocks that hold tokens (the others parked nothing): every read of a block in flight at once
    const float* BP = lds + MHA_TOK_BP + min(tid, 159);
    float v[6][4], bs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int m = 0; m < 6; ++m) v[m][k] = 0.f;
      bs[k] = 0.f;
      if (k < nb) {
#pragma unroll
        for (int m = 0; m < 6; ++m) v[m][k] = P[k * 1536 + m * 256];
        bs[k] = BP[k * 160];
      }
    }
    const float b0 = bs[0], b1 = bs[1], b2 = bs[2], b3 = bs[3];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int dst = m < 3 ? OFF_WIN + 256 * m : m == 3 ? OFF_WOUT : m == 4 ? OFF_W1 : OFF_W2;
      gp[dst + tid] = (v[m][0] + v[m][1]) + (v[m][2] + v[m][3]);
    }
    if (tid < 160) {
      const int vec = tid >> 4;
      const int dst = vec == 0 ? OFF_L2W : vec == 1 ? OFF_L2B : vec == 2 ? OFF_C2 : vec == 3 ? OFF_C1 : vec == 4 ? OFF_L1W : vec == 5 ? OFF_L1B
                      : vec == 6 ? OFF_BOUT : OFF_BIN + 16 * (vec - 7);
      gp[dst + (tid & 15)] = (b0 + b1) + (b2 + b3);
    }
  }
#ifdef MHA_STAMPS
  MHA_STAMP(12);
  __syncthreads();
  if (tid == 0)
    for (int i = 0; i < 13; ++i) gp[i] = __builtin_bit_cast(float, mha_st[i]);
#endif
}
