// Token-axis Linear at large batch (modules.py:222-234, 358-361, 648-650): out[b, i, e] = sum_k A(i, k) x[b, k, e] over the 16 embedding
// columns e of every sample b — the forward product (A = W, binding KC / TOKR / TOKJ, K-concatenated input segments) and the input
// gradient (A = W^T, binding RC / TOKR / TOKJ, a batch of independent problems).  M = output tokens <= 80, K = input tokens, and
// B * 16 columns: a memory-bound stream (per sample (K + M) * 64 bytes) with as many flops as the fp32 MFMA does in the same time.
//
// The general GEMM template tiles this as 64 x 64 (i, (b, e)) blocks: every workgroup re-reads the weights from L2, stages 18 KB
// through LDS with dword loads and runs two k-tiles — 52-78 us per launch at B = 4096 whatever its size, 3-7x its HBM bytes.
// Here the roles are turned round:
//   * the WEIGHTS live in LDS for the whole workgroup (k-major [k][MP], MP = 16 / 48 / 80 so that the four k-groups of an MFMA
//     operand read land on disjoint banks), staged once per workgroup of 16 wavefronts;
//   * a wavefront owns one SAMPLE at a time: x[b] is a contiguous [K, 16] block, and the B operand of v_mfma_f32_16x16x4_f32
//     (lane = (k-group g, column e)) for k-step kk is exactly the 64 consecutive floats x[b][4 kk .. 4 kk + 3][0 .. 15] —
//     one fully coalesced 256-byte load per MFMA k-step, straight to registers (buffer loads bounded by the sample's K * 64
//     bytes: the ragged last k-step reads zeros), no LDS round trip for the streamed operand;
//   * the A operand of each MFMA is one ds_read_b32; the accumulators D[i = 4 g + r][e] of the M / 16 row blocks go straight
//     to out[b][i][e] (64-byte segments) through the same epilogue as the general kernel (bias on rows, activation, prefix
//     mask on rows, saved pre-activation, accumulation).
// Exact fp32 FMA chains; the k order inside a sample is the natural one, so results do not depend on the launch geometry.
#include "gemm_tile.h"

#define TL_WAVES 16
#define TL_CHUNK 8        // k-steps (of 4 k) loaded before their MFMAs
#define TL_MAX_LDS 147456  // bytes of staged weights per workgroup (one 16-wave workgroup per CU; 160 KB LDS)
#define TL_BIAS_FLOATS 80  // the row biases sit in front of the weights (M <= 80)

template <int RB>
struct TlPad {
  static constexpr int v = RB == 1 ? 16 : (RB <= 3 ? 48 : 80);
};

template <int AM, int RB>
__global__ __launch_bounds__(1024) void token_linear_kernel(const nasrec_gemm_desc_t d, int wgs) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float* const Bl = lds_all;                   // row biases (or zeros)
  float* const Wl = lds_all + TL_BIAS_FLOATS;  // weights
  constexpr int MP = TlPad<RB>::v;
  // (the wave index as a SCALAR: the buffer resources below are built from it, and a resource the compiler takes for lane-dependent is
  // wrapped in a readfirstlane loop around every load)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = d.zmode ? (int)blockIdx.x / wgs : 0;
  const int wg = (int)blockIdx.x - z * wgs;
  const int s_lo = d.zmode ? z : 0, s_hi = d.zmode ? z + 1 : d.nseg;
  const nasrec_gemm_seg_t& s0 = d.seg[s_lo];
  const int M = s0.M, Bs = s0.N >> 4;

  // ---- weights -> LDS, k-major, zero-padded to MP rows and to whole k-steps ---------------------------------------------------
  int kbase = 0;
  for (int s = s_lo; s < s_hi; ++s) {
    const nasrec_gemm_seg_t& sg = d.seg[s];
    if (!sg.A || sg.K <= 0) continue;
    const int Kp = (sg.K + 3) & ~3;
    const int total = Kp * MP;
    for (int idx = tid; idx < total; idx += 1024) {
      int i, k;
      if (AM == NASREC_AM_KC) {  // A(i,k) = a[i * lda + k]: k fastest
        i = idx / Kp;
        k = idx - i * Kp;
      } else {                   // A(i,k) = a[k * lda + i]: i fastest
        k = idx / MP;
        i = idx - k * MP;
      }
      float v = 0.f;
      if (i < M && k < sg.K) v = AM == NASREC_AM_KC ? sg.A[(long)i * sg.lda + k] : sg.A[(long)k * sg.lda + i];
      Wl[(kbase + k) * MP + i] = v;
    }
    kbase += Kp;
  }
  // the row biases go through LDS too: an LDS read in the epilogue is counted by lgkmcnt, a global one by vmcnt — behind the stores
  if (tid < TL_BIAS_FLOATS) Bl[tid] = (d.bias && d.bias_on_rows && tid < M) ? d.bias[tid] : 0.f;
  __syncthreads();

  const int g = lane >> 4, e = lane & 15;
  const bool acc_c = d.zmode ? s0.accumulate != 0 : d.beta != 0;
  // what the epilogue needs of the descriptor, once per workgroup (registers)
  const bool has_bias = d.bias != nullptr, bias_rows = d.bias_on_rows != 0, mask_rows = d.mask_on_rows != 0;
  const int dims = d.dims_in_use, act = d.act;
  float* const zbase = d.save_z;
  for (int b = wg * TL_WAVES + wave; b < Bs; b += wgs * TL_WAVES) {
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int kb = 0;
    for (int s = s_lo; s < s_hi; ++s) {
      const nasrec_gemm_seg_t& sg = d.seg[s];
      if (!sg.A || sg.K <= 0) continue;
      const int K4 = (sg.K + 3) >> 2;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.B) + (long)b * sg.ldb, 0, sg.K * 64, 0x00020000);
      for (int c0 = 0; c0 < K4; c0 += TL_CHUNK) {
        float xv[TL_CHUNK];
#pragma unroll
        for (int u = 0; u < TL_CHUNK; ++u)  // beyond the sample's K rows: zeros (hardware range check)
          xv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (c0 + u) * 256 + lane * 4, 0, 0));
#pragma unroll
        for (int u = 0; u < TL_CHUNK; ++u) {
          if (c0 + u < K4) {  // (uniform) LDS rows beyond the staged weights are not zero
            const float* wrow = Wl + (kb + 4 * (c0 + u) + g) * MP + e;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[rb * 16], xv[u], acc[rb], 0, 0, 0);
          }
        }
      }
      kb += 4 * K4;
    }
    // ---- epilogue == epilogue_store<NASREC_CM_TOKJ> (gemm_tile.h); D: row = 4 * (lane >> 4) + reg, column = lane & 15 ---------
    // Everything the sample's elements READ comes first (the accumulation target: 4 RB loads in flight; the row biases wait in LDS), ONE wait, then nothing but arithmetic and stores: vmcnt counts loads and stores in one in-order queue, so a load
    // behind a store — element by element: bias, C, store, bias, C, store — makes the wave wait for the store's acknowledgement each time.
    float* C = s0.C + (long)b * s0.ldc + e;
    float* Z = zbase ? zbase + (long)b * s0.ldc + e : nullptr;
    float cv[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[rb][r] = 0.f;
    if (acc_c) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[rb][r] = C[min(rb * 16 + 4 * g + r, M - 1) * 16];  // (clamped: rows >= M are never stored)
    }
    const float bcol = (has_bias && !bias_rows) ? d.bias[b * 16 + e] : 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), spelled out: the compiler cannot count the conditional stores below
    const bool dead_col = dims >= 0 && !mask_rows && b * 16 + e >= dims;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rb * 16 + 4 * g + r;
        if (i >= M) continue;
        float v = acc[rb][r];
        if (has_bias) v += bias_rows ? Bl[i] : bcol;
        if (Z) Z[i * 16] = v;
        v = act_apply(v, act);
        if (dead_col || (dims >= 0 && mask_rows && i >= dims)) v = 0.f;
        if (acc_c) v += cv[rb][r];
        C[i * 16] = v;
      }
  }
}

// Which launches take this path (the general template keeps everything else: small batches, ReLU-mask operands, split-K, ...)
bool token_linear_eligible(const nasrec_gemm_desc_t* d) {
  if (d->cmode != NASREC_CM_TOKJ || d->bmode != NASREC_AM_TOKR) return false;
  if (d->amode != NASREC_AM_KC && d->amode != NASREC_AM_RC) return false;
  if (d->splitk > 1 || d->pre_add || d->save_act || d->mul_nseg > 0) return false;
  const int nprob = d->zmode ? d->nseg : 1;
  for (int p = 0; p < nprob; ++p) {
    const nasrec_gemm_seg_t& s0 = d->seg[p];
    if (s0.M < 1 || s0.M > 80 || (s0.N & 15) || (s0.N >> 4) < 1024 || s0.ones_col) return false;
    if (s0.Mvalid > 0 && s0.Mvalid < s0.M) return false;
    const int mp = s0.M <= 16 ? 16 : (s0.M <= 48 ? 48 : 80);
    long kp = 0;
    for (int q = d->zmode ? p : 0; q < (d->zmode ? p + 1 : d->nseg); ++q) {
      const nasrec_gemm_seg_t& s = d->seg[q];
      if (s.Aaux || s.Baux) return false;
      if (!d->zmode && (s.M != s0.M || s.N != s0.N)) return false;
      if (s.A && s.K > 0) kp += (s.K + 3) & ~3;
      if ((long)s.K * 64 > 0x7fffffffL) return false;
    }
    if (kp * mp * 4 + TL_BIAS_FLOATS * 4 > TL_MAX_LDS) return false;
    if (d->zmode && p > 0 && (s0.N != d->seg[0].N || (s0.M + 15) / 16 != (d->seg[0].M + 15) / 16)) return false;  // one grid, one row-block count
  }
  return true;
}

template <int AM, int RB>
static void launch_token_linear_rb(hipStream_t st, const nasrec_gemm_desc_t* d, int grid, int wgs, size_t lds) {
  static unsigned long long big_lds_devices = 0;  // more than the default 64 KB of dynamic LDS must be requested once per kernel and device
  if (lds > 65536 && nasrec_lds_attr_needed(big_lds_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&token_linear_kernel<AM, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, TL_MAX_LDS);
  hipLaunchKernelGGL((token_linear_kernel<AM, RB>), dim3(grid), dim3(1024), lds, st, *d, wgs);
}

template <int AM>
static void launch_token_linear_t(hipStream_t st, const nasrec_gemm_desc_t* d, int rb, int grid, int wgs, size_t lds) {
  switch (rb) {
    case 1: launch_token_linear_rb<AM, 1>(st, d, grid, wgs, lds); break;
    case 2: launch_token_linear_rb<AM, 2>(st, d, grid, wgs, lds); break;
    case 3: launch_token_linear_rb<AM, 3>(st, d, grid, wgs, lds); break;
    case 4: launch_token_linear_rb<AM, 4>(st, d, grid, wgs, lds); break;
    default: launch_token_linear_rb<AM, 5>(st, d, grid, wgs, lds); break;
  }
}

int launch_token_linear(hipStream_t st, const nasrec_gemm_desc_t* d) {
  const int nprob = d->zmode ? d->nseg : 1;
  const int Bs = d->seg[0].N >> 4;
  const int rb = (d->seg[0].M + 15) / 16;
  const int mp = rb == 1 ? 16 : (rb <= 3 ? 48 : 80);
  long kp_max = 0;
  for (int p = 0; p < nprob; ++p) {
    long kp = 0;
    for (int q = d->zmode ? p : 0; q < (d->zmode ? p + 1 : d->nseg); ++q)
      if (d->seg[q].A && d->seg[q].K > 0) kp += (d->seg[q].K + 3) & ~3;
    if (kp > kp_max) kp_max = kp;
  }
  // a wavefront per sample, 16 per workgroup: the chip holds 256-512 workgroups; with several problems each one gets fewer
  // workgroups (more samples per wavefront, the weights are staged less often)
  int wgs = 256 / nprob;
  if (wgs < 64) wgs = 64;
  const int need = (Bs + TL_WAVES - 1) / TL_WAVES;
  if (wgs > need) wgs = need;
  const size_t lds = (size_t)(kp_max > 0 ? kp_max : 4) * mp * 4 + TL_BIAS_FLOATS * 4;
  if (d->amode == NASREC_AM_KC)
    launch_token_linear_t<NASREC_AM_KC>(st, d, rb, wgs * nprob, wgs, lds);
  else
    launch_token_linear_t<NASREC_AM_RC>(st, d, rb, wgs * nprob, wgs, lds);
  return nasrec_check_launch("token_linear");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Token-axis weight gradient at large batch (binding TOKK / TOKK / PLAIN): dW[i, j] = sum_b sum_e dz[b, i, e] x[b, j, e]
// (+ the bias gradient as a virtual ones-column), a batch of independent problems, K = B * 16.
// Per sample the product is [M, 16] x [16, N]: with the MFMA's k index taken as e = 4 g + step (g = lane >> 4) a lane's four
// A values for the four k-steps are ONE 16-byte load of row i = lane & 15 (all 64 lanes together read a contiguous 1 KB block
// of 16 rows), likewise for B — RB + CB loads and 4 RB CB MFMAs per sample, nothing staged through LDS.  A wavefront sums its
// samples in registers, the 16 wavefronts of a workgroup are added in fixed order through LDS, every workgroup writes one split-K
// slab and the general second pass (gemm_splitk_epilogue: fixed-order sum over the S slabs, row mask, accumulation, bias column)
// finishes — desc.splitk = S workgroups per problem, chosen by the plan.
// ---------------------------------------------------------------------------------------------------------------------------------
#define TDW_WAVES 16

template <int RB, int CB>
__global__ __launch_bounds__(64 * TDW_WAVES) void token_dw_kernel(const nasrec_gemm_desc_t d, int Mmax, int Nmax) {
  __shared__ __attribute__((aligned(16))) float red[4 * RB * CB * 4 * 64];
  // (the wave index as a SCALAR: the buffer resources below are built from it, and a resource the compiler takes for lane-dependent is
  // wrapped in a readfirstlane loop around every load)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = d.splitk;
  const int z = (int)blockIdx.x / S, ks = (int)blockIdx.x - z * S;
  const nasrec_gemm_seg_t& sg = d.seg[z];
  const int M = sg.M, N = sg.N, Nr = sg.ones_col ? N - 1 : N;
  const int Bs = sg.K >> 4;
  const int i16 = lane & 15, g = lane >> 4;
  f32x4 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ones_cb = sg.ones_col ? (N - 1) >> 4 : -1, ones_j = (N - 1) & 15;
  if (sg.A) {
    for (int b = ks * TDW_WAVES + wave; b < Bs; b += S * TDW_WAVES) {
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A) + (long)b * sg.lda, 0, M * 64, 0x00020000);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.B) + (long)b * sg.ldb, 0, Nr * 64, 0x00020000);
      f32x4 a[RB], x[CB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)  // rows beyond M: zeros (range check)
        a[rb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ((rb * 16 + i16) * 16 + 4 * g) * 4, 0, 0));
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        x[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ((cb * 16 + i16) * 16 + 4 * g) * 4, 0, 0));
        if (cb == ones_cb && i16 == ones_j) x[cb] = (f32x4){1.f, 1.f, 1.f, 1.f};
      }
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][st], x[cb][st], acc[rb][cb], 0, 0, 0);
    }
  }
  // ---- the workgroup's 16 partial sums: four LDS accumulators, wave w joins accumulator w % 4 in round w / 4 (fixed order) ------
  for (int round = 0; round < TDW_WAVES / 4; ++round) {
    if ((wave >> 2) == round) {
      float* mine = red + (wave & 3) * (RB * CB * 256);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = &mine[((rb * CB + cb) * 4 + r) * 64 + lane];
            *p = round == 0 ? acc[rb][cb][r] : *p + acc[rb][cb][r];
          }
    }
    __syncthreads();
  }
  // ---- slab of this split: D row = 4 * (lane >> 4) + reg, column = lane & 15 -----------------------------------------------------
  const int Mv = (sg.Mvalid > 0 && sg.Mvalid < M) ? sg.Mvalid : M;
  float* slab = d.workspace + ((long)(z * S + ks)) * Mmax * Nmax;
  for (int idx = tid; idx < RB * CB * 256; idx += 64 * TDW_WAVES) {
    const int blk = idx >> 8, r = (idx >> 6) & 3, l = idx & 63;
    const int rb = blk / CB, cb = blk - rb * CB;
    const int i = rb * 16 + 4 * (l >> 4) + r, j = cb * 16 + (l & 15);
    const int o = (blk * 4 + r) * 64 + l;
    const float v = (red[o] + red[RB * CB * 256 + o]) + (red[2 * RB * CB * 256 + o] + red[3 * RB * CB * 256 + o]);
    if (i < M && j < N) slab[(long)i * N + j] = i < Mv ? v : 0.f;
  }
}

bool token_dw_eligible(const nasrec_gemm_desc_t* d) {
  if (d->amode != NASREC_AM_TOKK || d->bmode != NASREC_AM_TOKK || d->cmode != NASREC_CM_PLAIN || !d->zmode) return false;
  if (d->splitk < 2 || !d->workspace) return false;
  for (int q = 0; q < d->nseg; ++q) {
    const nasrec_gemm_seg_t& s = d->seg[q];
    if (s.Aaux || s.Baux) return false;
    if (s.M < 1 || s.M > 80 || s.N < 1 || s.N > 80) return false;
    if ((s.K & 15) || (s.K >> 4) < 1024) return false;
  }
  return true;
}

template <int RB>
static void launch_token_dw_rb(hipStream_t st, const nasrec_gemm_desc_t* d, int cb, int grid, int Mmax, int Nmax) {
  switch (cb) {
    case 1: hipLaunchKernelGGL((token_dw_kernel<RB, 1>), dim3(grid), dim3(64 * TDW_WAVES), 0, st, *d, Mmax, Nmax); break;
    case 2: hipLaunchKernelGGL((token_dw_kernel<RB, 2>), dim3(grid), dim3(64 * TDW_WAVES), 0, st, *d, Mmax, Nmax); break;
    case 3: hipLaunchKernelGGL((token_dw_kernel<RB, 3>), dim3(grid), dim3(64 * TDW_WAVES), 0, st, *d, Mmax, Nmax); break;
    case 4: hipLaunchKernelGGL((token_dw_kernel<RB, 4>), dim3(grid), dim3(64 * TDW_WAVES), 0, st, *d, Mmax, Nmax); break;
    default: hipLaunchKernelGGL((token_dw_kernel<RB, 5>), dim3(grid), dim3(64 * TDW_WAVES), 0, st, *d, Mmax, Nmax); break;
  }
}

// main pass only: the caller (launch_gemm_t) runs the split-K second pass as for every other split launch
int launch_token_dw(hipStream_t st, const nasrec_gemm_desc_t* d, int Mmax, int Nmax) {
  const int rb = (Mmax + 15) / 16, cb = (Nmax + 15) / 16;
  const int grid = d->nseg * d->splitk;
  switch (rb) {
    case 1: launch_token_dw_rb<1>(st, d, cb, grid, Mmax, Nmax); break;
    case 2: launch_token_dw_rb<2>(st, d, cb, grid, Mmax, Nmax); break;
    case 3: launch_token_dw_rb<3>(st, d, cb, grid, Mmax, Nmax); break;
    case 4: launch_token_dw_rb<4>(st, d, cb, grid, Mmax, Nmax); break;
    default: launch_token_dw_rb<5>(st, d, cb, grid, Mmax, Nmax); break;
  }
  return 0;
}
