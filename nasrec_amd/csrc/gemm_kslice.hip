// Single-pass GEMM for the large forward products of the batch-256 step: y[M, N] = x[M, K] W[N, K]^T (+ epilogue) with M = 256,
// N ~ 768, K ~ 800 .. 1600 in up to 8 K-segments (binding KC / KC / PLAIN, no mask operands).
//
// The general template gives such a product 64x64 tiles and splits K over 5 workgroups per tile, so that 240 workgroups cover the
// chip; the partial tiles go through a workspace and a second launch sums them: 15 us + 5 us for 615 MFLOP (0.19 of the fp32 MFMA
// peak; hipBLASLt: 13 us).  Here K is split INSIDE the workgroup instead:
//   * a workgroup = 16 wavefronts = a 32 x 32 output tile (4 MFMA tiles of 16 x 16) x 4 k-slices: 192 workgroups for 256 x 768;
//   * a staged chunk is 256 deep: A[32][256] and B[32][256] (64 KB per chunk, 16 floats per thread as 16-byte buffer loads whose
//     extent check replaces the row predicates), parked in one of TWO LDS buffers — the loads of chunk t + 1 are in flight while the
//     16 waves run the MFMAs of chunk t, one barrier per chunk;
//   * wave (tile, slice) multiplies its 16 x 16 tile over k = 64 slice .. 64 slice + 63 of the chunk (16 v_mfma_f32_16x16x4_f32 per
//     chunk); at the end the four slices of a tile are added in fixed order through LDS and slice 0 runs the epilogue.
// No workspace, no second launch; the summation order (slice 0 + slice 1 + slice 2 + slice 3, k ascending inside a slice) is a fixed
// function of K, so results are reproducible run to run.
#include <cstdlib>
#include <cstring>
#include "gemm_tile.h"

#define KS_TM 32
#define KS_TN 32
#define KS_KC 256            // k depth of a staged chunk = 4 slices x 64
#define KS_LD (KS_KC + 4)    // LDS row stride (floats)
#define KS_BUF ((KS_TM + KS_TN) * KS_LD)
#define KS_PF 1             // chunks in flight in registers beyond the one being multiplied (measured: 1 -> 10.0 / 17.7 us for the
                            // K = 780 / 1565 products, 3 -> 11.2 / 19.4 us: the loop is LDS-write + MFMA bound, not load-latency bound)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ks_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats < 0 ? 0 : (floats > 0x1fffffffL ? 0x7fffffffL : 4 * floats)), 0x00020000);
}

__global__ __launch_bounds__(1024) void gemm_kslice_kernel(const nasrec_gemm_desc_t d, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float ks_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = wave & 3, slice = wave >> 2;
  const int tm = tile >> 1, tn = tile & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const nasrec_gemm_seg_t& s0 = d.seg[0];
  const int M = s0.M, N = s0.N;
  // XCD-contiguous tile order: the workgroups that share a 32-row panel of x are dealt to one L2
  const int nwg = gridDim.x, lin0 = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = lin0 & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin0 >> 3);
  const int by = lin / tiles_n, bx = lin - by * tiles_n;
  const int m0 = by * KS_TM, n0 = bx * KS_TN;

  // staging slots: thread -> (row, 4 consecutive k) of a [32][256] operand chunk, two per operand
  const int srow = tid >> 6, sk4 = (tid & 63) * 4;  // rows srow and srow + 16
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // a ring of KS_PF chunks in registers (the loads of chunk t + KS_PF are issued while chunk t is multiplied)
  f32x4 ra[KS_PF][2], rb[KS_PF][2];
  int lim[KS_PF];  // valid k of the chunk held in the slot (relative to its first k)

  // chunk cursor: (segment, chunk inside the segment)
  int seg = 0, ck = 0;
  while (seg < d.nseg && !(d.seg[seg].A && d.seg[seg].K > 0)) ++seg;
  auto fetch = [&](int slot) {  // loads of chunk (seg, ck) -> slot; past the last chunk: the null resource (zeros, no memory traffic)
    const bool live = seg < d.nseg;
    const nasrec_gemm_seg_t& sg = d.seg[live ? seg : 0];
    const int k0 = ck * KS_KC;
    lim[slot] = live ? sg.K - k0 : 0;
    const __amdgpu_buffer_rsrc_t rA = ks_rsrc(sg.A, live ? (long)(M - 1) * sg.lda + sg.K : 0);
    const __amdgpu_buffer_rsrc_t rB = ks_rsrc(sg.B, live ? (long)(N - 1) * sg.ldb + sg.K : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = srow + 16 * h;
      // rows beyond the operand are clamped (their products land in rows / columns nobody stores)
      const long oa = (long)min(m0 + r, M - 1) * sg.lda + k0 + sk4, ob = (long)min(n0 + r, N - 1) * sg.ldb + k0 + sk4;
      ra[slot][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(4 * oa), 0, 0));
      rb[slot][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, (int)(4 * ob), 0, 0));
    }
    if (live) {
      ++ck;
      if (ck * KS_KC >= sg.K) {
        ck = 0;
        do {
          ++seg;
        } while (seg < d.nseg && !(d.seg[seg].A && d.seg[seg].K > 0));
      }
    }
  };
  auto park = [&](int slot, float* buf) {  // registers -> LDS, k beyond the segment zeroed (a 16-byte load may run into the next row)
    float* As = buf;
    float* Bs = buf + KS_TM * KS_LD;
    const int l = lim[slot];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 a = ra[slot][h], b = rb[slot][h];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = sk4 + e < l;
        a[e] = in ? a[e] : 0.f;
        b[e] = in ? b[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(&As[(srow + 16 * h) * KS_LD + sk4]) = a;
      *reinterpret_cast<f32x4*>(&Bs[(srow + 16 * h) * KS_LD + sk4]) = b;
    }
  };
  auto multiply = [&](const float* buf, int cur_lim) {
    // a slice that lies entirely beyond the chunk's k (short last chunk of a segment) has nothing to add
    if (64 * slice < cur_lim) {
      const float* As = buf + (tm * 16 + fr) * KS_LD + 64 * slice + 4 * fg;
      const float* Bs = buf + KS_TM * KS_LD + (tn * 16 + fr) * KS_LD + 64 * slice + 4 * fg;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(As + 16 * kb);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(Bs + 16 * kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
      }
    }
  };

  int nchunks = 0;
  for (int q = 0; q < d.nseg; ++q)
    if (d.seg[q].A && d.seg[q].K > 0) nchunks += (d.seg[q].K + KS_KC - 1) / KS_KC;

#pragma unroll
  for (int r = 0; r < KS_PF; ++r) fetch(r);
  // chunk t lives in register slot t % KS_PF and goes to LDS buffer t & 1: the loop is unrolled KS_PF-fold so that slots are static
  // registers (the parity of t is a run-time select of one pointer).  A buffer is re-parked two chunks after it was read: every
  // reader has passed the barrier in between.
  int t = 0;
  for (; t + KS_PF <= nchunks; t += KS_PF) {
#pragma unroll
    for (int r = 0; r < KS_PF; ++r) {
      float* buf = ks_lds + ((t + r) & 1) * KS_BUF;
      const int cur = lim[r];
      park(r, buf);
      fetch(r);  // chunk t + r + KS_PF
      __syncthreads();
      multiply(buf, cur);
    }
  }
#pragma unroll
  for (int r = 0; r < KS_PF - 1; ++r) {
    if (t + r < nchunks) {
      float* buf = ks_lds + ((t + r) & 1) * KS_BUF;
      const int cur = lim[r];
      park(r, buf);
      __syncthreads();
      multiply(buf, cur);
    }
  }
  __syncthreads();
  // ---- the four k-slices of a tile, added in fixed order; slice 0 stores ---------------------------------------------------------
  float* red = ks_lds;  // [slice 1..3][tile][lane][4]
  if (slice > 0) *reinterpret_cast<f32x4*>(&red[(((slice - 1) * 4 + tile) * 64 + lane) * 4]) = acc;
  __syncthreads();
  if (slice == 0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = acc + *reinterpret_cast<const f32x4*>(&red[((s * 4 + tile) * 64 + lane) * 4]);
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = m0 + tm * 16 + 4 * fg + r, j = n0 + tn * 16 + fr;
      if (i < M && j < N) epilogue_store<NASREC_CM_PLAIN>(d, s0, i, j, acc[r]);
    }
  }
}


// ---- second form: no LDS staging at all --------------------------------------------------------------------------------------------
// The MFMA operand layout of v_mfma_f32_16x16x4_f32 (lane (r, g) supplies row r, k = g) is also a perfectly good global-load layout:
// lane (r, g) loads 16 bytes = k0 + 4 g .. + 3 of row r, four MFMAs consume them.  So a wave can feed itself: it owns the whole
// 32 x 32 tile (2 x 2 MFMA tiles: every loaded value is used twice) over every 16th k-step of 16, loads its four fragments per step
// straight into registers KD steps ahead, and never meets another wave until the final 16-way sum through LDS (64 KB, once).  Every
// byte of the two operand panels is loaded exactly once per workgroup (400 KB for K = 1565, the same as the staged form), there is no
// ds_write of operands (the staged form spends ~830 cycles per chunk on it), no barrier in the loop, and waves drift so that the
// loads of one overlap the MFMAs of another.  Summation order: wave w adds steps w, w + 16, ... in ascending k, the 16 waves are
// added 0 .. 15: a fixed function of the segment list.
#define KD 4       // k-steps in flight per wave (16 VGPRs each)
#define KS_SEGS 4  // live K-segments the kernel keeps in scalar registers

__global__ __launch_bounds__(1024) void gemm_kdirect_kernel(const nasrec_gemm_desc_t d, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float ks_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const nasrec_gemm_seg_t& s0 = d.seg[0];
  const int M = s0.M, N = s0.N;
  // XCD-compact tile order: workgroup ids are dealt round-robin to the 8 XCDs; XCD x gets a contiguous run of an order that walks
  // panels of 4 tile-rows column by column, i.e. a block of about 4 x 6 tiles: 128 rows of x + 192 rows of W per L2 (2 MB for
  // K = 1565) instead of one row panel and ALL of W (5 MB)
  const int nwg = gridDim.x, lin0 = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = lin0 & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin0 >> 3);
  const int PM = tiles_m < 4 ? tiles_m : 4;
  const int panel = lin / (PM * tiles_n), within = lin - panel * PM * tiles_n;
  const int ph = min(PM, tiles_m - panel * PM);
  const int bx = within / ph, by = panel * PM + (within - bx * ph);
  const int m0 = by * KS_TM, n0 = bx * KS_TN;

  // the (at most KS_SEGS) K-segments, in scalar registers: a step's segment is found by compare / select, no descriptor reads in the loop
  const float* sA[KS_SEGS];
  const float* sB[KS_SEGS];
  int sK[KS_SEGS], sLa[KS_SEGS], sLb[KS_SEGS], sN[KS_SEGS];
  int total = 0;
#pragma unroll
  for (int q = 0; q < KS_SEGS; ++q) {
    const bool live = q < d.nseg && d.seg[q].A && d.seg[q].K > 0;
    sA[q] = live ? d.seg[q].A : nullptr;
    sB[q] = d.seg[q].B;
    sK[q] = live ? d.seg[q].K : 0;
    sLa[q] = d.seg[q].lda;
    sLb[q] = d.seg[q].ldb;
    sN[q] = (sK[q] + 15) >> 4;
    total += sN[q];
  }

  const int ra0 = min(m0 + fr, M - 1), ra1 = min(m0 + 16 + fr, M - 1);  // rows beyond the operand are clamped (nobody stores them)
  const int rb0 = min(n0 + fr, N - 1), rb1 = min(n0 + 16 + fr, N - 1);
  f32x4 fa[KD][2], fb[KD][2];
  int klim[KD];  // valid k of the step held in the slot, relative to its first k (<= 0: nothing)
  int g = wave;  // the wave's steps: g = wave, wave + 16, ... over the concatenated segments
  auto fetch = [&](int slot) {
    const float *A = nullptr, *B = nullptr;
    int K = 0, la = 0, lb = 0, k0 = 0, rel = g;
#pragma unroll
    for (int q = 0; q < KS_SEGS; ++q) {
      const bool here = rel >= 0 && rel < sN[q];
      A = here ? sA[q] : A;
      B = here ? sB[q] : B;
      K = here ? sK[q] : K;
      la = here ? sLa[q] : la;
      lb = here ? sLb[q] : lb;
      k0 = here ? 16 * rel : k0;
      rel = here ? -1 : rel - sN[q];
    }
    klim[slot] = K - k0;  // past the last step: K = 0 and the null resource (zeros, no memory traffic)
    const __amdgpu_buffer_rsrc_t rA = ks_rsrc(A, A ? (long)(M - 1) * la + K : 0);
    const __amdgpu_buffer_rsrc_t rB = ks_rsrc(B, A ? (long)(N - 1) * lb + K : 0);
    const int kk = k0 + 4 * fg;
    fa[slot][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, 4 * (ra0 * la + kk), 0, 0));
    fa[slot][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, 4 * (ra1 * la + kk), 0, 0));
    fb[slot][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, 4 * (rb0 * lb + kk), 0, 0));
    fb[slot][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, 4 * (rb1 * lb + kk), 0, 0));
    g += 16;
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](int slot) {
    const int l = klim[slot];
    if (l <= 0) return;
    f32x4 a0 = fa[slot][0], a1 = fa[slot][1], b0 = fb[slot][0], b1 = fb[slot][1];
    if (l < 16) {  // last step of a segment: a 16-byte load may run into the next row
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = 4 * fg + e < l;
        a0[e] = in ? a0[e] : 0.f;
        a1[e] = in ? a1[e] : 0.f;
        b0[e] = in ? b0[e] : 0.f;
        b1[e] = in ? b1[e] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
    }
  };
#pragma unroll
  for (int r = 0; r < KD; ++r) fetch(r);
  const int mine = total > wave ? (total - wave + 15) >> 4 : 0;
  for (int t = 0; t < mine; t += KD) {
#pragma unroll
    for (int r = 0; r < KD; ++r) {
      multiply(r);  // (a slot past the wave's last step holds klim 0)
      fetch(r);
    }
  }
  // ---- 16-way sum in wave order; thread (e, lane) owns one output element -------------------------------------------------------
  float* red = ks_lds;  // [wave][a][b][r][lane]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * 16) + (a * 2 + b) * 4 + r) * 64 + lane] = acc[a][b][r];
  __syncthreads();
  {
    const int e = wave;  // (a, b, r) of this thread's element
    float v = red[e * 64 + lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) v += red[(w * 16 + e) * 64 + lane];
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    const int i = m0 + 16 * (e >> 3) + 4 * fg + (e & 3), j = n0 + 16 * ((e >> 2) & 1) + fr;
    if (i < M && j < N) epilogue_store<NASREC_CM_PLAIN>(d, s0, i, j, v);
  }
}

// Which launches take this kernel: one dense forward-type product (k-contiguous operands, plain output), no mask operands / virtual
// column / row prefix, enough 32 x 32 tiles to cover most of the chip but not so many that 1024-thread workgroups queue up, and
// a K deep enough that splitting it is what the general template would do anyway.  plan.py mirrors this (`kslice_eligible`) and
// gives such launches splitk = 1.
bool gemm_kslice_eligible(const nasrec_gemm_desc_t* d) {
  if (d->amode != NASREC_AM_KC || d->bmode != NASREC_AM_KC || d->cmode != NASREC_CM_PLAIN || d->zmode || d->splitk > 1) return false;
  const nasrec_gemm_seg_t& s0 = d->seg[0];
  long K = 0;
  for (int q = 0; q < d->nseg; ++q) {
    const nasrec_gemm_seg_t& s = d->seg[q];
    if (s.Aaux || s.Baux || s.ones_col || (s.Mvalid > 0 && s.Mvalid < s0.M)) return false;
    if (s.A && s.K > 0) {
      K += s.K;
      if ((long)s0.M * s.lda >= (1L << 29) || (long)s0.N * s.ldb >= (1L << 29)) return false;
    }
  }
  if (d->nseg > KS_SEGS) return false;
  const long tiles = (long)((s0.M + KS_TM - 1) / KS_TM) * ((s0.N + KS_TN - 1) / KS_TN);
  return s0.M <= 512 && tiles >= 128 && tiles <= 512 && K >= 512;
}

int launch_gemm_kslice(hipStream_t st, const nasrec_gemm_desc_t* d) {
  const nasrec_gemm_seg_t& s0 = d->seg[0];
  const int tiles_m = (s0.M + KS_TM - 1) / KS_TM, tiles_n = (s0.N + KS_TN - 1) / KS_TN;
  static int form = -1;  // NASREC_KSLICE_FORM=staged keeps the LDS-staged kernel (A/B measurements)
  static bool attr = false;
  if (form < 0) {
    const char* e = getenv("NASREC_KSLICE_FORM");
    form = (e && !strcmp(e, "staged")) ? 1 : 0;
  }
  if (form == 0) {
    hipLaunchKernelGGL(gemm_kdirect_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(1024), sizeof(float) * 16 * 16 * 64, st, *d, tiles_m, tiles_n);
    return nasrec_check_launch("gemm_kdirect");
  }
  const size_t lds = sizeof(float) * 2 * KS_BUF;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kslice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_kslice_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(1024), lds, st, *d, tiles_n);
  return nasrec_check_launch("gemm_kslice");
}
