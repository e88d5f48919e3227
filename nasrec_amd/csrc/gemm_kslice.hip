// Single-pass GEMM for the large forward products of the batch-256 step: y[M, N] = x[M, K] W[N, K]^T (+ epilogue) with M = 256,
// N ~ 768, K ~ 800 .. 1600 in up to KS_SEGS K-segments (binding KC / KC / PLAIN, no mask operands).
//
// The general template gives such a product 64x64 tiles and splits K over 5 workgroups per tile, so that 240 workgroups cover the
// chip; the partial tiles go through a workspace and a second launch sums them: 15 us + 5 us for 615 MFLOP (0.19 of the fp32 MFMA
// peak).  Here K is split INSIDE the workgroup instead:
//   * a workgroup = 16 wavefronts = a 32 x 32 output tile: 192 workgroups for 256 x 768, in an XCD-compact order (each XCD's L2
//     serves a block of about 4 x 6 tiles: 128 rows of x + 192 rows of W; 16 MB of fabric reads per launch by the FETCH_SIZE counter);
//   * operands are staged by LDS-DMA (buffer_load_dwordx4 ... lds) in chunks of 128 k: 64 rows x 512 B = 32 KB per chunk, 1-KB
//     pieces (two rows x 512 B), four LDS buffers, three chunks in flight across raw barriers (counted vmcnt).  What this form is
//     built around (tools/micro/panel_probe.hip, kdma_probe.hip on the MI355X): a workgroup that pulls its 400 KB of operand panels
//     with MFMA-fragment-shaped loads (16 rows x 64 B per wave-instruction) runs at 36 GB/s per CU — the texture-address path, not
//     L2 or the fabric, bounds it (12.4 us for K = 1536 with no MFMA at all) — while whole 512-B row pieces arrive at 130 - 145 GB/s
//     per CU; the DMA writes them into LDS without a register round trip or ds_write pass;
//   * the LDS image is [row][k-group] with the 16-byte k-group index XORed with (row & 15) — applied to the per-lane SOURCE address,
//     the DMA destination is lane-linear — so the fragment reads (16 rows, same k-group) are conflict-free;
//   * the waves are specialised: 8 feeders only issue DMA pieces (a wave stalls 100 - 200 cycles per piece; with every wave doing
//     both jobs in barrier lock-step the MFMA pipes idled through every issue phase and DMA time and MFMA time simply added up),
//     8 multipliers (tile, k-half) read fragments and run the MFMAs, one iteration behind their reads, two accumulators each;
//     the feeders are the OLDER waves 0..7 of the workgroup (issue arbitration favours them: 13.9 -> 13.55 us);
//   * K-segments of at most 16 (the raw dense features) never become chunks: the multipliers load their fragments directly at the
//     start and multiply them after the loop;
//   * the descriptor fields the prologue needs are read by one block of scalar loads behind one wait (every DEPENDENT read of the
//     cold argument segment is a memory round trip: a lazily compiled prologue started its first DMA 6 us after entry).
//   At the end the two k-halves of a tile are added through LDS (half 0 + half 1) and all 1024 threads run the epilogue, one element each.
// No workspace, no second launch; the summation order is a fixed function of the segment list, so results are reproducible run to run.
// Measured (256 x 768, cold L2 every launch, tools/kslice_probe.py): K = 13 + 768 + 768 + 16 with bias + ReLU 13.4 us (46 TFLOP/s,
// 0.29 of peak; general template + second pass 20.0 us, a register-staged 256-deep form 17.7 us, fragment loads straight to
// registers 18.6 us, this form before the waves were specialised 16.0 us), K = 780 9.6 us.  Where the rest goes (in-kernel stamps,
// tools/kslice_stamps.py, K = 1565): ~1800 cycles per chunk against 1024 cycles of MFMA per SIMD and chunk.  The feeders never wait
// for their pieces (a fifth LDS buffer, 128 KB in flight, changed nothing); they spend 500 (x) to 900 (W) cycles per chunk issuing
// four pieces each, and the multipliers' 16 MFMAs per chunk take 1150 - 1600 cycles while feeders issue on the same SIMD (raising
// the multipliers' s_setprio changed nothing either).  Launch + drain outside the workgroups' lifetime: ~3 us of the 13.4.
#include <cstdlib>
#include <cstring>
#include "gemm_tile.h"

#define KS_TM 32
#define KS_TN 32
#define KS_DC 128                       // k depth of a staged chunk = 2 halves x 64
#define KS_NB 4                         // LDS buffers (KS_NB - 1 chunks in flight; a fifth buffer, 128 KB in flight, changed nothing: the
                                        // feeders never wait for their pieces, tools/kslice_stamps.py)
#define KS_CH ((KS_TM + KS_TN) * KS_DC)  // floats per buffer: rows 0..31 = x tile, 32..63 = W tile
#define KS_SEGS 4
#ifdef KS_STAMPS  // tools/kslice_stamps.py: wave 0 of every workgroup writes the 100 MHz clock at six points into desc.workspace
#define KS_STAMP(i) \
  if (tid == 0 && d.workspace) d.workspace[blockIdx.x * 64 + (i)] = __builtin_bit_cast(float, (unsigned)__builtin_readcyclecounter())
// per-chunk attribution for wave 0 (a multiplier): cycles from reaching a barrier to leaving it (workspace[.. + 7]); everything else
// of the k-loop is issue + MFMA time
#define KS_LOOP_STAMP_A const unsigned ks_ta = (unsigned)__builtin_readcyclecounter()
#define KS_LOOP_STAMP_B ks_wait += (unsigned)__builtin_readcyclecounter() - ks_ta
#else
#define KS_STAMP(i)
#define KS_LOOP_STAMP_A
#define KS_LOOP_STAMP_B
#endif                       // live K-segments the kernel keeps in scalar registers

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ks_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats <= 0 ? 0 : (floats > 0x1fffffffL ? 0x7fffffffL : 4 * floats)), 0x00020000);
}

__global__ __launch_bounds__(1024) void gemm_kslice_kernel(const nasrec_gemm_desc_t d, int tiles_m_, int tiles_n_) {
  extern __shared__ __attribute__((aligned(16))) float ks_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  KS_STAMP(0);
  warm_kernarg<1024>();  // (the epilogue's fields are read lazily, one by one)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef KS_FEED_FIRST
#define KS_FEED_FIRST 1
#endif
  // role index: roles 8..15 feed, 0..7 multiply.  The feeders take the OLDER waves 0..7 (the issue arbiter favours the older wave of a
  // SIMD, and the feeders' piece issue is what sets the chunk period): the launch 13.9 -> 13.55 us on both boxes of the A/B
  const int rw = KS_FEED_FIRST ? (wave ^ 8) : wave;
  const int tile = rw & 3;
  const int tm = tile >> 1, tn = tile & 1;
  const int fr = lane & 15, fg = lane >> 4;
  // the (at most KS_SEGS) K-segments, in scalar registers: a chunk's segment is found by compare / select, no descriptor reads in the loop
  // They are read from the descriptor by ONE block of scalar loads with one wait: the descriptor is a kernel argument, cold in every
  // cache when the kernel starts, and the compiler loads fields lazily next to their first use — a prologue that tested
  // `q < nseg && seg[q].A` field by field, or divided by tiles_n before it read the segments, paid a memory round trip per level
  // (0.6 - 2 us each) and started its first DMA up to 6 us late.
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 ab[KS_SEGS], mnkl[KS_SEGS];
  i32x2 ll[KS_SEGS], tiles;
  int nseg;
  {
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int S0 = offsetof(nasrec_gemm_desc_t, seg), SS = sizeof(nasrec_gemm_seg_t), OM = offsetof(nasrec_gemm_seg_t, M),
                  OL = offsetof(nasrec_gemm_seg_t, ldb);
    static_assert(KS_SEGS == 4 && offsetof(nasrec_gemm_seg_t, B) == 8 && offsetof(nasrec_gemm_seg_t, lda) == OM + 12 &&
                      offsetof(nasrec_gemm_seg_t, ldc) == OL + 4 && sizeof(nasrec_gemm_desc_t) % 8 == 0,
                  "descriptor layout the scalar loads below assume");
    asm volatile(
        "s_load_dword %12, %14, %c15\n"
        "s_load_dwordx2 %13, %14, %c16\n"
        "s_load_dwordx4 %0, %14, %c17\n"
        "s_load_dwordx4 %1, %14, %c18\n"
        "s_load_dwordx4 %2, %14, %c19\n"
        "s_load_dwordx4 %3, %14, %c20\n"
        "s_load_dwordx4 %4, %14, %c21\n"
        "s_load_dwordx4 %5, %14, %c22\n"
        "s_load_dwordx4 %6, %14, %c23\n"
        "s_load_dwordx4 %7, %14, %c24\n"
        "s_load_dwordx2 %8, %14, %c25\n"
        "s_load_dwordx2 %9, %14, %c26\n"
        "s_load_dwordx2 %10, %14, %c27\n"
        "s_load_dwordx2 %11, %14, %c28\n"
        "s_waitcnt lgkmcnt(0)"  // (these hit the lines warm_kernarg has just pulled in)
        : "=&s"(ab[0]), "=&s"(ab[1]), "=&s"(ab[2]), "=&s"(ab[3]), "=&s"(mnkl[0]), "=&s"(mnkl[1]), "=&s"(mnkl[2]), "=&s"(mnkl[3]),
          "=&s"(ll[0]), "=&s"(ll[1]), "=&s"(ll[2]), "=&s"(ll[3]), "=&s"(nseg), "=&s"(tiles)
        : "s"(ka), "n"(offsetof(nasrec_gemm_desc_t, nseg)), "n"(sizeof(nasrec_gemm_desc_t)), "n"(S0), "n"(S0 + SS), "n"(S0 + 2 * SS),
          "n"(S0 + 3 * SS), "n"(S0 + OM), "n"(S0 + SS + OM), "n"(S0 + 2 * SS + OM), "n"(S0 + 3 * SS + OM), "n"(S0 + OL),
          "n"(S0 + SS + OL), "n"(S0 + 2 * SS + OL), "n"(S0 + 3 * SS + OL)
        : "memory");
  }
  const float* sA[KS_SEGS];
  const float* sB[KS_SEGS];
  int sK[KS_SEGS], sLa[KS_SEGS], sLb[KS_SEGS], sN[KS_SEGS];
#pragma unroll
  for (int q = 0; q < KS_SEGS; ++q) {
    sA[q] = reinterpret_cast<const float*>(((unsigned long long)(unsigned)ab[q][1] << 32) | (unsigned)ab[q][0]);
    sB[q] = reinterpret_cast<const float*>(((unsigned long long)(unsigned)ab[q][3] << 32) | (unsigned)ab[q][2]);
    sK[q] = mnkl[q][2];
    sLa[q] = mnkl[q][3];
    sLb[q] = ll[q][0];
  }
  const nasrec_gemm_seg_t& s0 = d.seg[0];
  const int M = mnkl[0][0], N = mnkl[0][1];
  const int tiles_m = tiles[0], tiles_n = tiles[1];
  // XCD-compact tile order: workgroup ids are dealt round-robin to the 8 XCDs; XCD x gets a contiguous run of an order that walks
  // panels of 4 tile-rows column by column
  const int nwg = tiles_m * tiles_n, lin0 = blockIdx.x;  // (gridDim.x would be one more read of the argument segment)
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = lin0 & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin0 >> 3);
  const int PM = tiles_m < 4 ? tiles_m : 4;
  const int panel = lin / (PM * tiles_n), within = lin - panel * PM * tiles_n;
  const int ph = min(PM, tiles_m - panel * PM);
  const int bx = within / ph, by = panel * PM + (within - bx * ph);
  const int m0 = by * KS_TM, n0 = bx * KS_TN;

  int nchunks = 0, ntiny = 0, sT[KS_SEGS];
#pragma unroll
  for (int q = 0; q < KS_SEGS; ++q) {
    const bool live = (q < nseg) & (sA[q] != nullptr) & (sK[q] > 0);
    sK[q] = live ? sK[q] : 0;
    // at most two tiny segments are taken out of the chunk sequence (one per k-half of the multipliers)
    const bool tiny = (sK[q] > 0) & (sK[q] <= 16) & (ntiny < 2);
    sT[q] = tiny ? sK[q] : 0;
    sK[q] = tiny ? 0 : sK[q];
    ntiny += tiny ? 1 : 0;
    sN[q] = (sK[q] + KS_DC - 1) / KS_DC;
    nchunks += sN[q];
  }
  // ---- roles: rw 0..7 multiply (tile = rw & 3, k-half = rw >> 2: two per SIMD), rw 8..15 only feed the LDS-DMA ------------
  // A wave that issues a DMA piece stalls 100 - 200 cycles per piece, and with every wave doing both jobs in barrier lock-step the
  // MFMA pipes idled through every issue phase (measured: DMA and MFMA time added up, 1900 cycles per chunk for 1024 of MFMA).
  // With the jobs split the issue stalls of the feeders sit under the MFMAs of the multipliers.
  const bool feeder = rw >= 8;
  float* red = ks_lds;  // (after the loop) [half][tile][reg][lane]
  if (feeder) {
    // feeder f = rw - 8 owns pieces 4f .. 4f + 3 of every chunk: piece q fills LDS rows 2q, 2q + 1 (512 B each; rows 0..31 = x tile,
    // 32..63 = W tile: feeders 0..3 read x, 4..7 read W); lane -> (row, 16-byte slot p), which holds k-group p ^ (row & 15)
    const int f = rw - 8;
    const bool isB = f >= 4;
    const int pslot = lane & 31;
    int grow[4], gkg[4], growld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 2 * (4 * f + i) + (lane >> 5);
      gkg[i] = pslot ^ (row & 15);
      // rows beyond the operand are clamped (their products land in rows / columns nobody stores)
      grow[i] = isB ? min(n0 + row - 32, N - 1) : min(m0 + row, M - 1);
    }
    // cursor over the chunk sequence (segment, first k): an add and a compare per chunk; the segment's parameters are re-selected only
    // when a segment ends (a per-chunk select chain over the segments cost 0.4 us per chunk on the shared scalar unit)
    auto next_live = [&](int seg) {  // first live segment after `seg` (KS_SEGS: none)
      int r = KS_SEGS;
#pragma unroll
      for (int q = KS_SEGS - 1; q >= 0; --q) r = (q > seg && sK[q] > 0) ? q : r;
      return r;
    };
    const float* iP;
    int iseg = next_live(-1), iK, ild, ik0 = 0, ibuf = 0;
    __amdgpu_buffer_rsrc_t irs;
    auto issue_segment = [&]() {
      iP = nullptr, iK = 0, ild = 0;
#pragma unroll
      for (int q = 0; q < KS_SEGS; ++q) {
        const bool here = iseg == q;
        iP = here ? (isB ? sB[q] : sA[q]) : iP;
        iK = here ? sK[q] : iK;
        ild = here ? (isB ? sLb[q] : sLa[q]) : ild;
      }
      // past the last segment K = 0: every lane out of range, the DMA writes zeros and moves no memory
      irs = ks_rsrc(iP, iK > 0 ? (long)((isB ? N : M) - 1) * ild + iK : 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) growld[i] = grow[i] * ild;
    };
    issue_segment();
    auto issue = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = ik0 + 4 * gkg[i];
        const int vo = k < iK ? 4 * (growld[i] + k) : 0x7ffffff0;
        float* dst = ks_lds + ibuf * KS_CH + (4 * f + i) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
      }
      ibuf = ibuf + 1 == KS_NB ? 0 : ibuf + 1;
      ik0 += KS_DC;
      if (ik0 >= iK && iseg < KS_SEGS) {
        iseg = next_live(iseg);
        ik0 = 0;
        issue_segment();
      }
    };
#pragma unroll
    for (int c = 0; c < KS_NB - 1; ++c) issue();
    // barrier c: this feeder's pieces of chunk c have landed (counted vmcnt: chunks c + 1, c + 2 may still be in flight) and the
    // multipliers are done reading chunk c - 1, whose buffer chunk c + 3 overwrites
#ifdef KS_STAMPS
    unsigned fw_data = 0, fw_bar = 0, fw_issue = 0;
#endif
    for (int c = 0; c < nchunks; ++c) {
#ifdef KS_STAMPS
      const unsigned ta = (unsigned)__builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // 4 pieces per chunk, KS_NB - 2 later chunks may stay in flight
#ifdef KS_STAMPS
      const unsigned tb = (unsigned)__builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef KS_STAMPS
      const unsigned tc = (unsigned)__builtin_readcyclecounter();
#endif
      issue();
#ifdef KS_STAMPS
      const unsigned td = (unsigned)__builtin_readcyclecounter();
      fw_data += tb - ta, fw_bar += tc - tb, fw_issue += td - tc;
#endif
    }
#ifdef KS_STAMPS  // feeder 0 (wave 8): cycles waiting for its pieces / at barriers / issuing, over the k-loop
    if (lane == 0 && d.workspace) {
      d.workspace[blockIdx.x * 64 + 16 + wave * 3 + 0] = __builtin_bit_cast(float, fw_data);
      d.workspace[blockIdx.x * 64 + 16 + wave * 3 + 1] = __builtin_bit_cast(float, fw_bar);
      d.workspace[blockIdx.x * 64 + 16 + wave * 3 + 2] = __builtin_bit_cast(float, fw_issue);
    }
#endif
    KS_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    const int half = rw >> 2;
#ifdef KS_STAMPS
    unsigned ks_wait = 0;
    const unsigned ks_t0 = (unsigned)__builtin_readcyclecounter();
#endif
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2][4], fb[2][4];
    int flim[2];
    int ck = 0, cbuf = 0;  // chunk cursor: (first k inside its segment); segment ends are found from the limits below
    int cseg = 0, cK = 0;
    auto seg_k = [&](int seg) {
      int K = 0;
#pragma unroll
      for (int q = 0; q < KS_SEGS; ++q) K = seg == q ? sK[q] : K;
      return K;
    };
    while (cseg < KS_SEGS && seg_k(cseg) == 0) ++cseg;
    cK = seg_k(cseg);
    // Tiny segments (K <= 16: the raw dense features of the Criteo networks ride along as 13- and 16-wide segments) do not get a
    // chunk of their own — a chunk costs a barrier round of ~1800 cycles whatever it holds.  Multiplier (tile, half) loads the
    // fragments of tiny segment number `half` straight into registers now (two 16-byte buffer loads: 16 rows x 64 B each, nothing at
    // this size) and multiplies them after the loop.  (An L2 prefetch by the multipliers — one dword of every line of the panels, up
    // front — was tried to turn the feeders' Infinity-Cache trips (2.3 us per chunk, the bound of this loop: 96 KB in flight) into L2
    // hits: its 64-lines-per-instruction loads cost the shared texture-address path more than they saved, 14.3 -> 16.4 us.)
    f32x4 ta = {0.f, 0.f, 0.f, 0.f}, tb = {0.f, 0.f, 0.f, 0.f};
    int tK = 0;
    {
      const float *tA = nullptr, *tB = nullptr;
      int tla = 0, tlb = 0, seen = 0;
#pragma unroll
      for (int q = 0; q < KS_SEGS; ++q) {
        const bool tiny = sT[q] > 0;
        const bool mine = tiny & (seen == half);
        tA = mine ? sA[q] : tA;
        tB = mine ? sB[q] : tB;
        tla = mine ? sLa[q] : tla;
        tlb = mine ? sLb[q] : tlb;
        tK = mine ? sT[q] : tK;
        seen += tiny ? 1 : 0;
      }
      if (tK > 0) {
        const __amdgpu_buffer_rsrc_t ra = ks_rsrc(tA, (long)(M - 1) * tla + tK), rb = ks_rsrc(tB, (long)(N - 1) * tlb + tK);
        const int ia = min(m0 + tm * 16 + fr, M - 1), ib = min(n0 + tn * 16 + fr, N - 1);
        ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, 4 * (ia * tla + 4 * fg), 0, 0));
        tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, 4 * (ib * tlb + 4 * fg), 0, 0));
      }
    }
    auto readfrag = [&](int set) {
      const float* buf = ks_lds + cbuf * KS_CH;
      flim[set] = cK - ck;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int g = 16 * half + 4 * kb + fg;
        fa[set][kb] = *reinterpret_cast<const f32x4*>(buf + (tm * 16 + fr) * KS_DC + 4 * (g ^ fr));
        fb[set][kb] = *reinterpret_cast<const f32x4*>(buf + (32 + tn * 16 + fr) * KS_DC + 4 * (g ^ fr));
      }
      cbuf = cbuf + 1 == KS_NB ? 0 : cbuf + 1;
      ck += KS_DC;
      if (ck >= cK) {
        do {
          ++cseg;
        } while (cseg < KS_SEGS && seg_k(cseg) == 0);
        ck = 0;
        cK = seg_k(cseg);
      }
    };
    auto multiply = [&](int set) {
      const int lim = flim[set];
      if (lim < KS_DC) {  // last chunk of a segment: a 16-byte piece may straddle K (whole pieces beyond K were written as zeros)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool in = 4 * (16 * half + 4 * kb + fg) + e < lim;
            fa[set][kb][e] = in ? fa[set][kb][e] : 0.f;
            fb[set][kb][e] = in ? fb[set][kb][e] : 0.f;
          }
      }
      // two accumulators (even / odd k-steps): consecutive MFMAs of a wave do not wait for each other
#pragma unroll
      for (int kb = 0; kb < 4; kb += 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][kb][j], fb[set][kb][j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][kb + 1][j], fb[set][kb + 1][j], acc1, 0, 0, 0);
        }
    };
    // fragments of chunk c are read right after barrier c and multiplied one iteration later (two register sets): the MFMAs of chunk
    // c - 1 run while the reads of chunk c are in flight.  lgkmcnt(0) before a barrier: the reads of the buffer the feeders are
    // about to overwrite have returned.
    for (int t = 0; t < nchunks + 1; t += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = t + h;
        if (c < nchunks) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          KS_LOOP_STAMP_A;
          __builtin_amdgcn_s_barrier();
          KS_LOOP_STAMP_B;
          readfrag(h);
        }
        if (c >= 1 && c <= nchunks) multiply(h ^ 1);
      }
    }
    KS_STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (the feeders' trailing zero pieces have landed: the buffers are free)
    if (tK > 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = 4 * fg + e < tK;  // (a 16-byte load may run into the next row)
        ta[e] = in ? ta[e] : 0.f;
        tb[e] = in ? tb[e] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[j], tb[j], acc0, 0, 0, 0);
    }
    const f32x4 acc = acc0 + acc1;
#ifdef KS_STAMPS
    if (lane == 0 && d.workspace) {
      d.workspace[blockIdx.x * 64 + 16 + wave * 3 + 1] = __builtin_bit_cast(float, ks_wait);
      d.workspace[blockIdx.x * 64 + 16 + wave * 3 + 2] = __builtin_bit_cast(float, (unsigned)__builtin_readcyclecounter() - ks_t0);
    }
#endif
    *reinterpret_cast<f32x4*>(&red[((half * 4 + tile) * 64 + lane) * 4]) = acc;
  }
  __syncthreads();
  KS_STAMP(5);
  // ---- the two k-halves of a tile are added (half 0 + half 1); all 1024 threads run the epilogue, one output element each ------------
  {
    const int e = wave;  // (tile, register) of this thread's element: C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    const int etile = e >> 2, r = e & 3;
    const float v = red[((0 * 4 + etile) * 64 + lane) * 4 + r] + red[((1 * 4 + etile) * 64 + lane) * 4 + r];
    const int i = m0 + (etile >> 1) * 16 + 4 * fg + r, j = n0 + (etile & 1) * 16 + fr;
    if (i < M && j < N) epilogue_store<NASREC_CM_PLAIN>(d, s0, i, j, v);
  }
  KS_STAMP(6);
}

// Which launches take this kernel: one dense forward-type product (k-contiguous operands, plain output), no mask operands / virtual
// column / row prefix, at most KS_SEGS segments, enough 32 x 32 tiles to cover most of the chip but not so many that 1024-thread
// workgroups queue up, and a K deep enough that splitting it is what the general template would do anyway.  plan.py mirrors this
// (`kslice_eligible`) and gives such launches splitk = 1.
bool gemm_kslice_eligible(const nasrec_gemm_desc_t* d) {
  if (d->amode != NASREC_AM_KC || d->bmode != NASREC_AM_KC || d->cmode != NASREC_CM_PLAIN || d->zmode || d->splitk > 1) return false;
  if (d->nseg > KS_SEGS) return false;
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
  const long tiles = (long)((s0.M + KS_TM - 1) / KS_TM) * ((s0.N + KS_TN - 1) / KS_TN);
  return s0.M <= 512 && tiles >= 128 && tiles <= 512 && K >= 512;
}

int launch_gemm_kslice(hipStream_t st, const nasrec_gemm_desc_t* d) {
  const nasrec_gemm_seg_t& s0 = d->seg[0];
  const int tiles_m = (s0.M + KS_TM - 1) / KS_TM, tiles_n = (s0.N + KS_TN - 1) / KS_TN;
  static unsigned long long attr_devices = 0;
  const size_t lds = sizeof(float) * KS_NB * KS_CH;
  if (nasrec_lds_attr_needed(attr_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kslice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gemm_kslice_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(1024), lds, st, *d, tiles_m, tiles_n);
  return nasrec_check_launch("gemm_kslice");
}
