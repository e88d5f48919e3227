// Latency-regime main loop of the fp32 MFMA GEMM family: the same block tile, operand bindings, fragment order and epilogue as
// gemm_tile (gemm_tile.h) — results are bit-identical — but with a RING of staged k-tiles in registers.
//
// At batch 256 a launch has ~1 workgroup per CU and every kernel starts on a cold L2: a staged k-tile is a dependent
// MALL/HBM round trip (1.6 - 3.3 us per 64-deep tile measured: the dominant product spends 19.6 us on 5 tiles, 40 us on 25
// without split-K) against ~0.1 us of MFMA work.  gemm_tile keeps ONE tile in flight, so a workgroup pays that latency once
// per tile.  Here the loads of the next RING tiles are all in flight: the prologue issues them back to back, and every
// iteration re-issues the slot it just parked in LDS.  For the compiler's vmcnt counting to survive, the loop body is free of
// branches around loads and is unrolled RING-fold so that ring slots are static registers:
//   * ONE fetch path: per slot a byte offset fixed at segment entry; a k beyond the segment is redirected to the tile's first k
//     (valid memory) by one select and zeroed when the tile is parked — no separate tail path;
//   * refills are unconditional: past the last tile of the workgroup they reload its last tile (never parked);
//   * ReLU-mask operands are a template variant (AUX), not a run-time branch;
//   * the segment cursor is advanced in a branch that contains no loads.
// Row predicates (Mvalid, the virtual ones-column) are per workgroup: launch_gemm_t routes a launch whose k-segments disagree on
// them to gemm_tile.
#pragma once
#include <type_traits>
#include "gemm_tile.h"

// thread -> first element (rr, kk) of staging load `it`: V = 4 floats per lane (one 16-byte load) along the operand's contiguous axis
// — k for KC / TOKK (parked with one ds_write_b128), r for RC / TOKR (four dword parks) — or V = 1, the one-dword map of
// gemm_tile.h, when a thread's share of the tile is not a multiple of four floats (the shallow TK = 32 strips).  The LDS image is
// the same either way; the one-dword form needs 4x the staging instructions (gemm_rt.h, the worklist's copy of this loop).
template <int MODE, int V, int NT, int TK, int R>
__device__ __forceinline__ void ring_stage_coords(int tid, int it, int& rr, int& kk) {
  constexpr bool KC = MODE == NASREC_AM_KC || MODE == NASREC_AM_TOKK;
  if (V == 1) {
    stage_coords<MODE, NT, TK, R>(tid, it, rr, kk);
  } else if (KC) {
    kk = 4 * (tid & (TK / 4 - 1));
    rr = tid / (TK / 4) + (NT / (TK / 4)) * it;
  } else {
    rr = 4 * (tid & (R / 4 - 1));
    kk = tid / (R / 4) + (NT / (R / 4)) * it;
  }
}

template <int AM, int BMODE, int CM, int NT, int TK, int TBM, int TBN, bool AUX, int RING>
__device__ __forceinline__ void gemm_tile_ring(const nasrec_gemm_desc_t& d, int Mmax, int Nmax, const int bx, const int by, const int bz) {
  constexpr int LDS_LD = TK + 4;
  constexpr bool KCA = AM == NASREC_AM_KC || AM == NASREC_AM_TOKK, KCB = BMODE == NASREC_AM_KC || BMODE == NASREC_AM_TOKK;
  constexpr int VA = (TBM * TK / NT) % 4 == 0 && TK % 16 == 0 && TBM % 4 == 0 ? 4 : 1;  // floats per staging load
  constexpr int VB = (TBN * TK / NT) % 4 == 0 && TK % 16 == 0 && TBN % 4 == 0 ? 4 : 1;
  constexpr int NITA = TBM * TK / NT / VA, NITB = TBN * TK / NT / VB;                    // staging loads per thread and k-tile
  constexpr int PER_WAVE = (TBM / 16) * (TBN / 16) / (NT / 64);
  constexpr int WTM = (PER_WAVE >= 2 && TBM >= 32) ? 32 : 16;
  constexpr int WTN = 16 * PER_WAVE / (WTM / 16);
  constexpr int FA = WTM / 16, FB = WTN / 16;
  static_assert(NITA >= 1 && NITB >= 1 && PER_WAVE >= 1 && (TBM / WTM) * (TBN / WTN) == NT / 64, "tile configuration");
  __shared__ __attribute__((aligned(16))) float As[TBM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[TBN * LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / (TBN / WTN), wn = wave % (TBN / WTN);
  const int fr = lane & 15, fg = lane >> 4;
  const int S = d.splitk > 1 ? d.splitk : 1;
  const int z = d.zmode ? bz / S : 0;
  const int ks = bz % S;
  const nasrec_gemm_seg_t& s0 = d.seg[z];
  const int M = s0.M, N = s0.N;
  const int m0 = by * TBM, n0 = bx * TBN;
  if (m0 >= M || n0 >= N) return;

  // Segment table in LDS: the descriptor sits in kernel-argument memory, cold at every launch, and walking its segments with
  // scalar loads is a chain of dependent ~1 us misses (measured: a forward product costs ~5 us + 2.3 us per k-segment).  Here
  // lane q fetches segment q — one round trip for all of them — and the cursor below reads LDS.
  struct SegInfo {
    const float *A, *B, *Ax, *Bx;
    int K, lda, ldb, live;
  };
  __shared__ SegInfo sinfo[NASREC_MAX_SEGS];
  const bool use_table = !d.zmode && d.nseg > 1;  // a single segment / one z-problem per workgroup: read it straight from the arguments
  if (use_table) {
    if (tid < d.nseg) {
      const nasrec_gemm_seg_t& sg = d.seg[tid];
      sinfo[tid] = SegInfo{sg.A, sg.B, sg.Aaux, sg.Baux, sg.K, sg.lda, sg.ldb, (sg.A != nullptr && sg.K > 0) ? 1 : 0};
    }
    __syncthreads();
  }
  int T = 0;
  if (!use_table) {
    T = (s0.A != nullptr && s0.K > 0) ? (s0.K + TK - 1) / TK : 0;
  } else {
    for (int q = 0; q < d.nseg; ++q)
      if (sinfo[q].live) T += (sinfo[q].K + TK - 1) / TK;
  }
  const int t0 = (int)((long)T * ks / S), t1 = (int)((long)T * (ks + 1) / S);

  f32x4 acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- per-workgroup row predicates ------------------------------------------------------------------------------------
  const int cOnes = s0.ones_col;
  const int Ra = (s0.Mvalid > 0 && s0.Mvalid < M) ? s0.Mvalid : M;
  const int Rb = cOnes ? N - 1 : N;
  const bool edgeA = (m0 + TBM > Ra), edgeB = (n0 + TBN > Rb);
  bool rvA[NITA], rvB[NITB];  // the piece's FIRST row exists (pieces that start outside the operand are redirected to row 0)
  int kkA[NITA], kkB[NITB];
#pragma unroll
  for (int it = 0; it < NITA; ++it) {
    int rr;
    ring_stage_coords<AM, VA, NT, TK, TBM>(tid, it, rr, kkA[it]);
    rvA[it] = (m0 + rr) < Ra;
  }
#pragma unroll
  for (int it = 0; it < NITB; ++it) {
    int rr;
    ring_stage_coords<BMODE, VB, NT, TK, TBN>(tid, it, rr, kkB[it]);
    rvB[it] = (n0 + rr) < Rb;
  }

  // ---- fetch cursor (segment state) ------------------------------------------------------------------------------------
  int fs = z, fkt = t0;  // segment / k-tile of the NEXT tile to fetch
  if (use_table) {
    fs = 0;
    int skip = t0;
    while (fs < d.nseg) {
      const int nt = sinfo[fs].live ? (sinfo[fs].K + TK - 1) / TK : 0;
      if (skip < nt) break;
      skip -= nt;
      ++fs;
    }
    fkt = skip;
  }
  // staging loads are buffer loads: resource = the segment's operand (offsets are always in range: rows / k outside the operand
  // are redirected), or the NULL resource (num_records 0: returns 0 without touching memory) once the workgroup has no tile left
  // to fetch and for an absent ReLU-mask operand — which keeps the loop free of branches around loads
  const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsA = rs_null, rsB = rs_null, rsAx = rs_null, rsBx = rs_null;
  bool hasAaux = false, hasBaux = false;
  int cK = 0, seg_tiles = 0;
  int stepA = 0, stepB = 0;
  unsigned voffA[NITA], voffB[NITB];  // byte offset of the slot's (row, kk) at k-tile 0; rows outside the operand -> row 0
  unsigned koffA[NITA], koffB[NITB];  // byte offset contribution of kk: subtracting it redirects the slot to the tile's first k
  auto load_seg = [&](int sq) {
    // (uniform LDS reads land in VGPRs: readfirstlane makes them scalars again for the buffer resources)
    auto uptr = [](const float* p) -> const float* {
      const unsigned long long v = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return (const float*)(((unsigned long long)hi << 32) | lo);
    };
    struct {
      const float *A, *B, *Aaux, *Baux;
      int K, lda, ldb;
    } sg;
    if (use_table) {
      sg = {uptr(sinfo[sq].A), uptr(sinfo[sq].B), uptr(sinfo[sq].Ax), uptr(sinfo[sq].Bx), __builtin_amdgcn_readfirstlane(sinfo[sq].K),
            __builtin_amdgcn_readfirstlane(sinfo[sq].lda), __builtin_amdgcn_readfirstlane(sinfo[sq].ldb)};
    } else {
      const nasrec_gemm_seg_t& g = d.seg[sq];
      sg = {g.A, g.B, g.Aaux, g.Baux, g.K, g.lda, g.ldb};
    }
    // resources end with the operand's last element (offset of (rows - 1, K - 1) + 1): a 16-byte piece that starts inside the operand
    // and runs past its end gets zeros for the dwords beyond it (the range check of a raw buffer is per dword: tools/micro/
    // buffer_oob_probe.hip) instead of touching memory behind the tensor — operands may be plain torch tensors (opexec plans)
    const int Rb_mem = cOnes ? N - 1 : N;
    const int extA = (M > 0 && sg.K > 0) ? (int)(4 * (operand_offset<AM>(M - 1, sg.K - 1, sg.lda) + 1)) : 0;
    const int extB = (Rb_mem > 0 && sg.K > 0) ? (int)(4 * (operand_offset<BMODE>(Rb_mem - 1, sg.K - 1, sg.ldb) + 1)) : 0;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A), 0, extA, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.B), 0, extB, 0x00020000);
    hasAaux = AUX && sg.Aaux != nullptr;
    hasBaux = AUX && sg.Baux != nullptr;
    rsAx = hasAaux ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.Aaux), 0, extA, 0x00020000) : rs_null;
    rsBx = hasBaux ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.Baux), 0, extB, 0x00020000) : rs_null;
    cK = sg.K;
    seg_tiles = (cK + TK - 1) / TK;
    const int lda = sg.lda, ldb = sg.ldb;
    stepA = (int)(4 * operand_offset<AM>(0, TK, lda));
    stepB = (int)(4 * operand_offset<BMODE>(0, TK, ldb));
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      int rr, kk;
      ring_stage_coords<AM, VA, NT, TK, TBM>(tid, it, rr, kk);
      koffA[it] = 4u * (unsigned)operand_offset<AM>(0, kk, lda);
      voffA[it] = 4u * (unsigned)operand_offset<AM>(rvA[it] ? m0 + rr : 0, kk, lda);
    }
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
      int rr, kk;
      ring_stage_coords<BMODE, VB, NT, TK, TBN>(tid, it, rr, kk);
      koffB[it] = 4u * (unsigned)operand_offset<BMODE>(0, kk, ldb);
      voffB[it] = 4u * (unsigned)operand_offset<BMODE>(rvB[it] ? n0 + rr : 0, kk, ldb);
    }
  };

  float ra[RING][NITA][VA], rb[RING][NITB][VB];
  float xa[AUX ? RING : 1][AUX ? NITA : 1][VA], xb[AUX ? RING : 1][AUX ? NITB : 1][VB];
  auto stage_load = [](auto vtag, const __amdgpu_buffer_rsrc_t& rs, int voff, int soff, float* dst) {
    constexpr int V = decltype(vtag)::value;
    if (V == 4) {
      const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
      dst[0] = t[0], dst[1 % V] = t[1], dst[2 % V] = t[2], dst[3 % V] = t[3];
    } else {
      dst[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    }
  };
  using TagA = std::integral_constant<int, VA>;
  using TagB = std::integral_constant<int, VB>;
  int lim[RING];  // valid k of the tile held in the slot (>= TK: a full tile; 0: no tile)
  int fetched = t0;  // index of the next tile to fetch
  // one fetch path for full and partial tiles: a slot whose k lies beyond the segment reads the tile's first k instead (zeroed
  // when parked); past the workgroup's last tile the loads go to the null resource
  auto fetch = [&](int slot) {
    const bool live = fetched < t1;
    const int l = live ? cK - fkt * TK : 0;
    lim[slot] = l;
    const __amdgpu_buffer_rsrc_t ua = live ? rsA : rs_null, ub = live ? rsB : rs_null;
    const int sa = fkt * stepA, sb = fkt * stepB;
    // (a 16-byte piece whose first element lies in the operand may run past its last k / row: those elements are zeroed when parked)
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      const int o = (int)((kkA[it] < l) ? voffA[it] : voffA[it] - koffA[it]);
      stage_load(TagA{}, ua, o, sa, ra[slot][it]);
      if (AUX) stage_load(TagA{}, live ? rsAx : rs_null, o, sa, xa[slot][it]);
    }
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
      const int o = (int)((kkB[it] < l) ? voffB[it] : voffB[it] - koffB[it]);
      stage_load(TagB{}, ub, o, sb, rb[slot][it]);
      if (AUX) stage_load(TagB{}, live ? rsBx : rs_null, o, sb, xb[slot][it]);
    }
  };
  // advance the fetch cursor by one tile (the segment switch is a branch without loads)
  auto advance = [&]() {
    ++fetched;
    if (fetched >= t1) return;
    ++fkt;
    if (fkt >= seg_tiles) {
      if (use_table) {
        do {
          ++fs;
        } while (fs < d.nseg && !sinfo[fs].live);
        fkt = 0;
        load_seg(fs);
      }
    }
  };
  auto commit = [&](int slot) {
    const int l = lim[slot];
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      int rr, kk;
      ring_stage_coords<AM, VA, NT, TK, TBM>(tid, it, rr, kk);
      float a[VA];
#pragma unroll
      for (int e = 0; e < VA; ++e) {  // element e: (rr, kk + e) along k, (rr + e, kk) along r
        a[e] = ra[slot][it][e];
        if (AUX) a[e] = (!hasAaux || xa[slot][it][e] > 0.f) ? a[e] : 0.f;
        if (edgeA) a[e] = (m0 + rr + (KCA ? 0 : e) < Ra) ? a[e] : 0.f;
        a[e] = (kk + (KCA ? e : 0) < l) ? a[e] : 0.f;
      }
      if (VA == 4 && KCA) {
        *reinterpret_cast<f32x4*>(&As[rr * LDS_LD + kk]) = (f32x4){a[0], a[1 % VA], a[2 % VA], a[3 % VA]};
      } else {
#pragma unroll
        for (int e = 0; e < VA; ++e) As[(rr + (KCA ? 0 : e)) * LDS_LD + kk + (KCA ? e : 0)] = a[e];
      }
    }
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
      int rr, kk;
      ring_stage_coords<BMODE, VB, NT, TK, TBN>(tid, it, rr, kk);
      float b[VB];
#pragma unroll
      for (int e = 0; e < VB; ++e) {
        b[e] = rb[slot][it][e];
        if (AUX) b[e] = (!hasBaux || xb[slot][it][e] > 0.f) ? b[e] : 0.f;
        if (edgeB) b[e] = (n0 + rr + (KCB ? 0 : e) < Rb) ? b[e] : 0.f;
        const bool kin = kk + (KCB ? e : 0) < l;
        b[e] = kin ? b[e] : 0.f;
        if (cOnes && (n0 + rr + (KCB ? 0 : e) == N - 1)) b[e] = kin ? 1.f : 0.f;
      }
      if (VB == 4 && KCB) {
        *reinterpret_cast<f32x4*>(&Bs[rr * LDS_LD + kk]) = (f32x4){b[0], b[1 % VB], b[2 % VB], b[3 % VB]};
      } else {
#pragma unroll
        for (int e = 0; e < VB; ++e) Bs[(rr + (KCB ? 0 : e)) * LDS_LD + kk + (KCB ? e : 0)] = b[e];
      }
    }
  };
  auto mfma_tile = [&]() {
#pragma unroll
    for (int kb = 0; kb < TK / 16; ++kb) {
      f32x4 af[FA], bf[FB];
#pragma unroll
      for (int a = 0; a < FA; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[(wm * WTM + a * 16 + fr) * LDS_LD + kb * 16 + 4 * fg]);
#pragma unroll
      for (int b = 0; b < FB; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WTN + b * 16 + fr) * LDS_LD + kb * 16 + 4 * fg]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < FA; ++a)
#pragma unroll
          for (int b = 0; b < FB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
    }
  };

  if (t0 < t1) {
    load_seg(fs);
    // prologue: RING tiles in flight
#pragma unroll
    for (int r = 0; r < RING; ++r) {
      fetch(r);
      advance();
    }
    int t = t0;
    // steady state: whole groups of RING tiles, no branch around a load (refills past the last tile hit the null resource)
    for (; t + RING <= t1; t += RING) {
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        __syncthreads();
        commit(r);
        __syncthreads();
        fetch(r);  // refill the slot just parked with tile t + r + RING
        advance();
        mfma_tile();
      }
    }
    // remainder (< RING tiles, all already in flight)
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) {
      if (t + r < t1) {
        __syncthreads();
        commit(r);
        __syncthreads();
        mfma_tile();
      }
    }
  }

  if (S > 1) {
    float* slab = d.workspace + ((long)(z * S + ks)) * Mmax * Nmax;
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int i = m0 + wm * WTM + a * 16 + 4 * fg + r, j = n0 + wn * WTN + b * 16 + fr;
          if (i < M && j < N) slab[(long)i * N + j] = acc[a][b][r];
        }
    return;
  }
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      const int i0 = m0 + wm * WTM + a * 16 + 4 * fg, j = n0 + wn * WTN + b * 16 + fr;
      const float v4[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (i0 < M && j < N) epilogue_store_col4<CM>(d, s0, i0, j, M, v4);
    }
}
