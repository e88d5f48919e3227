// Latency-regime GEMM body with RUN-TIME operand bindings: the loop, the ring of staged k-tiles, the fragment order and the
// epilogue of gemm_tile_ring (gemm_ring.h) — results are bit-identical for equal (TK, split-K) — but the four addressing modes of
// include/nasrec_hip.h are one stride formula evaluated when a k-segment is entered:
//     P(r, k) = p[(r >> 4) * R1 + (r & 15) * R0 + (k >> 4) * K1 + (k & 15) * K0]
//     KC   (R1, R0, K1, K0) = (16 ld, ld, 16, 1)      RC   = (16, 1, 16 ld, ld)
//     TOKR                  = (ld, 1, 256, 16)         TOKK = (256, 16, ld, 1)
// (every mode is linear in k across k-tiles because TK is a multiple of 16).  Only WHICH axis is contiguous stays a template
// parameter per operand (KCA / KCB: k-contiguous = KC, TOKK; else RC, TOKR) — it fixes the thread -> (row, k) map of the staging
// loads, and with it what is a compile-time constant per staged slot (leaving that to run time doubled the register count).  That makes the binding a property of a PROBLEM, not of
// a kernel: one launch can carry a dense Linear next to a token-axis Linear next to a weight gradient — what the heterogeneous
// launches of csrc/worklist.hip need (nasrec_amd/schedule.py puts independent operators of a batch-256 step side by side), with
// 3 contiguity pairs x 4 tiles x mask operand = 24 instantiations instead of 6 bindings x 16.  The LDS tiles live in a buffer handed in by the
// caller (the worklist kernel shares one buffer among all its bodies).
#pragma once
#include <type_traits>
#include "gemm_tile.h"

struct RtStride {
  int R1, R0, K1, K0;
  bool kcontig;  // k is the contiguous axis of the operand (staging lanes run along k), else r
};

__device__ __forceinline__ RtStride rt_stride(int mode, int ld) {
  if (mode == NASREC_AM_KC) return RtStride{16 * ld, ld, 16, 1, true};
  if (mode == NASREC_AM_RC) return RtStride{16, 1, 16 * ld, ld, false};
  if (mode == NASREC_AM_TOKR) return RtStride{ld, 1, 256, 16, false};
  return RtStride{256, 16, ld, 1, true};  // TOKK
}

__device__ __forceinline__ long rt_offset(const RtStride& s, int r, int k) {
  return (long)(r >> 4) * s.R1 + (long)(r & 15) * s.R0 + (long)(k >> 4) * s.K1 + (long)(k & 15) * s.K0;
}

// thread -> (row, k) of the staging loads of an R x TK operand tile: lanes run along the contiguous axis
// Operands are staged four floats per lane (one 16-byte load): along k for k-contiguous operands (parked with one ds_write_b128: a
// wave-instruction covers 4 rows x 256 B), along r for the others (RC is contiguous in r, TOKR in groups of 16; parked as four
// dwords in four LDS rows).  The one-dword form spent the texture-address path on 64 instructions per 16 KB tile.
// `it` counts LOADS: (rows x TK / NT) / 4 of them per thread and k-tile; (rr, kk) is the piece's first element.
template <bool KCONTIG, int NT, int TK, int R>
__device__ __forceinline__ void rt_stage_coords(int tid, int it, int& rr, int& kk) {
  if (KCONTIG) {
    kk = 4 * (tid & (TK / 4 - 1));
    rr = tid / (TK / 4) + (NT / (TK / 4)) * it;
  } else {
    rr = 4 * (tid & (R / 4 - 1));
    kk = tid / (R / 4) + (NT / (R / 4)) * it;
  }
}

__device__ __forceinline__ void rt_epilogue_store(int cm, const nasrec_gemm_desc_t& d, const nasrec_gemm_seg_t& sg, int i, int j, float v) {
  if (cm == NASREC_CM_PLAIN)
    epilogue_store<NASREC_CM_PLAIN>(d, sg, i, j, v);
  else
    epilogue_store<NASREC_CM_TOKJ>(d, sg, i, j, v);
}

__device__ __forceinline__ void rt_epilogue_store_col4(int cm, const nasrec_gemm_desc_t& d, const nasrec_gemm_seg_t& sg, int i0, int j, int M, const float (&v4)[4]) {
  if (cm == NASREC_CM_PLAIN)
    epilogue_store_col4<NASREC_CM_PLAIN>(d, sg, i0, j, M, v4);
  else
    epilogue_store_col4<NASREC_CM_TOKJ>(d, sg, i0, j, M, v4);
}

#define GEMM_RT_LDS_FLOATS(TBM, TBN, TK) (((TBM) + (TBN)) * ((TK) + 4) + 128)

template <bool KCA, bool KCB, int NT, int TK, int TBM, int TBN, bool AUX, int RING>
__device__ __forceinline__ void gemm_tile_rt(const nasrec_gemm_desc_t& d, int Mmax, int Nmax, const int bx, const int by, const int bz, float* lds) {
  const int AM = d.amode, BMODE = d.bmode, CM = d.cmode;
  constexpr int LDS_LD = TK + 4;
  constexpr int VA = 4, VB = 4;                                        // floats per staging load
  constexpr int NITA = TBM * TK / NT / VA, NITB = TBN * TK / NT / VB;  // staging loads per thread and k-tile
  static_assert((TBM * TK / NT) % VA == 0 && (TBN * TK / NT) % VB == 0 && TK % 16 == 0 && TBM % 4 == 0 && TBN % 4 == 0, "whole 16-byte pieces");
  constexpr int PER_WAVE = (TBM / 16) * (TBN / 16) / (NT / 64);
  constexpr int WTM = (PER_WAVE >= 2 && TBM >= 32) ? 32 : 16;
  constexpr int WTN = 16 * PER_WAVE / (WTM / 16);
  constexpr int FA = WTM / 16, FB = WTN / 16;
  static_assert(NITA >= 1 && NITB >= 1 && PER_WAVE >= 1 && (TBM / WTM) * (TBN / WTN) == NT / 64, "tile configuration");
  float* As = lds;
  float* Bs = lds + TBM * LDS_LD;
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
  static_assert(sizeof(SegInfo) * NASREC_MAX_SEGS <= 128 * sizeof(float), "segment table area");
  SegInfo* sinfo = reinterpret_cast<SegInfo*>(lds + (TBM + TBN) * LDS_LD);
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
    rt_stage_coords<KCA, NT, TK, TBM>(tid, it, rr, kkA[it]);
    rvA[it] = (m0 + rr) < Ra;
  }
#pragma unroll
  for (int it = 0; it < NITB; ++it) {
    int rr;
    rt_stage_coords<KCB, NT, TK, TBN>(tid, it, rr, kkB[it]);
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
    const int lda = sg.lda, ldb = sg.ldb;
    const RtStride sa_ = rt_stride(AM, lda), sb_ = rt_stride(BMODE, ldb);
    // resources end with the operand's last element (offset of (rows - 1, K - 1) + 1): a 16-byte piece that starts inside the operand
    // and runs past its end gets zeros for the dwords beyond it (the range check of a raw buffer is per dword: tools/micro/
    // buffer_oob_probe.hip) instead of touching memory behind the tensor
    const int Rb_mem = cOnes ? N - 1 : N;
    const int extA = (M > 0 && sg.K > 0) ? (int)(4 * (rt_offset(sa_, M - 1, sg.K - 1) + 1)) : 0;
    const int extB = (Rb_mem > 0 && sg.K > 0) ? (int)(4 * (rt_offset(sb_, Rb_mem - 1, sg.K - 1) + 1)) : 0;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A), 0, extA, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.B), 0, extB, 0x00020000);
    hasAaux = AUX && sg.Aaux != nullptr;
    hasBaux = AUX && sg.Baux != nullptr;
    rsAx = hasAaux ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.Aaux), 0, extA, 0x00020000) : rs_null;
    rsBx = hasBaux ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.Baux), 0, extB, 0x00020000) : rs_null;
    cK = sg.K;
    seg_tiles = (cK + TK - 1) / TK;
    stepA = (int)(4 * rt_offset(sa_, 0, TK));
    stepB = (int)(4 * rt_offset(sb_, 0, TK));
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      int rr, kk;
      rt_stage_coords<KCA, NT, TK, TBM>(tid, it, rr, kk);
      koffA[it] = 4u * (unsigned)rt_offset(sa_, 0, kk);
      voffA[it] = 4u * (unsigned)rt_offset(sa_, rvA[it] ? m0 + rr : 0, kk);
    }
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
      int rr, kk;
      rt_stage_coords<KCB, NT, TK, TBN>(tid, it, rr, kk);
      koffB[it] = 4u * (unsigned)rt_offset(sb_, 0, kk);
      voffB[it] = 4u * (unsigned)rt_offset(sb_, rvB[it] ? n0 + rr : 0, kk);
    }
  };

  float ra[RING][NITA][VA], rb[RING][NITB][VB];
  float xa[AUX ? RING : 1][AUX ? NITA : 1][VA], xb[AUX ? RING : 1][AUX ? NITB : 1][VB];
  auto stage_load = [](auto vtag, const __amdgpu_buffer_rsrc_t& rs, int voff, int soff, float* dst) {
    if (decltype(vtag)::value == 4) {
      const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
      dst[0] = t[0], dst[1 % decltype(vtag)::value] = t[1], dst[2 % decltype(vtag)::value] = t[2], dst[3 % decltype(vtag)::value] = t[3];
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
    // (a 16-byte piece whose first element lies in the segment may run past its last k / row: those elements are zeroed when parked)
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
      rt_stage_coords<KCA, NT, TK, TBM>(tid, it, rr, kk);
      float a[VA];
#pragma unroll
      for (int e = 0; e < VA; ++e) {  // element e: (rr, kk + e) along k, (rr + e, kk) along r
        a[e] = ra[slot][it][e];
        if (AUX) a[e] = (!hasAaux || xa[slot][it][e] > 0.f) ? a[e] : 0.f;
        if (edgeA) a[e] = (m0 + rr + (KCA ? 0 : e) < Ra) ? a[e] : 0.f;
        a[e] = (kk + (KCA ? e : 0) < l) ? a[e] : 0.f;
      }
      if (KCA) {
        *reinterpret_cast<f32x4*>(&As[rr * LDS_LD + kk]) = (f32x4){a[0], a[1], a[2], a[3]};
      } else {
#pragma unroll
        for (int e = 0; e < VA; ++e) As[(rr + e) * LDS_LD + kk] = a[e];
      }
    }
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
      int rr, kk;
      rt_stage_coords<KCB, NT, TK, TBN>(tid, it, rr, kk);
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
      if (KCB) {
        *reinterpret_cast<f32x4*>(&Bs[rr * LDS_LD + kk]) = (f32x4){b[0], b[1], b[2], b[3]};
      } else {
#pragma unroll
        for (int e = 0; e < VB; ++e) Bs[(rr + e) * LDS_LD + kk] = b[e];
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
      if (i0 < M && j < N) rt_epilogue_store_col4(CM, d, s0, i0, j, M, v4);
    }
}
