// Constants and helpers of the fused Transformer kernels (bodies: attention_tok.h; launchers: attention.hip; items of csrc/worklist.hip).
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

// Forward state kept for the backward: N * NASREC_MHA_SAVED floats per sample, as PLANES so that every access is a contiguous 16-byte
// piece per lane.  Round 6: 36 floats per token instead of 148 — what the backward cannot rebuild with a few MFMAs from x and the
// parameters (q, k, v, both LayerNorm x-hats, the LayerNorm-1 output and the FFN hidden layer are recomputed there: six 16 x 16 products
// per token on the matrix cores, bit-identical to the forward's because they are the same instructions on the same operands).
#define SV_O 0      // plane [token][16]: attention output (before the out-projection)
#define SV_M 1      // plane [token][16]: per-head softmax max (8) and 1/sum (8)
#define SV_STAT 2   // [token][4]: 1/std of LayerNorm 1 and 2, mean of LayerNorm 1 and 2
__device__ __forceinline__ float* sv_plane(float* saved, int b, int N, int p) { return saved + ((long)b * N) * NASREC_MHA_SAVED + (long)p * N * 16; }
__device__ __forceinline__ const float* sv_plane(const float* saved, int b, int N, int p) {
  return saved + ((long)b * N) * NASREC_MHA_SAVED + (long)p * N * 16;
}
static_assert(NASREC_MHA_SAVED == 36, "two planes of 16 and one of 4 floats per token");

// All 1696 parameters of the node are staged once per workgroup into LDS (6.8 KB; the backward parks the six matrices transposed).
static __device__ const int kParamOff[12] = {OFF_WIN, OFF_BIN, OFF_WOUT, OFF_BOUT, OFF_L1W, OFF_L1B,
                                             OFF_W1,  OFF_C1,  OFF_W2,   OFF_C2,   OFF_L2W, OFF_L2B};
static __device__ const int kParamLen[12] = {768, 48, 256, 16, 16, 16, 256, 16, 256, 16, 16, 16};

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
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

#ifdef MHA_STAMPS  // timing-only build (tools/mha_stamps.py): wave 0 keeps the clock at the stage boundaries and overwrites the first words of its parameter-gradient partial (backward) / output row (forward) with it
#define MHA_STAMP(i) mha_st[i] = (unsigned)__builtin_readcyclecounter()
#else
#define MHA_STAMP(i)
#endif
