// Per-sample chain kernel (NASREC_OP_SAMPLE_CHAIN, include/nasrec_hip.h): workgroup b runs a short run of sample-local
// forward operators for sample b back to back — the token-axis Linear (one 64x16 GEMM tile), the Transformer body, the FM /
// DotProduct cores, a segmented copy — instead of one launch each.  The stages are the very bodies of the stand-alone
// kernels (gemm_tile.h, attention_body.h, interact_bodies.h), so results are bit-identical; what disappears is one cold-L2
// kernel start (~5 us at batch 256) per fused stage.
#include "attention_body.h"
#include "gemm_tile.h"
#include "interact_bodies.h"

template <int TK>
__global__ __launch_bounds__(256) void sample_chain_kernel(const nasrec_chain_desc_t c, int Mmax, int Nmax, int copy_w) {
  __shared__ __attribute__((aligned(16))) float tri_lds[TRI_MAXK1 * TRI_LD];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll 1  // ONE copy of each stage body; the stage loop is uniform
  for (int s = 0; s < c.n; ++s) {
    {
      if (s > 0) {
        __threadfence_block();  // this workgroup's stores of the previous stage are visible to all of its waves
        __syncthreads();
      }
      switch (c.stage[s]) {
        case NASREC_OP_GEMM:
          gemm_tile<NASREC_AM_KC, NASREC_AM_TOKR, NASREC_CM_TOKJ, 256, TK, 64, 16>(c.gemm, Mmax, Nmax, b, 0, 0);
          break;
        case NASREC_OP_MHA_FWD:
          mha_fwd_sample<4>(c.mha, b);
          break;
        case NASREC_OP_FM_FWD:
          if (wave == 0) fm_fwd_sample(c.fm, b, lane);
          break;
        case NASREC_OP_DOT_TRI_FWD:
          if (wave == 0) dot_tri_fwd_sample(c.tri, b, lane, tri_lds);
          break;
        case NASREC_OP_COPY_SEGS:
          for (int j = tid; j < copy_w; j += 256) copy_segs_element(c.copy, b, j);
          break;
        default:
          break;
      }
    }
  }
}

int launch_sample_chain(hipStream_t st, const nasrec_chain_desc_t* c) {
  if (c->n < 1 || c->n > NASREC_CHAIN_MAX) return nasrec_set_error(-2, "sample_chain: n=%d", c->n);
  if (c->B < 1) return 0;
  int seen = 0, Mmax = 0, Nmax = 0, Kmax = 0, copy_w = 0;
  for (int s = 0; s < c->n; ++s) {
    const int k = c->stage[s];
    int bit = 0;
    if (k == NASREC_OP_GEMM) {
      const nasrec_gemm_desc_t& g = c->gemm;
      bit = 1;
      if (g.amode != NASREC_AM_KC || g.bmode != NASREC_AM_TOKR || g.cmode != NASREC_CM_TOKJ || g.zmode != 0 || g.splitk > 1 || g.nseg < 1 ||
          g.nseg > NASREC_MAX_SEGS)
        return nasrec_set_error(-2, "sample_chain: the GEMM stage must be a forward token-axis Linear without split-K");
      Mmax = g.seg[0].M;
      Nmax = g.seg[0].N;
      if (Mmax < 1 || Mmax > 64 || Nmax != 16 * c->B) return nasrec_set_error(-2, "sample_chain: GEMM stage M=%d N=%d for B=%d", Mmax, Nmax, c->B);
      for (int q = 0; q < g.nseg; ++q)
        if (g.seg[q].A && g.seg[q].K > Kmax) Kmax = g.seg[q].K;
    } else if (k == NASREC_OP_MHA_FWD) {
      bit = 2;
      if (c->mha.B != c->B || c->mha.N < 1 || c->mha.N > MHA_N) return nasrec_set_error(-2, "sample_chain: mha stage B=%d N=%d", c->mha.B, c->mha.N);
    } else if (k == NASREC_OP_FM_FWD) {
      bit = 4;
      if (c->fm.B != c->B) return nasrec_set_error(-2, "sample_chain: fm stage B=%d", c->fm.B);
    } else if (k == NASREC_OP_DOT_TRI_FWD) {
      bit = 8;
      if (c->tri.B != c->B || c->tri.k1 > TRI_MAXK1) return nasrec_set_error(-2, "sample_chain: dot_tri stage B=%d k1=%d", c->tri.B, c->tri.k1);
    } else if (k == NASREC_OP_COPY_SEGS) {
      bit = 16;
      if (c->copy.B != c->B) return nasrec_set_error(-2, "sample_chain: copy stage B=%d", c->copy.B);
      for (int q = 0; q < c->copy.nseg; ++q) copy_w = max(copy_w, c->copy.off[q] + c->copy.width[q]);
    } else {
      return nasrec_set_error(-2, "sample_chain: stage kind %d is not a sample-local forward operator", k);
    }
    if (seen & bit) return nasrec_set_error(-2, "sample_chain: two stages of kind %d", k);
    seen |= bit;
  }
  if (Kmax > 32)
    hipLaunchKernelGGL(sample_chain_kernel<64>, dim3(c->B), dim3(256), 0, st, *c, Mmax, Nmax, copy_w);
  else
    hipLaunchKernelGGL(sample_chain_kernel<32>, dim3(c->B), dim3(256), 0, st, *c, Mmax, Nmax, copy_w);
  return nasrec_check_launch("sample_chain");
}
