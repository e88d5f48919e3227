// Fused Transformer body (modules.py:664-686): nn.MultiheadAttention(embed 16, 8 heads x head_dim 2,
// batch_first, no dropout/mask) + residual + LayerNorm(16) + Linear(16,16) + ReLU + Linear(16,16) + residual +
// LayerNorm(16) (+ supernet token prefix mask).
//
// Mapping: one workgroup = one sample; lane = token (N <= 64), wave w owns the S-column slice [S*w, S*w+S) of every
// 16-wide vector — i.e. S/2 heads of the attention, S rows of every projection (template parameter S: 2 -> 8 waves, one
// head each; 4 -> 4 waves).  Full 16-vectors that a slice computation needs (the other waves' columns) go through LDS rows
// [token][16].  With one sample per wavefront only 256 of the chip's 1024 SIMDs had work at batch 256; 4 waves give every
// SIMD one wave; 8 waves give it two, so that one wave's LDS / dependent-FMA latency hides under the other's issue —
// which pays in the backward launch (long dependent chains) and not in the forward one (MHA_SLICE_FWD / _BWD).
// Attention at head_dim 2 has nothing for MFMA to chew on (K = 2): K/V slices are parked in LDS and every lane walks
// the keys (one ds_read_b128 per key for K, one for V; 2 FMA + 1 exp per key and head).
// The forward launch saves per-token state (NASREC_MHA_SAVED floats); the backward launch reads it, runs the
// flash-style two-phase attention backward (lane = query for dq, lane = key for dk/dv), and reduces the 1696
// parameter gradients of the node over the sample's tokens: weight gradients as LDS outer products (lane = one
// (row, column) entry of the wave's 4x16 slice), bias / LayerNorm gradients as wave reductions; per-sample
// partials are summed across the batch in fixed order by NASREC_OP_REDUCE_ROWS (deterministic).
#include "attention_tok.h"
#ifndef MHA_TOK
#define MHA_TOK 1  // token-major bodies (attention_tok.h) for the 4-wave launches; 0: the column-slice bodies of attention_body.h
#endif

template <int S>
__global__ __launch_bounds__(1024 / S) void mha_fwd_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float lds[MHA_FWD_LDS_FLOATS(S)];
  mha_fwd_sample<S>(d, blockIdx.x, lds);
}

template <int S>
__global__ __launch_bounds__(1024 / S) __attribute__((amdgpu_waves_per_eu(3))) void mha_bwd_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float lds[MHA_BWD_LDS_FLOATS(S)];
  mha_bwd_sample<S>(d, blockIdx.x, lds);
}

__global__ __launch_bounds__(256) void mha_fwd_tok_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float lds[MHA_TOK_FWD_LDS_FLOATS];
  mha_fwd_tok(d, blockIdx.x, lds);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void mha_bwd_tok_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float lds[MHA_TOK_BWD_LDS_FLOATS];
  mha_bwd_tok(d, blockIdx.x, lds);
}

int launch_mha(hipStream_t st, const nasrec_mha_desc_t* d) {
  if (d->N < 1 || d->N > MHA_N) return nasrec_set_error(-2, "mha: N=%d out of range [1,%d]", d->N, MHA_N);
  if (d->B == 0) return 0;
  if (d->kind == NASREC_OP_MHA_FWD) {
    if (MHA_TOK)
      hipLaunchKernelGGL(mha_fwd_tok_kernel, dim3(d->B), dim3(256), 0, st, *d);
    else
      hipLaunchKernelGGL(mha_fwd_kernel<MHA_SLICE_FWD>, dim3(d->B), dim3(1024 / MHA_SLICE_FWD), 0, st, *d);
  } else {
    if (d->saved == nullptr) return nasrec_set_error(-2, "mha backward needs the state saved by the forward launch (desc.saved)");
    // 8 waves per sample win where latency counts (batch 256: 23.5 against 26.7 us); at large batch the 4-wave form does the
    // same work with fewer wave-instructions per sample (B = 4096: 269 against 286 us)
    if ((d->B >= 1024 || d->bwd_form == 4) && MHA_TOK)
      hipLaunchKernelGGL(mha_bwd_tok_kernel, dim3(d->B), dim3(256), 0, st, *d);
    else if (d->B >= 1024 || d->bwd_form == 4)
      hipLaunchKernelGGL(mha_bwd_kernel<4>, dim3(d->B), dim3(256), 0, st, *d);
    else
      hipLaunchKernelGGL(mha_bwd_kernel<MHA_SLICE_BWD>, dim3(d->B), dim3(1024 / MHA_SLICE_BWD), 0, st, *d);
  }
  return nasrec_check_launch("mha");
}
