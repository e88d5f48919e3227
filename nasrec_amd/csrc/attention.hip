// Fused Transformer body (modules.py:664-686): nn.MultiheadAttention(embed 16, 8 heads x head_dim 2,
// batch_first, no dropout/mask) + residual + LayerNorm(16) + Linear(16,16) + ReLU + Linear(16,16) + residual +
// LayerNorm(16) (+ supernet token prefix mask).
//
// Mapping (attention_tok.h): one workgroup of 4 waves = one sample; lane (r, g) of wave w owns token 16 w + r and columns 4 g .. 4 g + 3 of
// every 16-wide vector of it; the per-token 16 x 16 products run on the matrix cores (4 MFMAs per product and wave), the attention at
// head_dim 2 (nothing for MFMA: K = 2) as packed-fp32 loops over K / V rows parked in LDS.  The forward launch saves NASREC_MHA_SAVED = 36
// floats per token (attention output, softmax statistics, LayerNorm statistics); the backward launch recomputes the rest, runs the
// flash-style two-phase attention backward (the lane's token as query for dq, as key for dk / dv), and reduces the 1696 parameter
// gradients of the node over the sample's tokens (weight gradients as MFMA chains over the tokens, bias-like ones as DPP row sums);
// per-sample partials are summed across the batch in fixed order by NASREC_OP_REDUCE_ROWS (deterministic).
// (Rounds 1-5 ran a column-slice mapping — lane = token, wave = 2 or 4 columns of every vector, every intermediate through LDS and a
// barrier: forward 13.6 -> 9.7 us, backward 17.6 -> 13.2 us at batch 256 and 64 tokens with this one, before the saved state shrank.)
#include "attention_tok.h"

__global__ __launch_bounds__(256) void mha_fwd_tok_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float lds[MHA_TOK_FWD_LDS_FLOATS];
  mha_fwd_tok(d, blockIdx.x, lds);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void mha_bwd_tok_kernel(const nasrec_mha_desc_t d) {
  __shared__ __attribute__((aligned(16))) float lds[MHA_TOK_BWD_LDS_FLOATS];
  mha_bwd_tok(d, blockIdx.x, lds);
}

int launch_mha(hipStream_t st, const nasrec_mha_desc_t* d) {
  if (d->N < 1 || d->N > MHA_N) return nasrec_set_error(-2, "mha: N=%d out of range [1,%d]", d->N, MHA_N);
  if (d->B == 0) return 0;
  if (d->kind == NASREC_OP_MHA_FWD) {
    hipLaunchKernelGGL(mha_fwd_tok_kernel, dim3(d->B), dim3(256), 0, st, *d);
  } else {
    if (d->saved == nullptr) return nasrec_set_error(-2, "mha backward needs the state saved by the forward launch (desc.saved)");
    hipLaunchKernelGGL(mha_bwd_tok_kernel, dim3(d->B), dim3(256), 0, st, *d);
  }
  return nasrec_check_launch("mha");
}
