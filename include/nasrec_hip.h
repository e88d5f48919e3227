/*
 * nasrec_hip.h — C-ABI of the MI355X (gfx950) NASRec supernet engine.
 *
 * This is the drop-in boundary for ONE hot path of facebookresearch/NasRec: the supernet
 * forward / backward / optimizer step (reference: nasrec/supernet/{supernet,modules}.py and the step
 * body nasrec/utils/train_utils.py:255-287).  The reference has no FFI of its own (it is 100 % Python
 * on ATen); each entry point below replaces the ATen call sites named in its comment.  A maintainer
 * binds these with ctypes; nasrec_amd/_lib.py is that binding.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / HIP types.  `stream` is a hipStream_t passed as void*.
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch owns all tensors).
 *   - every function returns 0 on success, >0 = hipError_t, <0 = engine error; never throws, never
 *     exits, never synchronises the device.  nasrec_last_error() returns the text of the last failure
 *     on the calling thread.
 *   - all arithmetic is fp32 (indices int64).  GEMMs run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
 *   - "dense" tensors are [B, D] row-major with a row stride `ld` (floats);
 *     "sparse" tensors are [B, N, 16] with a batch stride `ld` (floats) so that token sub-ranges of a
 *     larger slab can be addressed without copies.
 *   - concatenations (supernet.py:570-573, 635-638) are never materialised: consumers take up to
 *     NASREC_MAX_SEGS segments.  A NULL segment pointer means "all zeros" (supernet.py:540-568).
 */
#ifndef NASREC_HIP_H
#define NASREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NASREC_MAX_SEGS 8
#define NASREC_MAX_TABLES 32
#define NASREC_EMB_DIM 16
#define NASREC_MHA_PARAMS 1696 /* floats of parameter gradient produced per Transformer node */
#define NASREC_MHA_SAVED 36    /* floats of forward state kept per token for the Transformer backward (rounds 1-5: 148) */

/* operand addressing modes of the GEMM family (see DESIGN.md "One GEMM, six bindings") */
enum {
  NASREC_AM_KC = 0,   /* P(r,k) = p[r*ld + k]                       k-contiguous rows              */
  NASREC_AM_RC = 1,   /* P(r,k) = p[r + k*ld]                       r-contiguous                   */
  NASREC_AM_TOKR = 2, /* P(r,k) = p[(r>>4)*ld + (r&15) + k*16]      r = (sample, e) of a [B,N,16]  */
  NASREC_AM_TOKK = 3  /* P(r,k) = p[(k>>4)*ld + (k&15) + r*16]      k = (sample, e) of a [B,N,16]  */
};
enum { NASREC_CM_PLAIN = 0, /* C(i,j) = c[i*ldc + j] */
       NASREC_CM_TOKJ = 1   /* C(i,j) = c[(j>>4)*ldc + (j&15) + i*16] */ };
enum { NASREC_ACT_NONE = 0, NASREC_ACT_RELU = 1, NASREC_ACT_SILU = 2, NASREC_ACT_SIGMOID = 3 };

/* op kinds (first int32 of every descriptor) */
enum {
  NASREC_OP_GEMM = 1,
  NASREC_OP_EMBED_GATHER = 2,
  NASREC_OP_DOT_TRI_FWD = 3,
  NASREC_OP_DOT_TRI_BWD = 4,
  NASREC_OP_FM_FWD = 5,
  NASREC_OP_FM_BWD = 6,
  NASREC_OP_MHA_FWD = 7,
  NASREC_OP_MHA_BWD = 8,
  NASREC_OP_REDUCE_ROWS = 9,
  NASREC_OP_COPY_SEGS = 10,
  NASREC_OP_GATE_BWD = 11,
  NASREC_OP_ROWSUM = 12,
  NASREC_OP_FINAL_FWD = 13,
  NASREC_OP_BCE = 14,
  NASREC_OP_FINAL_BWD = 15,
  NASREC_OP_EMB_DEDUP = 16,
  NASREC_OP_SUMSQ = 17,
  NASREC_OP_CLIP_COEF = 18,
  NASREC_OP_ADAGRAD_DENSE = 19,
  NASREC_OP_ADAGRAD_ROWS = 20,
  NASREC_OP_MEMSET = 21,
  NASREC_OP_LAYERNORM_FWD = 22,
  NASREC_OP_LAYERNORM_BWD = 23,
  NASREC_OP_ADD_SEGS = 24,
  NASREC_OP_SCALE = 25,
  NASREC_OP_ACT_BWD = 26,
  NASREC_OP_STAGE_INPUTS = 27,
  NASREC_OP_OPT_REDUCE = 28,
  NASREC_OP_OPT_APPLY = 29,
  NASREC_OP_WORKLIST = 30,
  NASREC_OP_CONST_I64 = 31,
  NASREC_OP_SPLITK_EPILOGUES = 32,
  NASREC_OP_DEDUP_IDS = 33,
  NASREC_OP_OPT_REDUCE2 = 34,
  NASREC_OP_FINAL_FUSED = 35,
  NASREC_OP_PERSIST = 36,
  NASREC_OP_WORKLIST_DEV = 38 /* (37 is taken by a layout-check slot of nasrec_desc_sizes) */
};

/* ------------------------------------------------------------------------------------------------
 * GEMM family.  C(i,j) = epilogue( sum_k A(i,k) * B(j,k) ).   Replaces every nn.Linear / LazyLinear
 * call site (modules.py:171,223,340,359,385,489,515,584,648,740; supernet.py:1140,1221) forward AND
 * the three autograd products behind it (x·Wᵀ, dy·W, dyᵀ·x), for both dense [B,D] operands and
 * token-axis operands ([B,N,16] transposed views, modules.py:222-234,358-361,648-650).
 * ---------------------------------------------------------------------------------------------- */
typedef struct nasrec_gemm_seg {
  const float* A;    /* NULL => this k-segment is all zeros (skipped) */
  const float* B;
  float* C;          /* used per z-problem (zmode=1); seg[0].C otherwise */
  const float* Aaux; /* optional: A(r,k) is taken as 0 where Aaux(r,k) <= 0 (fused ReLU backward) */
  const float* Baux; /* same for B */
  float* rowsum;     /* ones_col destination of THIS problem (overrides desc.rowsum_out when non-NULL): lets one zmode
                        launch carry the bias gradients of several independent weight-gradient products */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t Mvalid;    /* rows r >= Mvalid of A read as 0 (prefix mask on the row axis) */
  int32_t accumulate;/* zmode: C += result */
  int32_t ones_col;  /* 1: this problem's last column j = N-1 is virtual: B(N-1,k) = 1, and C(i,N-1) = sum_k A(i,k) is
                        written to rowsum[i] (or desc.rowsum_out[i]) instead of C (bias gradient fused into the weight-gradient product) */
  int32_t _pad;
} nasrec_gemm_seg_t;

typedef struct nasrec_gemm_desc {
  int32_t kind;      /* NASREC_OP_GEMM */
  int32_t amode, bmode, cmode;
  int32_t nseg;
  int32_t zmode;     /* 0: segments are K-ranges accumulated into seg[0].C (M,N from seg[0]);
                        1: segments are independent problems (blockIdx.z) */
  int32_t act;       /* NASREC_ACT_* applied to (acc + bias) */
  int32_t bias_on_rows;  /* bias index = i (token-axis linear) instead of j */
  int32_t mask_on_rows;  /* prefix mask index = i instead of j */
  int32_t dims_in_use;   /* < 0: no mask; else entries with index >= dims_in_use are written as 0
                            (CleverMaskGenerator, modules.py:57-96) */
  int32_t beta;      /* zmode=0: C += result */
  int32_t splitk;    /* >1: K is split over `splitk` workgroups per tile; partial slabs go to `workspace`, a second
                        pass sums them in fixed order and runs the epilogue.
                        NASREC_SPLITK_BALANCED (throughput-regime launches whose tiles all have the same K): the
                        k-iterations of the tiles that do not fill a round of 512 workgroups are shared equally by 512
                        workgroups, partial tiles go through `workspace` (>= NASREC_SK_WORKSPACE_FLOATS floats) and a
                        second pass sums them in ascending k order (deterministic) */
  const float* bias;
  float* save_z;     /* optional store of (acc+bias), addressed like C */
  float* save_act;   /* optional store of act(acc+bias), addressed like C */
  /* optional elementwise multiplier R(i,j) given as column segments of dense tensors (SigmoidGating,
     modules.py:576-582): out = act(z) * R, R = 0 outside every segment */
  const float* mul_ptr[NASREC_MAX_SEGS];
  int32_t mul_off[NASREC_MAX_SEGS], mul_width[NASREC_MAX_SEGS], mul_ld[NASREC_MAX_SEGS];
  int32_t mul_nseg;
  int32_t defer_second_pass; /* splitk > 1 only: this launch just writes its slabs; a NASREC_OP_SPLITK_EPILOGUES launch runs the
                                second passes of up to NASREC_EPILOGUES_MAX such launches at once (batch 256: every launch on
                                the critical path costs ~5 us whatever it does) */
  float* workspace;  /* splitk slabs: splitk*M*N floats */
  int32_t* counters; /* reserved (must be NULL) */
  float* rowsum_out; /* destination of the ones_col row sums */
  const float* pre_add; /* optional, addressed like C: added to the accumulator BEFORE bias/activation — lets a
                           K-range longer than NASREC_MAX_SEGS segments be chained over several launches */
  nasrec_gemm_seg_t seg[NASREC_MAX_SEGS];
} nasrec_gemm_desc_t;

/* ------------------------------------------------------------------------------------------------
 * Embedding stem.  out[b,f,:] = table_f[idx[b,f],:]  — replaces Fs × nn.Embedding + torch.stack
 * (supernet.py:404-430).  Bit-exact copy.
 * ---------------------------------------------------------------------------------------------- */
typedef struct nasrec_embed_desc {
  int32_t kind; /* NASREC_OP_EMBED_GATHER */
  int32_t B, Fs;
  int32_t _pad;
  const int64_t* idx;                      /* [B, Fs] */
  const float* table[NASREC_MAX_TABLES];   /* table f: [rows_f, 16] */
  int64_t rows[NASREC_MAX_TABLES];
  float* out;                              /* [B, Fs, 16] */
  int32_t* oob;                            /* optional: set to 1 if any index is out of range */
} nasrec_embed_desc_t;

/* Row-sparse embedding backward: per (b,f) decide whether b is the first occurrence ("leader") of its row
 * within field f and, for leaders, sum the gradients of all occurrences in ascending b (deterministic).
 * Mathematically identical to embedding_dense_backward + dense Adagrad when weight_decay == 0, because an
 * untouched row has g = 0 and Adagrad(g=0) is a no-op (train_utils.py:283-286; SURVEY §7 hard parts). */
typedef struct nasrec_emb_dedup_desc {
  int32_t kind; /* NASREC_OP_EMB_DEDUP */
  int32_t B, Fs;
  int32_t _pad;
  const int64_t* idx;   /* [B, Fs] */
  const float* dout;    /* [B, Fs, 16] */
  int32_t* leader;      /* [B, Fs] out: 1 if leader */
  float* gsum;          /* [B, Fs, 16] out: summed gradient (valid where leader) */
  float* sumsq_partial; /* [Fs * ceil(B/256)] out: sum of squares of leader gradients per workgroup */
  int32_t* overflow;    /* optional: set to 1 if a hash partition of the large-batch merge ran out of slots (never seen with
                           real id distributions; the step's result must then be discarded) */
  int32_t rank_B;       /* layout of `dout` as in nasrec_adagrad_rows_desc_t (0: one contiguous [B,Fs,16] array; > 0: the receive */
  int32_t _pad2;        /* buffer of an all-gather, rank_B samples per rank chunk, chunks rank_stride floats apart) — gsum stays contiguous */
  int64_t rank_stride;
} nasrec_emb_dedup_desc_t;

/* ------------------------------------------------------------------------------------------------
 * DotProduct core (modules.py:366-383): T[B,k1,16] -> out[B, k1(k1-1)/2], out[b, i(i-1)/2 + j] =
 * <T[b,i,:], T[b,j,:]>, i>j  (== Z[:, li, lj] with torch.tril_indices(k1,k1,-1)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct nasrec_dot_tri_desc {
  int32_t kind; /* NASREC_OP_DOT_TRI_FWD / _BWD */
  int32_t B, k1;
  int32_t ld_out;   /* row stride of out / dout */
  const float* T;   /* [B,k1,16] contiguous */
  float* out;       /* fwd: out */
  const float* dout;/* bwd */
  float* dT;        /* bwd: [B,k1,16] contiguous, overwritten */
} nasrec_dot_tri_desc_t;

/* FactorizationMachine3D core (modules.py:736-738): ix[b,e] = (sum_n x)^2 - sum_n x^2. */
typedef struct nasrec_fm_desc {
  int32_t kind; /* NASREC_OP_FM_FWD / _BWD */
  int32_t B, N;
  int32_t ldx;      /* batch stride of x / dx */
  int32_t ld_ix;    /* row stride of ix / dix */
  int32_t accumulate; /* fwd: ix += ; bwd: dx += */
  const float* x;
  float* ix;        /* fwd out [B,16] */
  const float* dix; /* bwd in  [B,16] */
  float* dx;        /* bwd out [B,N,16] */
  const float* add; /* fwd, optional [B,16] with row stride ld_ix: ix = add + fm (instead of accumulating in place) */
} nasrec_fm_desc_t;

/* ------------------------------------------------------------------------------------------------
 * Transformer body after the token projection (modules.py:664-686): nn.MultiheadAttention(16, 8 heads,
 * head_dim 2) + residual + LN + FC(16,16)+ReLU+FC(16,16) + residual + LN, optional token prefix mask on
 * the output.  One workgroup of four wavefronts per sample, lane = (token, 4 columns); everything between x and
 * out stays in registers / LDS, the per-token 16 x 16 products run on the matrix cores (csrc/attention_tok.h).
 * `params` points to the 12 parameter tensors in this order: in_proj_weight[48,16], in_proj_bias[48],
 * out_proj.weight[16,16], out_proj.bias[16], attn_ln.weight[16], attn_ln.bias[16], fc1.weight[16,16],
 * fc1.bias[16], fc2.weight[16,16], fc2.bias[16], fc_ln.weight[16], fc_ln.bias[16]  (12 pointers).
 * Backward: reads the state the forward launch saved (`saved`), recomputes the rest of the forward from x, writes dx
 * and per-sample parameter-gradient partials [B, NASREC_MHA_PARAMS] that NASREC_OP_REDUCE_ROWS sums in fixed order
 * (deterministic).
 * ---------------------------------------------------------------------------------------------- */
typedef struct nasrec_mha_desc {
  int32_t kind; /* NASREC_OP_MHA_FWD / _BWD */
  int32_t B, N;
  int32_t ldx, ldo;     /* batch strides of x/dx and out/dout */
  int32_t dims_in_use;  /* < 0 none; else output tokens >= dims_in_use are zero (modules.py:678-686) */
  const float* x;
  float* out;
  const float* dout;
  float* dx;            /* overwritten */
  float* dparams_partial; /* [B, NASREC_MHA_PARAMS] */
  const float* params[12];
  float* saved;         /* optional [B, N * NASREC_MHA_SAVED]: per sample the planes [N][16] attention output, [N][16] softmax max /
                           1/sum per head, [N][4] 1/std and mean of both LayerNorms, written by the forward launch; the backward
                           launch reads them and recomputes q, k, v, the x-hats and the FFN hidden layer from x (six 16 x 16
                           products per token on the matrix cores) instead of reading 112 more floats per token */
  int32_t partial_ld;   /* bwd: row stride of dparams_partial in floats (0 = NASREC_MHA_PARAMS); lets several Transformer nodes
                           share one partial buffer [B, n * NASREC_MHA_PARAMS] that ONE NASREC_OP_REDUCE_ROWS launch sums */
  int32_t bwd_form;     /* ignored since ABI 16 (one backward form: 4 waves per sample, token-major; rounds 3-5: 0 = the launcher
                           picks 8 or 4 waves per sample by the batch, 4 = the form a worklist launch runs) */
} nasrec_mha_desc_t;

/* out[c] = sum_r in[r*ld + c], r < R, c < C, fixed order.  Optionally scattered to up to NASREC_REDUCE_MAX_DST destination
 * tensors (parameter .grad buffers) by column ranges. */
#define NASREC_REDUCE_MAX_DST 48
typedef struct nasrec_reduce_rows_desc {
  int32_t kind; /* NASREC_OP_REDUCE_ROWS */
  int32_t R, C, ld;
  const float* in;
  int32_t ndst;
  int32_t _pad;
  float* dst[NASREC_REDUCE_MAX_DST];
  int32_t dst_off[NASREC_REDUCE_MAX_DST];   /* first column of each destination */
  int32_t dst_len[NASREC_REDUCE_MAX_DST];
} nasrec_reduce_rows_desc_t;

/* Strided copy / accumulate of a segmented dense view:  dst[b, j] (=|+=) concat_j(seg)[b, j].
 * Used where the reference reuses a tensor without a projection (modules.py:343,352,362,388; supernet.py
 * 1145-1146) and for gradient fan-in of such aliases.  `reverse`=1 scatters dst-shaped gradient back:
 * seg[b, j-off] += src[b, j]. */
typedef struct nasrec_copy_segs_desc {
  int32_t kind; /* NASREC_OP_COPY_SEGS */
  int32_t B, nseg;
  int32_t ld_dst;
  int32_t accumulate;
  int32_t reverse;
  float* dst;                         /* forward: destination; reverse: source of gradients (read only) */
  float* seg[NASREC_MAX_SEGS];        /* forward: sources (NULL = zeros); reverse: gradient destinations (NULL = skip) */
  int32_t width[NASREC_MAX_SEGS], ld[NASREC_MAX_SEGS], off[NASREC_MAX_SEGS];
  int32_t seg_accumulate[NASREC_MAX_SEGS]; /* reverse: += instead of = */
} nasrec_copy_segs_desc_t;

/* out[b,j] = (sum over segments covering j) — elementwise sum of two segmented dense views
 * (Sum without projection, modules.py:487-491). Implemented through COPY_SEGS with accumulate. */

/* SigmoidGating backward elementwise part (modules.py:578-582): given dout (grad of g*R), g = sigmoid(z)
 * and R (segmented): dz = dout * R * g * (1-g)   (written to dz),  dR_seg += dout * g. */
typedef struct nasrec_gate_bwd_desc {
  int32_t kind; /* NASREC_OP_GATE_BWD */
  int32_t B, D;
  int32_t ld_dout, ld_g, ld_dz;
  const float* dout;
  const float* g;
  float* dz;
  int32_t nseg;
  const float* r_ptr[NASREC_MAX_SEGS];
  float* dr_ptr[NASREC_MAX_SEGS];        /* NULL = no gradient wanted */
  int32_t r_off[NASREC_MAX_SEGS], r_width[NASREC_MAX_SEGS], r_ld[NASREC_MAX_SEGS];
  int32_t dr_accumulate[NASREC_MAX_SEGS];
} nasrec_gate_bwd_desc_t;

/* Bias gradients: out[r] = sum_k P(r,k) with P addressed by `mode` (RC for dense dy[M,N] -> db[N];
 * TOKK for token-axis dy[B,N',16] -> db[N']), optional fused ReLU mask (aux) and prefix mask (rvalid). */
typedef struct nasrec_rowsum_desc {
  int32_t kind; /* NASREC_OP_ROWSUM */
  int32_t mode, R, K, ld;
  int32_t rvalid;   /* rows >= rvalid produce 0 */
  const float* p;
  const float* aux;
  float* out;
} nasrec_rowsum_desc_t;

/* Final logit (supernet.py:592-598 / 657-664): logits[b] = <feats[b,:], w> + bias, feats = segments. */
typedef struct nasrec_final_desc {
  int32_t kind; /* NASREC_OP_FINAL_FWD / _BWD / _FUSED */
  int32_t B, nseg;
  float grad_scale;    /* bwd with y != NULL: see nasrec_bce_desc_t */
  const float* w;      /* [1, K] */
  const float* bias;   /* [1] */
  float* logits;       /* [B] */
  const float* dlogits;/* bwd in [B] */
  float* dw;           /* bwd out [K], overwritten */
  float* dbias;        /* bwd out [1] */
  const float* seg[NASREC_MAX_SEGS];
  float* dseg[NASREC_MAX_SEGS];  /* bwd: gradient destinations (NULL = skip) */
  int32_t width[NASREC_MAX_SEGS], ld[NASREC_MAX_SEGS], off[NASREC_MAX_SEGS];
  int32_t dseg_accumulate[NASREC_MAX_SEGS];
  /* bwd, optional: fuse BCEWithLogitsLoss into this launch (one kernel less on the step's critical path).  With
     y != NULL the incoming gradient is (sigmoid(logits[b]) - y[b]) * grad_scale instead of dlogits[b]; one extra
     workgroup writes loss[0] (mean BCE, as NASREC_OP_BCE) and, if dlogits_out != NULL, the per-sample gradient. */
  const float* y;
  float* loss;
  float* dlogits_out;
  /* bwd, large batches: nsplit > 1 cuts the batch into nsplit slices, one set of workgroups each; dw is then a partial buffer
     [nsplit, K + 1] (column K = the bias gradient, dbias is unused) that a NASREC_OP_REDUCE_ROWS launch sums in fixed order */
  int32_t nsplit;
  /* NASREC_OP_FINAL_FUSED (round 5; needs y != NULL, nsplit <= 1) = the forward AND the per-sample part of the backward in one launch: the
     workgroup that has summed sample b's logit writes it, derives (sigmoid(logit) - y[b]) * grad_scale and writes dseg[b, :] — the
     gradients the rest of the backward pass waits for — one launch boundary earlier.  The parts that need every sample's logit (dw, dbias,
     the loss, dlogits_out) stay a NASREC_OP_FINAL_BWD with dseg_done = 1: it skips the dseg part.  Same expression per element: same bits. */
  int32_t dseg_done;
  /* last_n_blocks_out > 1 (supernet.py:592-596 / 657-661): the reference concatenates the last blocks' sparse outputs on the LAST
     dim ([B, N, n*16]) before flattening, so block j's token t sits at weight columns off + t*(n*16) + [0, 16).  tok_stride[q] != 0
     gives such a segment: feature j of the (contiguous [B, N*16]) segment meets weight w[off[q] + (j / 16) * tok_stride[q] + j % 16];
     0 = the plain contiguous placement w[off[q] + j]. */
  int32_t tok_stride[NASREC_MAX_SEGS];
} nasrec_final_desc_t;

/* BCEWithLogitsLoss(mean) forward + dlogits (main_train.py:122; train_utils.py:266):
 * loss = mean(max(z,0) - z*y + log1p(exp(-|z|))), dlogits[b] = (sigmoid(z_b) - y_b) * grad_scale. */
typedef struct nasrec_bce_desc {
  int32_t kind; /* NASREC_OP_BCE */
  int32_t B;
  float grad_scale;   /* 1/B for a single process, 1/(B*world) under data parallel */
  int32_t _pad;
  const float* logits;
  const float* y;
  float* loss;        /* [1] */
  float* dlogits;     /* [B] */
} nasrec_bce_desc_t;

/* A chunk table restricts a flat-arena op to the parameters of ONE sampled path (a weight-sharing supernet step trains a
 * fraction of the arena; torch skips parameters whose grad is None, train_utils.py:285-286 — touching only the path's ranges
 * is the same arithmetic on a quarter of the bytes): chunks = device array of nchunks (offset, count) int64 pairs in units of
 * 4-byte elements from the op's base pointer, offsets 16-byte aligned; chunks == NULL means the whole buffer [0, n).
 * NASREC_OP_CONST_I64 writes such a table from the descriptor itself (stream-ordered, no host buffer to keep alive). */
#define NASREC_SPLITK_BALANCED (-1)
#define NASREC_SK_WORKSPACE_FLOATS (512L * 3 * 128 * 128)
#define NASREC_CHUNK_ELEMS 65536
#define NASREC_CONST_I64_MAX 448

/* sum of squares of a flat fp32 buffer -> partial[nblocks] (deterministic two-stage). */
typedef struct nasrec_sumsq_desc {
  int32_t kind; /* NASREC_OP_SUMSQ */
  int32_t nblocks;
  int64_t n;
  const float* x;
  float* partial;
  const int64_t* chunks; /* optional chunk table */
  int64_t nchunks;
} nasrec_sumsq_desc_t;

/* clip_grad_norm_ coefficient (train_utils.py:285): total = sqrt(sum partials), coef = min(1, max_norm /
 * (total + 1e-6)); out[0] = coef, out[1] = total.  max_norm <= 0 => coef = 1. */
typedef struct nasrec_clip_coef_desc {
  int32_t kind; /* NASREC_OP_CLIP_COEF */
  int32_t n_a, n_b;
  float max_norm;
  const float* partial_a;
  const float* partial_b;
  float* out;
} nasrec_clip_coef_desc_t;

/* torch.optim.Adagrad(lr, eps, lr_decay=0, weight_decay=0) on a flat buffer (main_train.py:152-154):
 * g' = g*coef; state += g'^2; p -= lr * g' / (sqrt(state) + eps).  lr and coef are read from device
 * memory so a captured graph can be replayed with a new learning rate (lr_schedule.py). */
typedef struct nasrec_adagrad_dense_desc {
  int32_t kind; /* NASREC_OP_ADAGRAD_DENSE */
  float eps;
  int64_t n;
  float* p;
  const float* g;
  float* state;
  const float* lr;    /* device scalar */
  const float* coef;  /* device scalar (clip) */
  const int64_t* chunks; /* optional chunk table (offsets into p / g / state alike) */
  int64_t nchunks;
} nasrec_adagrad_dense_desc_t;

typedef struct nasrec_adagrad_rows_desc {
  int32_t kind; /* NASREC_OP_ADAGRAD_ROWS */
  int32_t B, Fs;
  float eps;
  const int64_t* idx;      /* [B,Fs] */
  const int32_t* leader;   /* [B,Fs] */
  const float* gsum;       /* [B,Fs,16] */
  float* table[NASREC_MAX_TABLES];
  float* state[NASREC_MAX_TABLES];
  const float* lr;
  const float* coef;
  int64_t rows[NASREC_MAX_TABLES]; /* rows of table f: an id outside [0, rows) is skipped (the gather has already flagged it;
                                      torch would have raised IndexError in the forward pass) */
  int32_t rank_B;       /* 0: gsum is one contiguous [B,Fs,16] array.  > 0: gsum is the receive buffer of an all-gather whose per-rank */
  int32_t _pad;         /*    chunks hold rank_B samples each and start rank_stride floats apart (something else rides behind the rows): */
  int64_t rank_stride;  /*    row (b,f) lives at gsum + (b / rank_B) * rank_stride + ((b % rank_B) * Fs + f) * 16 */
} nasrec_adagrad_rows_desc_t;

/* Fused optimizer tail for batch <= 256 (the two grid-wide dependencies of clip_grad_norm_ + Adagrad need two
 * launches, not five):
 *   OPT_REDUCE = EMB_DEDUP (workgroups [0, Fs)) + SUMSQ of the dense gradient arena (the remaining workgroups);
 *   OPT_APPLY  = every workgroup re-derives the clip coefficient from the partial sums (same fixed-order fp64 sum as
 *                CLIP_COEF, so all workgroups agree bit-for-bit; workgroup 0 also writes clip.out), then ADAGRAD_DENSE
 *                (workgroups [0, dense_blocks)) and ADAGRAD_ROWS (the remaining workgroups).
 * The embedded descriptors have the meaning of their stand-alone ops; their `kind`, `dense.coef` and `rows.coef` fields
 * are ignored. */
typedef struct nasrec_opt_reduce_desc {
  int32_t kind; /* NASREC_OP_OPT_REDUCE */
  int32_t _pad;
  nasrec_emb_dedup_desc_t dedup; /* B <= 256 */
  nasrec_sumsq_desc_t sumsq;
} nasrec_opt_reduce_desc_t;

typedef struct nasrec_opt_apply_desc {
  int32_t kind; /* NASREC_OP_OPT_APPLY */
  int32_t dense_blocks;
  nasrec_clip_coef_desc_t clip;
  nasrec_adagrad_dense_desc_t dense;
  nasrec_adagrad_rows_desc_t rows;
} nasrec_opt_apply_desc_t;

/* ------------------------------------------------------------------------------------------------
 * The row-sparse embedding backward in two halves (round 5).  NASREC_OP_EMB_DEDUP above does everything behind the backward pass;
 * but which sample leads a row, and which samples repeat it, depends on the IDS only — known before the forward pass starts (one
 * GPU: when the batch is staged; N GPUs: when the ids all-gather lands, nasrec_amd/parallel.py).  So:
 *   NASREC_OP_DEDUP_IDS   (any time after the ids are known; B <= NASREC_DEDUP_IDS_MAX_B) per field.  A RUN = the samples of one id in
 *                         ascending order, a SUB-RUN = the part of a run inside one 256-sample chunk.  Writes leader[b,f] (0 duplicate,
 *                         1 leader without duplicates, 2 leader with duplicates) and
 *                         B <= 256 (one workgroup per field, all-pairs match masks): order[f][..] = the runs that have duplicates, each
 *                           contiguous and ascending; lists[f][k] = start in `order` | length << 16 | bit 31; counts[f][0] = their number;
 *                         B > 256 (one workgroup per field and chunk, all pairs against the field's ids, nothing crosses workgroups):
 *                           order[f][256 c + ..] = the members of chunk c's sub-runs with >= 2 members; lists[f][b], per SAMPLE = bit 16
 *                           (b heads such a sub-run) | position inside the chunk's region | (members - 1) << 8 | bit 31 (the sub-run is its
 *                           whole run) | bit 17 (b leads a run with sub-runs in later chunks); heads[f][b] = the head of the NEXT sub-run
 *                           of b's run (-1: none): the chain leader -> next -> ... is the run's sub-runs in chunk order.
 *                         NASREC_OP_STAGE_INPUTS can carry it (stage.dedup_ids.order != NULL, B <= 256).
 *   NASREC_OP_OPT_REDUCE2 (behind the backward pass) workgroups [0, Fs): per field, IN PLACE over the per-sample row gradients, the sum of
 *                         every sub-run into its first row in ascending sample order, then of every multi-chunk run's sub-run sums into the
 *                         leader's row in chunk order — the summation order of NASREC_OP_EMB_DEDUP at every batch size, bit for bit — and
 *                         the sum of squares of the leaders with duplicates; workgroups [Fs, Fs + row_blocks): sum of squares of the
 *                         leaders without; the remaining sumsq.nblocks workgroups: NASREC_OP_SUMSQ of the dense gradient arena.
 *                         (256-thread workgroups for B <= 256, 1024-thread workgroups above: a field's rows are staged in LDS at once.)
 * NASREC_OP_OPT_APPLY then reads leader / rows as before (rows.gsum = the same array: a leader's row now holds its sum).
 * ---------------------------------------------------------------------------------------------- */
#define NASREC_DEDUP_IDS_MAX_B 2048
typedef struct nasrec_dedup_ids_desc {
  int32_t kind; /* NASREC_OP_DEDUP_IDS */
  int32_t B, Fs;
  int32_t cap;          /* entries per field of order / of each list: a power of two >= B, 256 <= cap <= NASREC_DEDUP_IDS_MAX_B */
  const int64_t* idx;   /* [B, Fs] */
  int32_t* leader;      /* [B, Fs] out */
  int32_t* order;       /* [Fs, cap] out */
  int32_t* lists;       /* [Fs, cap] out */
  int32_t* counts;      /* [Fs, 2] out */
  int32_t* heads;       /* [Fs, cap] out: next-sub-run links (B > 256 only; may be NULL for B <= 256) */
} nasrec_dedup_ids_desc_t;

typedef struct nasrec_opt_reduce2_desc {
  int32_t kind; /* NASREC_OP_OPT_REDUCE2 */
  int32_t B, Fs, cap;
  int32_t rank_B;        /* layout of `rows` as in nasrec_adagrad_rows_desc_t (0: contiguous) */
  int32_t row_blocks;    /* workgroups of the pass over leaders without duplicates (>= 1) */
  int64_t rank_stride;
  float* rows;           /* [B, Fs, 16] per-sample row gradients, summed in place */
  const int32_t* leader;
  const int32_t* order;
  const int32_t* lists;
  const int32_t* counts;
  const int32_t* heads;
  float* sumsq_partial;  /* [Fs + row_blocks] out */
  nasrec_sumsq_desc_t sumsq; /* dense arena (sumsq.nblocks workgroups; 0: none) */
} nasrec_opt_reduce2_desc_t;

typedef struct nasrec_memset_desc {
  int32_t kind; /* NASREC_OP_MEMSET */
  int32_t _pad;
  int64_t bytes;
  void* ptr;
  const int64_t* chunks; /* optional chunk table: zero only these ranges of ptr (4-byte elements) */
  int64_t nchunks;
} nasrec_memset_desc_t;

/* the second passes (fixed-order slab sums + epilogue) of up to 3 split-K GEMM launches whose descriptors carry
 * defer_second_pass = 1 (CM_PLAIN outputs), as ONE launch: g[] are copies of those descriptors */
#define NASREC_EPILOGUES_MAX 3
typedef struct nasrec_splitk_epilogues_desc {
  int32_t kind; /* NASREC_OP_SPLITK_EPILOGUES */
  int32_t n;
  nasrec_gemm_desc_t g[NASREC_EPILOGUES_MAX];
} nasrec_splitk_epilogues_desc_t;

/* dst[0, n) = vals[0, n): small integer tables (chunk tables) travel in the kernel arguments */
typedef struct nasrec_const_i64_desc {
  int32_t kind; /* NASREC_OP_CONST_I64 */
  int32_t n;    /* <= NASREC_CONST_I64_MAX */
  int64_t* dst;
  int64_t vals[NASREC_CONST_I64_MAX];
} nasrec_const_i64_desc_t;

/* LayerNorm over the last axis of a dense [R,D] view (mode KC) or over the token axis of a [B,N',16]
 * tensor per (b,e) (mode TOKR: row r=(b,e), element i=n').  y = mask(act(LN(x)*w + b)) (modules.py:174-178,
 * 226-230, 341, 360, 392, 493, 589, 649, 741; supernet.py:1141). Saves mean / rstd per row for backward. */
typedef struct nasrec_layernorm_desc {
  int32_t kind; /* NASREC_OP_LAYERNORM_FWD / _BWD */
  int32_t mode, R, D;
  int32_t ldx, ldy;
  int32_t act, dims_in_use;
  int32_t accumulate;  /* fwd: y += ; bwd: dx += */
  float eps;
  const float* x;
  const float* w;
  const float* b;
  float* y;
  float* stats;        /* [R,2] mean, rstd */
  const float* dy;     /* bwd */
  float* dx;           /* bwd */
  float* dwb_partial;  /* bwd: [nblk, 2*D] partials of (dw, db); reduce with NASREC_OP_REDUCE_ROWS */
  int32_t nblk;
  int32_t _pad;
} nasrec_layernorm_desc_t;

/* y[i] = x[i] * s  */
typedef struct nasrec_scale_desc {
  int32_t kind; /* NASREC_OP_SCALE */
  int32_t _pad;
  int64_t n;
  float s;
  int32_t _pad2;
  const float* x;
  float* y;
} nasrec_scale_desc_t;

/* dz = dy * act'(z) * prefixmask, dense [R,D] views (SiLU path; ReLU is fused into the GEMM loads) */
typedef struct nasrec_act_bwd_desc {
  int32_t kind; /* NASREC_OP_ACT_BWD */
  int32_t mode, R, D;
  int32_t ld_dy, ld_z, ld_dz;
  int32_t act, dims_in_use;
  const float* dy;
  const float* z;
  float* dz;
} nasrec_act_bwd_desc_t;

/* Per-step input staging (train_utils.py:257-259 does three .to(gpu) copies): one launch copies the dense features,
 * the ids and the labels into the plan's static buffers and stores the step's learning rate into device memory
 * (lr_scheduler.step(), train_utils.py:386) so that the captured step graph can be replayed unchanged.  The launch can
 * carry the embedding gather of the step as well (the ids are in its hands anyway). */
typedef struct nasrec_stage_desc {
  int32_t kind; /* NASREC_OP_STAGE_INPUTS */
  int32_t B, Fd, Fs;
  float lr;
  int32_t _pad;
  const float* int_src;   float* int_dst;     /* [B, Fd] */
  const int64_t* cat_src; int64_t* cat_dst;   /* [B, Fs] */
  const float* y_src;     float* y_dst;       /* [B] (may be NULL) */
  float* lr_dst;                              /* device scalar (may be NULL) */
  nasrec_embed_desc_t gather;                 /* gather.out != NULL: the same launch also runs the embedding stem on
                                                 cat_src (gather.idx and gather.kind are ignored; B, Fs from above) */
  nasrec_dedup_ids_desc_t dedup_ids;          /* dedup_ids.order != NULL: Fs more workgroups run NASREC_OP_DEDUP_IDS on cat_src (B <= 256;
                                                 idx / B / Fs / kind are ignored): the id-only half of the optimizer's row dedup, off the
                                                 step's tail */
} nasrec_stage_desc_t;

/* ------------------------------------------------------------------------------------------------
 * Heterogeneous launch (batch <= 256).  Inside a choice block the dense nodes and the sparse nodes are independent
 * (supernet.py:1113-1134), blocks depend on a few earlier blocks only, weight-gradient products and split-K second passes have no
 * consumer nearby: nasrec_amd/schedule.py puts the operators of a step that do not depend on each other into LEVELS, and one
 * NASREC_OP_WORKLIST launch runs a level — item k owns the workgroups [first_k, first_k + nblk_k) and executes the very body of its
 * stand-alone kernel on them (bit-identical results; at batch 256 every launch costs ~5 us whatever it does, so a step is as long
 * as its number of dependent launches).
 *   item.kind   NASREC_OP_GEMM (part 0 = the whole launch of an unsplit product, 1 = main pass of a split-K product: slabs only,
 *               2 = its second pass: fixed-order slab sum + epilogue), _DOT_TRI_FWD/_BWD, _FM_FWD/_BWD, _MHA_FWD/_BWD,
 *               _COPY_SEGS, _GATE_BWD, _REDUCE_ROWS (compact form nasrec_wl_reduce_t, <= NASREC_WL_REDUCE_DST destinations);
 *   item.off    byte offset (multiple of 16) of the item's descriptor inside `blob`; a GEMM descriptor may be truncated behind its
 *               last used segment (offsetof(seg) + nseg * sizeof(seg) bytes);
 *   first / nblk / geom are filled in by the launcher (callers leave them 0).
 * All workgroups have 256 threads; GEMM items run the run-time-binding body (csrc/gemm_rt.h) on 64x64 / 32x32 / 64x16 / 16x64 tiles.
 * ---------------------------------------------------------------------------------------------- */
#define NASREC_WL_MAX_ITEMS 12
#define NASREC_WL_BLOB_BYTES 3680
#define NASREC_WL_REDUCE_DST 16
typedef struct nasrec_wl_item {
  int32_t kind, part, off;
  int32_t first, nblk;  /* launcher: workgroup range of the item */
  int32_t geom[3];      /* launcher: grid geometry of the item (GEMM: tiles along n, tiles along m, tile configuration) */
} nasrec_wl_item_t;

typedef struct nasrec_wl_reduce {   /* NASREC_OP_REDUCE_ROWS with few destinations, as a worklist item */
  int32_t kind, R, C, ld;
  const float* in;
  int32_t ndst, _pad;
  float* dst[NASREC_WL_REDUCE_DST];
  int32_t dst_off[NASREC_WL_REDUCE_DST], dst_len[NASREC_WL_REDUCE_DST];
} nasrec_wl_reduce_t;

typedef struct nasrec_worklist_desc {
  int32_t kind; /* NASREC_OP_WORKLIST */
  int32_t n;    /* items, <= NASREC_WL_MAX_ITEMS */
  int32_t total_blocks; /* launcher */
  int32_t _pad;
  nasrec_wl_item_t item[NASREC_WL_MAX_ITEMS];
  char blob[NASREC_WL_BLOB_BYTES] __attribute__((aligned(16)));
} nasrec_worklist_desc_t;

/* NASREC_OP_WORKLIST with the descriptor RESIDENT IN DEVICE MEMORY (ABI 17).  A worklist launch passes its 4 KB descriptor by value: a fresh
 * copy in the runtime's kernel-argument ring per launch, whose lines are cold when the workgroups read their item's descriptor from it —
 * 0.75 - 1.6 us per launch against a buffer that was uploaded once and is read again every step (tools/micro/kernarg_probe.hip), on each of
 * the ~21 dependent launches of a batch-256 step.  nasrec_worklist_prepare computes the items' geometry ONCE (the launcher does it per launch
 * otherwise), copies the completed descriptor to `dev_buf` (sizeof(nasrec_worklist_desc_t) bytes of device memory the caller owns and keeps
 * alive and unchanged; synchronous copy) and fills `out`, which then launches like any descriptor (nasrec_launch / nasrec_program_run /
 * nasrec_graph_create).  The descriptors inside the blob are frozen at that point: for plans whose operand pointers never change. */
typedef struct nasrec_worklist_dev_desc {
  int32_t kind;          /* NASREC_OP_WORKLIST_DEV */
  int32_t total_blocks;
  int32_t big;           /* the launch carries a Transformer backward (LDS size) */
  int32_t _pad;
  uint32_t pf[6], pm[6]; /* the packed item table (first workgroups / kind, part, offset), passed as preloaded scalar arguments */
  const nasrec_worklist_desc_t* dev;
} nasrec_worklist_dev_desc_t;
int nasrec_worklist_prepare(const nasrec_worklist_desc_t* w, void* dev_buf, nasrec_worklist_dev_desc_t* out);

/* ------------------------------------------------------------------------------------------------
 * Persistent step (batch <= 256, round 6): the worklist items of MANY levels in ONE launch, ordered topologically, with the
 * dependencies between them resolved inside the kernel instead of by kernel boundaries.  A level launch lasts as long as its slowest
 * item; here an item starts as soon as the items it depends on (read-after-write, write-after-read, write-after-write on the
 * scheduler's footprints) have finished, whatever else is still running.
 *   * item k owns the workgroups [first_k, first_k + nblk_k) of the launch; workgroups are dispatched in index order, and a workgroup
 *     only ever waits for items with smaller workgroup indices, so every wait ends (a bounded spin raises `err[0]` instead of hanging);
 *   * completion: every finishing workgroup drains its stores and adds to one of up to 8 arrival counters of its item (shard = its index
 *     in the item mod 8); the last arriver of a shard adds to the item's top counter; the last of those stores the step's epoch into
 *     NASREC_PS_REPL replica flags (one 128-byte line each); a waiting workgroup polls ONE replica of each of its <= NASREC_PS_MAX_DEPS
 *     dependencies (one lane per dependency).  No counter is ever reset: the epoch is the item's own counter divided by its size;
 *   * visibility: every buffer an item hands to another item of the same launch must live in UNCACHED device memory
 *     (nasrec_alloc_uncached: plain stores go through to memory), and a workgroup that waited runs an agent-scope acquire (its CU's L1)
 *     before its body — the bodies are the worklist's, unchanged (tools/micro/seam_probe.hip form `uc`).  The caller (schedule.py)
 *     checks the footprints of every dependency edge against the arena's address ranges.
 * The item table, the descriptor blob and the chunk index live in device memory (written by nasrec_persist_prepare, once per plan);
 * descriptors have the worklist's formats and truncation rule.
 * ---------------------------------------------------------------------------------------------- */
#define NASREC_PS_MAX_DEPS 8
#define NASREC_PS_REPL 16
#define NASREC_PS_SHARDS 8
#define NASREC_PS_COUNTER_STRIDE 16   /* uint64 per counter line (128 bytes) */
#define NASREC_PS_FLAG_STRIDE 32      /* uint32 per flag line (128 bytes) */
typedef struct nasrec_persist_item {
  int32_t kind, part, off;   /* as nasrec_wl_item_t; off = byte offset of the descriptor in the blob (multiple of 16) */
  int32_t first, nblk;       /* prepare: workgroup range */
  int32_t geom[3];           /* prepare: as nasrec_wl_item_t */
  int32_t ndeps;
  int32_t deps[NASREC_PS_MAX_DEPS];  /* item indices, all smaller than the item's own */
  int32_t _pad[3];
} nasrec_persist_item_t;     /* 80 bytes */

typedef struct nasrec_persist_desc {
  int32_t kind;          /* NASREC_OP_PERSIST */
  int32_t n;             /* items */
  int32_t total_blocks;  /* prepare */
  int32_t blob_bytes;
  /* device memory, owned by the caller, filled by nasrec_persist_prepare */
  nasrec_persist_item_t* items;     /* [n] */
  char* blob;                       /* [blob_bytes] */
  uint16_t* chunk_item;             /* [chunk_cap]: the item that owns workgroup 16 c */
  unsigned long long* counters;     /* [n][NASREC_PS_SHARDS + 1][NASREC_PS_COUNTER_STRIDE], zero-filled by prepare, never reset */
  uint32_t* flags;                  /* [n][NASREC_PS_REPL][NASREC_PS_FLAG_STRIDE], zero-filled by prepare */
  uint32_t* err;                    /* [4]: err[0] != 0 = a wait ran out of its budget (results invalid); [1] the item, [2] its dependency */
  int32_t chunk_cap, big;           /* capacity of chunk_item (entries); prepare: big = a Transformer backward is among the items */
  unsigned long long* trace;        /* device [total_blocks][4] or NULL: per workgroup the 100 MHz wall clock at entry, with its dependencies
                                       satisfied, behind its body, behind its arrival (tools/persist_timeline.py) */
  /* host memory, read by nasrec_persist_prepare only */
  const nasrec_persist_item_t* host_items;
  const char* host_blob;
} nasrec_persist_desc_t;

/* ------------------------------------------------------------------------------------------------
 * Entry points
 * ---------------------------------------------------------------------------------------------- */
/* geometry of every item, workgroup ranges, chunk index; copies tables and descriptors to the device buffers of `d` and zero-fills its
 * counters and flags (synchronous copies: once per plan, not per step).  Fills d->total_blocks / d->big. */
int nasrec_persist_prepare(nasrec_persist_desc_t* d);

/* Launch one op (any descriptor above) on `stream`. */
int nasrec_launch(void* stream, const void* desc);
/* Launch a program: n descriptors in order on `stream` (one host call per training step). */
int nasrec_program_run(void* stream, const void* const* descs, int n);

/* hipGraph capture of a program: capture once, replay many times (launch-bound B=256 step). */
int nasrec_graph_create(void* stream, const void* const* descs, int n, void** graph_out);
int nasrec_graph_launch(void* graph, void* stream);
int nasrec_graph_destroy(void* graph);

/* Typed convenience wrappers == nasrec_launch with a kind check (one per reference call site). */
int nasrec_gemm(void* stream, const nasrec_gemm_desc_t* d);
int nasrec_embedding_gather(void* stream, const nasrec_embed_desc_t* d);
int nasrec_embedding_dedup(void* stream, const nasrec_emb_dedup_desc_t* d);
int nasrec_dot_tri(void* stream, const nasrec_dot_tri_desc_t* d);
int nasrec_fm(void* stream, const nasrec_fm_desc_t* d);
int nasrec_mha_ffn(void* stream, const nasrec_mha_desc_t* d);
int nasrec_layernorm(void* stream, const nasrec_layernorm_desc_t* d);
int nasrec_final_logit(void* stream, const nasrec_final_desc_t* d);
int nasrec_bce_logits(void* stream, const nasrec_bce_desc_t* d);
int nasrec_adagrad_dense(void* stream, const nasrec_adagrad_dense_desc_t* d);
int nasrec_adagrad_rows(void* stream, const nasrec_adagrad_rows_desc_t* d);
int nasrec_opt_reduce(void* stream, const nasrec_opt_reduce_desc_t* d);
int nasrec_opt_apply(void* stream, const nasrec_opt_apply_desc_t* d);
int nasrec_dedup_ids(void* stream, const nasrec_dedup_ids_desc_t* d);
int nasrec_opt_reduce2(void* stream, const nasrec_opt_reduce2_desc_t* d);
int nasrec_final_fused(void* stream, const nasrec_final_desc_t* d);
int nasrec_worklist(void* stream, const nasrec_worklist_desc_t* d);

/* Uncached device memory (hipExtMallocWithFlags(hipDeviceMallocUncached)): the plan arena of a persistent step (NASREC_OP_PERSIST) —
 * buffers that one workgroup of a launch writes and another workgroup of the SAME launch reads.  The caller owns the memory. */
int nasrec_alloc_uncached(int64_t bytes, void** out);
int nasrec_free_uncached(void* p);

/* HIP-event timing on an arbitrary stream (bench.py measures kernels on the engine's own stream). */
int nasrec_event_create(void** ev);
int nasrec_event_record(void* ev, void* stream);
int nasrec_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int nasrec_event_destroy(void* ev);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline, host side (no GPU work): parse complete TSV lines `label \t Fd integers \t Fs hexadecimal ids` from
 * buf[0, len) into label[r], dense[r, Fd] (raw integers; the caller applies log(max(0,x)+1)) and cat[r, Fs] (already
 * fmod(id, table_rows[f]-1)+1, missing -> 0), at most max_rows of them.  Replaces csv.reader + int() + int(v,16) per field
 * (torchrec/utils.py:175-193, criteo.py:45-58, data_pipes.py:137-175).  Returns the number of rows written; *consumed =
 * bytes of buf they occupied (an incomplete last line is left for the next call); *status = NASREC_TSV_OK, or the reason
 * parsing stopped at the line starting at buf[*consumed]: NEEDS_PYTHON (the line contains a construct on which only
 * Python's own parsers define the result — the caller parses that one line itself) or BAD_COLUMNS.
 * Thread-safe (no shared state); callers parse several shards concurrently.
 * ---------------------------------------------------------------------------------------------- */
enum { NASREC_TSV_OK = 0, NASREC_TSV_NEEDS_PYTHON = 1, NASREC_TSV_BAD_COLUMNS = 2 };
int64_t nasrec_tsv_parse(const char* buf, int64_t len, int32_t Fd, int32_t Fs, const int64_t* table_rows, int64_t max_rows,
                         int64_t* label, int64_t* dense, int64_t* cat, int64_t* consumed, int32_t* status);

const char* nasrec_last_error(void);
int nasrec_abi_version(void);
/* sizeof() of every descriptor, so a binding can verify its struct layout: fills out[0..n) in the
 * order of the NASREC_OP_* enum (index = kind), returns number written. */
int nasrec_desc_sizes(int32_t* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* NASREC_HIP_H */
