// C-ABI entry points (include/nasrec_hip.h): descriptor dispatch, program execution, hipGraph capture,
// HIP-event timing and error reporting.  No torch types, no exceptions, no device synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

int nasrec_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int nasrec_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nasrec_set_error((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

static int dispatch(hipStream_t st, const void* desc) {
  if (desc == nullptr) return nasrec_set_error(-1, "null descriptor");
  const int kind = *reinterpret_cast<const int32_t*>(desc);
  switch (kind) {
    case NASREC_OP_GEMM: return launch_gemm(st, (const nasrec_gemm_desc_t*)desc);
    case NASREC_OP_EMBED_GATHER: return launch_embed_gather(st, (const nasrec_embed_desc_t*)desc);
    case NASREC_OP_DOT_TRI_FWD:
    case NASREC_OP_DOT_TRI_BWD: return launch_dot_tri(st, (const nasrec_dot_tri_desc_t*)desc);
    case NASREC_OP_FM_FWD:
    case NASREC_OP_FM_BWD: return launch_fm(st, (const nasrec_fm_desc_t*)desc);
    case NASREC_OP_MHA_FWD:
    case NASREC_OP_MHA_BWD: return launch_mha(st, (const nasrec_mha_desc_t*)desc);
    case NASREC_OP_REDUCE_ROWS: return launch_reduce_rows(st, (const nasrec_reduce_rows_desc_t*)desc);
    case NASREC_OP_COPY_SEGS: return launch_copy_segs(st, (const nasrec_copy_segs_desc_t*)desc);
    case NASREC_OP_GATE_BWD: return launch_gate_bwd(st, (const nasrec_gate_bwd_desc_t*)desc);
    case NASREC_OP_ROWSUM: return launch_rowsum(st, (const nasrec_rowsum_desc_t*)desc);
    case NASREC_OP_FINAL_FWD:
    case NASREC_OP_FINAL_FUSED:
    case NASREC_OP_FINAL_BWD: return launch_final(st, (const nasrec_final_desc_t*)desc);
    case NASREC_OP_BCE: return launch_bce(st, (const nasrec_bce_desc_t*)desc);
    case NASREC_OP_EMB_DEDUP: return launch_emb_dedup(st, (const nasrec_emb_dedup_desc_t*)desc);
    case NASREC_OP_SUMSQ: return launch_sumsq(st, (const nasrec_sumsq_desc_t*)desc);
    case NASREC_OP_CLIP_COEF: return launch_clip_coef(st, (const nasrec_clip_coef_desc_t*)desc);
    case NASREC_OP_ADAGRAD_DENSE: return launch_adagrad_dense(st, (const nasrec_adagrad_dense_desc_t*)desc);
    case NASREC_OP_ADAGRAD_ROWS: return launch_adagrad_rows(st, (const nasrec_adagrad_rows_desc_t*)desc);
    case NASREC_OP_MEMSET: {
      const nasrec_memset_desc_t* m = (const nasrec_memset_desc_t*)desc;
      if (m->chunks) return launch_memset_chunks(st, m);
      return launch_memset_flat(st, m);
    }
    case NASREC_OP_LAYERNORM_FWD:
    case NASREC_OP_LAYERNORM_BWD: return launch_layernorm(st, (const nasrec_layernorm_desc_t*)desc);
    case NASREC_OP_SCALE: return launch_scale(st, (const nasrec_scale_desc_t*)desc);
    case NASREC_OP_ACT_BWD: return launch_act_bwd(st, (const nasrec_act_bwd_desc_t*)desc);
    case NASREC_OP_STAGE_INPUTS: return launch_stage(st, (const nasrec_stage_desc_t*)desc);
    case NASREC_OP_CONST_I64: return launch_const_i64(st, (const nasrec_const_i64_desc_t*)desc);
    case NASREC_OP_PERSIST: return launch_persist(st, (const nasrec_persist_desc_t*)desc);
    case NASREC_OP_SPLITK_EPILOGUES: return launch_splitk_epilogues(st, (const nasrec_splitk_epilogues_desc_t*)desc);
    case NASREC_OP_OPT_REDUCE: return launch_opt_reduce(st, (const nasrec_opt_reduce_desc_t*)desc);
    case NASREC_OP_OPT_APPLY: return launch_opt_apply(st, (const nasrec_opt_apply_desc_t*)desc);
    case NASREC_OP_WORKLIST: return launch_worklist(st, (const nasrec_worklist_desc_t*)desc);
    case NASREC_OP_WORKLIST_DEV: return launch_worklist_dev(st, (const nasrec_worklist_dev_desc_t*)desc);
    case NASREC_OP_DEDUP_IDS: return launch_dedup_ids(st, (const nasrec_dedup_ids_desc_t*)desc);
    case NASREC_OP_OPT_REDUCE2: return launch_opt_reduce2(st, (const nasrec_opt_reduce2_desc_t*)desc);
    default: return nasrec_set_error(-1, "unknown op kind %d", kind);
  }
}

extern "C" {

int nasrec_launch(void* stream, const void* desc) { return dispatch((hipStream_t)stream, desc); }

int nasrec_program_run(void* stream, const void* const* descs, int n) {
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    int rc = dispatch(st, descs[i]);
    if (rc != 0) {
      char tmp[400];
      snprintf(tmp, sizeof(tmp), "%s", g_err);
      return nasrec_set_error(rc, "program op %d (kind %d): %s", i, descs[i] ? *(const int32_t*)descs[i] : -1, tmp);
    }
  }
  return 0;
}

struct nasrec_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

int nasrec_graph_create(void* stream, const void* const* descs, int n, void** graph_out) {
  hipStream_t st = (hipStream_t)stream;
  if (graph_out == nullptr) return nasrec_set_error(-1, "graph_out is null");
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return nasrec_set_error((int)e, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  int rc = nasrec_program_run(stream, descs, n);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st, &g);
  if (rc != 0) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) return nasrec_set_error((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    return nasrec_set_error((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  nasrec_graph* h = new nasrec_graph{g, ex};
  *graph_out = h;
  return 0;
}

int nasrec_graph_launch(void* graph, void* stream) {
  nasrec_graph* h = (nasrec_graph*)graph;
  if (!h) return nasrec_set_error(-1, "null graph");
  hipError_t e = hipGraphLaunch(h->exec, (hipStream_t)stream);
  if (e != hipSuccess) return nasrec_set_error((int)e, "hipGraphLaunch: %s", hipGetErrorString(e));
  return 0;
}

int nasrec_graph_destroy(void* graph) {
  nasrec_graph* h = (nasrec_graph*)graph;
  if (!h) return 0;
  (void)hipGraphExecDestroy(h->exec);
  (void)hipGraphDestroy(h->graph);
  delete h;
  return 0;
}

#define TYPED(name, type, cond)                                                               \
  int name(void* stream, const type* d) {                                                     \
    if (d == nullptr) return nasrec_set_error(-1, #name ": null descriptor");                 \
    const int kind = d->kind;                                                                 \
    if (!(cond)) return nasrec_set_error(-1, #name ": descriptor kind %d does not match", kind); \
    return dispatch((hipStream_t)stream, d);                                                  \
  }

TYPED(nasrec_gemm, nasrec_gemm_desc_t, kind == NASREC_OP_GEMM)
TYPED(nasrec_embedding_gather, nasrec_embed_desc_t, kind == NASREC_OP_EMBED_GATHER)
TYPED(nasrec_embedding_dedup, nasrec_emb_dedup_desc_t, kind == NASREC_OP_EMB_DEDUP)
TYPED(nasrec_dot_tri, nasrec_dot_tri_desc_t, kind == NASREC_OP_DOT_TRI_FWD || kind == NASREC_OP_DOT_TRI_BWD)
TYPED(nasrec_fm, nasrec_fm_desc_t, kind == NASREC_OP_FM_FWD || kind == NASREC_OP_FM_BWD)
TYPED(nasrec_mha_ffn, nasrec_mha_desc_t, kind == NASREC_OP_MHA_FWD || kind == NASREC_OP_MHA_BWD)
TYPED(nasrec_layernorm, nasrec_layernorm_desc_t, kind == NASREC_OP_LAYERNORM_FWD || kind == NASREC_OP_LAYERNORM_BWD)
TYPED(nasrec_final_logit, nasrec_final_desc_t, kind == NASREC_OP_FINAL_FWD || kind == NASREC_OP_FINAL_BWD)
TYPED(nasrec_bce_logits, nasrec_bce_desc_t, kind == NASREC_OP_BCE)
TYPED(nasrec_adagrad_dense, nasrec_adagrad_dense_desc_t, kind == NASREC_OP_ADAGRAD_DENSE)
TYPED(nasrec_adagrad_rows, nasrec_adagrad_rows_desc_t, kind == NASREC_OP_ADAGRAD_ROWS)
TYPED(nasrec_opt_reduce, nasrec_opt_reduce_desc_t, kind == NASREC_OP_OPT_REDUCE)
TYPED(nasrec_opt_apply, nasrec_opt_apply_desc_t, kind == NASREC_OP_OPT_APPLY)
TYPED(nasrec_worklist, nasrec_worklist_desc_t, kind == NASREC_OP_WORKLIST)
TYPED(nasrec_dedup_ids, nasrec_dedup_ids_desc_t, kind == NASREC_OP_DEDUP_IDS)
TYPED(nasrec_opt_reduce2, nasrec_opt_reduce2_desc_t, kind == NASREC_OP_OPT_REDUCE2)
TYPED(nasrec_final_fused, nasrec_final_desc_t, kind == NASREC_OP_FINAL_FUSED)

// Device memory that the XCDs' L2 caches do not hold (MTYPE uncached): plain stores go through to memory, and a plain load behind an
// agent-scope acquire (buffer_inv sc1: the CU's L1) reads what another workgroup of the SAME launch stored — what the persistent step
// kernel (csrc/persist.hip) needs of every buffer that one of its items writes and another reads (tools/micro/seam_probe.hip `uc`).
int nasrec_alloc_uncached(int64_t bytes, void** out) {
  void* p = nullptr;
  hipError_t rc = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "hipExtMallocWithFlags(%lld, uncached): %s", (long long)bytes, hipGetErrorString(rc));
  *out = p;
  return 0;
}

int nasrec_free_uncached(void* p) {
  hipError_t rc = hipFree(p);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "hipFree: %s", hipGetErrorString(rc));
  return 0;
}

int nasrec_event_create(void** ev) {
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "hipEventCreate: %s", hipGetErrorString(rc));
  *ev = (void*)e;
  return 0;
}

int nasrec_event_record(void* ev, void* stream) {
  hipError_t rc = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "hipEventRecord: %s", hipGetErrorString(rc));
  return 0;
}

int nasrec_event_elapsed_ms(void* start, void* stop, float* ms) {
  hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "hipEventSynchronize: %s", hipGetErrorString(rc));
  rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
  if (rc != hipSuccess) return nasrec_set_error((int)rc, "hipEventElapsedTime: %s", hipGetErrorString(rc));
  return 0;
}

int nasrec_event_destroy(void* ev) {
  (void)hipEventDestroy((hipEvent_t)ev);
  return 0;
}

const char* nasrec_last_error(void) { return g_err; }

int nasrec_abi_version(void) { return 17; }

int nasrec_desc_sizes(int32_t* out, int n) {
  static const int32_t sizes[] = {
      0,
      (int32_t)sizeof(nasrec_gemm_desc_t),          // 1
      (int32_t)sizeof(nasrec_embed_desc_t),         // 2
      (int32_t)sizeof(nasrec_dot_tri_desc_t),       // 3
      (int32_t)sizeof(nasrec_dot_tri_desc_t),       // 4
      (int32_t)sizeof(nasrec_fm_desc_t),            // 5
      (int32_t)sizeof(nasrec_fm_desc_t),            // 6
      (int32_t)sizeof(nasrec_mha_desc_t),           // 7
      (int32_t)sizeof(nasrec_mha_desc_t),           // 8
      (int32_t)sizeof(nasrec_reduce_rows_desc_t),   // 9
      (int32_t)sizeof(nasrec_copy_segs_desc_t),     // 10
      (int32_t)sizeof(nasrec_gate_bwd_desc_t),      // 11
      (int32_t)sizeof(nasrec_rowsum_desc_t),        // 12
      (int32_t)sizeof(nasrec_final_desc_t),         // 13
      (int32_t)sizeof(nasrec_bce_desc_t),           // 14
      (int32_t)sizeof(nasrec_final_desc_t),         // 15
      (int32_t)sizeof(nasrec_emb_dedup_desc_t),     // 16
      (int32_t)sizeof(nasrec_sumsq_desc_t),         // 17
      (int32_t)sizeof(nasrec_clip_coef_desc_t),     // 18
      (int32_t)sizeof(nasrec_adagrad_dense_desc_t), // 19
      (int32_t)sizeof(nasrec_adagrad_rows_desc_t),  // 20
      (int32_t)sizeof(nasrec_memset_desc_t),        // 21
      (int32_t)sizeof(nasrec_layernorm_desc_t),     // 22
      (int32_t)sizeof(nasrec_layernorm_desc_t),     // 23
      0,                                            // 24 (ADD_SEGS: served by COPY_SEGS)
      (int32_t)sizeof(nasrec_scale_desc_t),         // 25
      (int32_t)sizeof(nasrec_act_bwd_desc_t),       // 26
      (int32_t)sizeof(nasrec_stage_desc_t),         // 27
      (int32_t)sizeof(nasrec_opt_reduce_desc_t),    // 28
      (int32_t)sizeof(nasrec_opt_apply_desc_t),     // 29
      (int32_t)sizeof(nasrec_worklist_desc_t),      // 30
      (int32_t)sizeof(nasrec_const_i64_desc_t),     // 31
      (int32_t)sizeof(nasrec_splitk_epilogues_desc_t), // 32
      (int32_t)sizeof(nasrec_dedup_ids_desc_t),     // 33
      (int32_t)sizeof(nasrec_opt_reduce2_desc_t),   // 34
      (int32_t)sizeof(nasrec_final_desc_t),         // 35
      (int32_t)sizeof(nasrec_persist_desc_t),       // 36
      (int32_t)sizeof(nasrec_persist_item_t),       // 37 (not an op: the item record of NASREC_OP_PERSIST, for the binding's layout check)
      (int32_t)sizeof(nasrec_worklist_dev_desc_t),  // 38
  };
  const int total = (int)(sizeof(sizes) / sizeof(sizes[0]));
  int w = 0;
  for (; w < n && w < total; ++w) out[w] = sizes[w];
  return w;
}

}  // extern "C"
