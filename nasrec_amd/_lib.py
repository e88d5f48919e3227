"""ctypes binding of include/nasrec_hip.h (the C-ABI of the HIP engine).

This is the reference-side binding a maintainer would add: plain structs, raw device pointers,
a hipStream_t passed as void*.  There is NO CPU fallback: if the shared library is missing or a symbol / struct
layout does not match, loading fails loudly.
"""
import ctypes as C
import os

# torch must be imported BEFORE the engine is dlopen'ed: PyTorch-ROCm bundles its own libamdhip64, and the engine has
# to bind to that same HIP runtime instance (device pointers and streams are shared with torch).  Loading the engine
# first would pull in the system runtime as a second, separate HIP instance ("no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# NASREC_HIP_LIB points the loader at another build of the same library (kernel A/B experiments).
LIB_PATH = os.environ.get("NASREC_HIP_LIB") or os.path.join(_HERE, "lib", "libnasrec_hip.so")

MAX_SEGS = 8
MAX_TABLES = 32
EMB_DIM = 16
MHA_PARAMS = 1696
MHA_SAVED = 36
REDUCE_MAX_DST = 48

AM_KC, AM_RC, AM_TOKR, AM_TOKK = 0, 1, 2, 3
CM_PLAIN, CM_TOKJ = 0, 1
ACT_NONE, ACT_RELU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3
TSV_OK, TSV_NEEDS_PYTHON, TSV_BAD_COLUMNS = 0, 1, 2
ACT_BY_NAME = {"identity": ACT_NONE, "relu": ACT_RELU, "silu": ACT_SILU}

(OP_GEMM, OP_EMBED_GATHER, OP_DOT_TRI_FWD, OP_DOT_TRI_BWD, OP_FM_FWD, OP_FM_BWD, OP_MHA_FWD, OP_MHA_BWD, OP_REDUCE_ROWS,
 OP_COPY_SEGS, OP_GATE_BWD, OP_ROWSUM, OP_FINAL_FWD, OP_BCE, OP_FINAL_BWD, OP_EMB_DEDUP, OP_SUMSQ, OP_CLIP_COEF,
 OP_ADAGRAD_DENSE, OP_ADAGRAD_ROWS, OP_MEMSET, OP_LAYERNORM_FWD, OP_LAYERNORM_BWD, OP_ADD_SEGS, OP_SCALE,
 OP_ACT_BWD, OP_STAGE_INPUTS, OP_OPT_REDUCE, OP_OPT_APPLY, OP_WORKLIST, OP_CONST_I64, OP_SPLITK_EPILOGUES, OP_DEDUP_IDS,
 OP_OPT_REDUCE2, OP_FINAL_FUSED, OP_PERSIST) = range(1, 37)
OP_WORKLIST_DEV = 38  # (37: a layout-check slot of nasrec_desc_sizes)

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmSeg(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("Aaux", vp), ("Baux", vp), ("rowsum", vp), ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32), ("Mvalid", i32), ("accumulate", i32), ("ones_col", i32), ("_pad", i32)]


class GemmDesc(C.Structure):
    _fields_ = [("kind", i32), ("amode", i32), ("bmode", i32), ("cmode", i32), ("nseg", i32), ("zmode", i32), ("act", i32),
                ("bias_on_rows", i32), ("mask_on_rows", i32), ("dims_in_use", i32), ("beta", i32), ("splitk", i32),
                ("bias", vp), ("save_z", vp), ("save_act", vp), ("mul_ptr", vp * MAX_SEGS), ("mul_off", i32 * MAX_SEGS),
                ("mul_width", i32 * MAX_SEGS), ("mul_ld", i32 * MAX_SEGS), ("mul_nseg", i32), ("defer_second_pass", i32),
                ("workspace", vp), ("counters", vp), ("rowsum_out", vp), ("pre_add", vp), ("seg", GemmSeg * MAX_SEGS)]


EPILOGUES_MAX = 3  # NASREC_EPILOGUES_MAX


class SplitkEpiloguesDesc(C.Structure):
    _fields_ = [("kind", i32), ("n", i32), ("g", GemmDesc * EPILOGUES_MAX)]


class EmbedDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("Fs", i32), ("_pad", i32), ("idx", vp), ("table", vp * MAX_TABLES),
                ("rows", i64 * MAX_TABLES), ("out", vp), ("oob", vp)]


class EmbDedupDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("Fs", i32), ("_pad", i32), ("idx", vp), ("dout", vp), ("leader", vp), ("gsum", vp),
                ("sumsq_partial", vp), ("overflow", vp), ("rank_B", i32), ("_pad2", i32), ("rank_stride", i64)]


class DotTriDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("k1", i32), ("ld_out", i32), ("T", vp), ("out", vp), ("dout", vp), ("dT", vp)]


class FmDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("N", i32), ("ldx", i32), ("ld_ix", i32), ("accumulate", i32), ("x", vp), ("ix", vp),
                ("dix", vp), ("dx", vp), ("add", vp)]


class MhaDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("N", i32), ("ldx", i32), ("ldo", i32), ("dims_in_use", i32), ("x", vp), ("out", vp),
                ("dout", vp), ("dx", vp), ("dparams_partial", vp), ("params", vp * 12), ("saved", vp), ("partial_ld", i32), ("bwd_form", i32)]


class ReduceRowsDesc(C.Structure):
    _fields_ = [("kind", i32), ("R", i32), ("C", i32), ("ld", i32), ("in_", vp), ("ndst", i32), ("_pad", i32), ("dst", vp * REDUCE_MAX_DST),
                ("dst_off", i32 * REDUCE_MAX_DST), ("dst_len", i32 * REDUCE_MAX_DST)]


class CopySegsDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("nseg", i32), ("ld_dst", i32), ("accumulate", i32), ("reverse", i32), ("dst", vp),
                ("seg", vp * MAX_SEGS), ("width", i32 * MAX_SEGS), ("ld", i32 * MAX_SEGS), ("off", i32 * MAX_SEGS),
                ("seg_accumulate", i32 * MAX_SEGS)]


class GateBwdDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("D", i32), ("ld_dout", i32), ("ld_g", i32), ("ld_dz", i32), ("dout", vp), ("g", vp),
                ("dz", vp), ("nseg", i32), ("r_ptr", vp * MAX_SEGS), ("dr_ptr", vp * MAX_SEGS), ("r_off", i32 * MAX_SEGS),
                ("r_width", i32 * MAX_SEGS), ("r_ld", i32 * MAX_SEGS), ("dr_accumulate", i32 * MAX_SEGS)]


class RowsumDesc(C.Structure):
    _fields_ = [("kind", i32), ("mode", i32), ("R", i32), ("K", i32), ("ld", i32), ("rvalid", i32), ("p", vp), ("aux", vp),
                ("out", vp)]


class FinalDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("nseg", i32), ("grad_scale", f32), ("w", vp), ("bias", vp), ("logits", vp), ("dlogits", vp),
                ("dw", vp), ("dbias", vp), ("seg", vp * MAX_SEGS), ("dseg", vp * MAX_SEGS), ("width", i32 * MAX_SEGS),
                ("ld", i32 * MAX_SEGS), ("off", i32 * MAX_SEGS), ("dseg_accumulate", i32 * MAX_SEGS), ("y", vp), ("loss", vp),
                ("dlogits_out", vp), ("nsplit", i32), ("dseg_done", i32), ("tok_stride", i32 * MAX_SEGS)]


class BceDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("grad_scale", f32), ("_pad", i32), ("logits", vp), ("y", vp), ("loss", vp),
                ("dlogits", vp)]


class SumsqDesc(C.Structure):
    _fields_ = [("kind", i32), ("nblocks", i32), ("n", i64), ("x", vp), ("partial", vp), ("chunks", vp), ("nchunks", i64)]


class ClipCoefDesc(C.Structure):
    _fields_ = [("kind", i32), ("n_a", i32), ("n_b", i32), ("max_norm", f32), ("partial_a", vp), ("partial_b", vp), ("out", vp)]


class AdagradDenseDesc(C.Structure):
    _fields_ = [("kind", i32), ("eps", f32), ("n", i64), ("p", vp), ("g", vp), ("state", vp), ("lr", vp), ("coef", vp),
                ("chunks", vp), ("nchunks", i64)]


class AdagradRowsDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("Fs", i32), ("eps", f32), ("idx", vp), ("leader", vp), ("gsum", vp),
                ("table", vp * MAX_TABLES), ("state", vp * MAX_TABLES), ("lr", vp), ("coef", vp), ("rows", i64 * MAX_TABLES),
                ("rank_B", i32), ("_pad", i32), ("rank_stride", i64)]


class MemsetDesc(C.Structure):
    _fields_ = [("kind", i32), ("_pad", i32), ("bytes", i64), ("ptr", vp), ("chunks", vp), ("nchunks", i64)]


SPLITK_BALANCED = -1                  # NASREC_SPLITK_BALANCED
SK_WORKSPACE_FLOATS = 512 * 3 * 128 * 128  # NASREC_SK_WORKSPACE_FLOATS
CHUNK_ELEMS = 65536     # NASREC_CHUNK_ELEMS
CONST_I64_MAX = 448     # NASREC_CONST_I64_MAX


class ConstI64Desc(C.Structure):
    _fields_ = [("kind", i32), ("n", i32), ("dst", vp), ("vals", i64 * CONST_I64_MAX)]


class LayerNormDesc(C.Structure):
    _fields_ = [("kind", i32), ("mode", i32), ("R", i32), ("D", i32), ("ldx", i32), ("ldy", i32), ("act", i32),
                ("dims_in_use", i32), ("accumulate", i32), ("eps", f32), ("x", vp), ("w", vp), ("b", vp), ("y", vp), ("stats", vp),
                ("dy", vp), ("dx", vp), ("dwb_partial", vp), ("nblk", i32), ("_pad", i32)]


class ScaleDesc(C.Structure):
    _fields_ = [("kind", i32), ("_pad", i32), ("n", i64), ("s", f32), ("_pad2", i32), ("x", vp), ("y", vp)]


class ActBwdDesc(C.Structure):
    _fields_ = [("kind", i32), ("mode", i32), ("R", i32), ("D", i32), ("ld_dy", i32), ("ld_z", i32), ("ld_dz", i32), ("act", i32),
                ("dims_in_use", i32), ("dy", vp), ("z", vp), ("dz", vp)]


DEDUP_IDS_MAX_B = 2048  # NASREC_DEDUP_IDS_MAX_B


class DedupIdsDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("Fs", i32), ("cap", i32), ("idx", vp), ("leader", vp), ("order", vp), ("lists", vp), ("counts", vp),
                ("heads", vp)]


class OptReduce2Desc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("Fs", i32), ("cap", i32), ("rank_B", i32), ("row_blocks", i32), ("rank_stride", i64), ("rows", vp),
                ("leader", vp), ("order", vp), ("lists", vp), ("counts", vp), ("heads", vp), ("sumsq_partial", vp), ("sumsq", SumsqDesc)]


class StageDesc(C.Structure):
    _fields_ = [("kind", i32), ("B", i32), ("Fd", i32), ("Fs", i32), ("lr", f32), ("_pad", i32), ("int_src", vp), ("int_dst", vp),
                ("cat_src", vp), ("cat_dst", vp), ("y_src", vp), ("y_dst", vp), ("lr_dst", vp), ("gather", EmbedDesc), ("dedup_ids", DedupIdsDesc)]


class OptReduceDesc(C.Structure):
    _fields_ = [("kind", i32), ("_pad", i32), ("dedup", EmbDedupDesc), ("sumsq", SumsqDesc)]


class OptApplyDesc(C.Structure):
    _fields_ = [("kind", i32), ("dense_blocks", i32), ("clip", ClipCoefDesc), ("dense", AdagradDenseDesc), ("rows", AdagradRowsDesc)]


WL_MAX_ITEMS, WL_BLOB_BYTES, WL_REDUCE_DST = 12, 3680, 16  # NASREC_WL_*
WL_WHOLE, WL_MAIN, WL_EPI = 0, 1, 2                      # nasrec_wl_item_t.part of a GEMM item


class WlItem(C.Structure):
    _fields_ = [("kind", i32), ("part", i32), ("off", i32), ("first", i32), ("nblk", i32), ("geom", i32 * 3)]


class WlReduce(C.Structure):
    _fields_ = [("kind", i32), ("R", i32), ("C", i32), ("ld", i32), ("in_", vp), ("ndst", i32), ("_pad", i32), ("dst", vp * WL_REDUCE_DST),
                ("dst_off", i32 * WL_REDUCE_DST), ("dst_len", i32 * WL_REDUCE_DST)]


class WorklistDesc(C.Structure):
    _fields_ = [("kind", i32), ("n", i32), ("total_blocks", i32), ("_pad", i32), ("item", WlItem * WL_MAX_ITEMS),
                ("blob", C.c_char * WL_BLOB_BYTES)]


class WorklistDevDesc(C.Structure):
    """NASREC_OP_WORKLIST with its descriptor resident in device memory (nasrec_worklist_prepare, ABI 17)"""
    _fields_ = [("kind", i32), ("total_blocks", i32), ("big", i32), ("_pad", i32), ("pf", C.c_uint32 * 6), ("pm", C.c_uint32 * 6), ("dev", vp)]


PS_MAX_DEPS, PS_REPL, PS_SHARDS, PS_COUNTER_STRIDE, PS_FLAG_STRIDE = 8, 16, 8, 16, 32  # NASREC_PS_*


class PersistItem(C.Structure):
    _fields_ = [("kind", i32), ("part", i32), ("off", i32), ("first", i32), ("nblk", i32), ("geom", i32 * 3), ("ndeps", i32),
                ("deps", i32 * PS_MAX_DEPS), ("_pad", i32 * 3)]


class PersistDesc(C.Structure):
    _fields_ = [("kind", i32), ("n", i32), ("total_blocks", i32), ("blob_bytes", i32), ("items", vp), ("blob", vp), ("chunk_item", vp),
                ("counters", vp), ("flags", vp), ("err", vp), ("chunk_cap", i32), ("big", i32), ("trace", vp), ("host_items", vp), ("host_blob", vp)]


DESC_BY_KIND = {
    OP_GEMM: GemmDesc, OP_EMBED_GATHER: EmbedDesc, OP_DOT_TRI_FWD: DotTriDesc, OP_DOT_TRI_BWD: DotTriDesc, OP_FM_FWD: FmDesc,
    OP_FM_BWD: FmDesc, OP_MHA_FWD: MhaDesc, OP_MHA_BWD: MhaDesc, OP_REDUCE_ROWS: ReduceRowsDesc, OP_COPY_SEGS: CopySegsDesc,
    OP_GATE_BWD: GateBwdDesc, OP_ROWSUM: RowsumDesc, OP_FINAL_FWD: FinalDesc, OP_BCE: BceDesc, OP_FINAL_BWD: FinalDesc,
    OP_EMB_DEDUP: EmbDedupDesc, OP_SUMSQ: SumsqDesc, OP_CLIP_COEF: ClipCoefDesc, OP_ADAGRAD_DENSE: AdagradDenseDesc,
    OP_ADAGRAD_ROWS: AdagradRowsDesc, OP_MEMSET: MemsetDesc, OP_LAYERNORM_FWD: LayerNormDesc, OP_LAYERNORM_BWD: LayerNormDesc,
    OP_SCALE: ScaleDesc, OP_ACT_BWD: ActBwdDesc, OP_STAGE_INPUTS: StageDesc, OP_OPT_REDUCE: OptReduceDesc, OP_OPT_APPLY: OptApplyDesc,
    OP_WORKLIST: WorklistDesc, OP_CONST_I64: ConstI64Desc, OP_SPLITK_EPILOGUES: SplitkEpiloguesDesc, OP_DEDUP_IDS: DedupIdsDesc,
    OP_OPT_REDUCE2: OptReduce2Desc, OP_FINAL_FUSED: FinalDesc, OP_PERSIST: PersistDesc, OP_WORKLIST_DEV: WorklistDevDesc,
}

# every symbol include/nasrec_hip.h declares
SYMBOLS = [
    "nasrec_launch", "nasrec_program_run", "nasrec_graph_create", "nasrec_graph_launch", "nasrec_graph_destroy", "nasrec_gemm",
    "nasrec_embedding_gather", "nasrec_embedding_dedup", "nasrec_dot_tri", "nasrec_fm", "nasrec_mha_ffn", "nasrec_layernorm",
    "nasrec_final_logit", "nasrec_bce_logits", "nasrec_adagrad_dense", "nasrec_adagrad_rows", "nasrec_opt_reduce",
    "nasrec_opt_apply", "nasrec_worklist", "nasrec_dedup_ids", "nasrec_opt_reduce2", "nasrec_final_fused", "nasrec_event_create",
    "nasrec_event_record", "nasrec_event_elapsed_ms", "nasrec_event_destroy", "nasrec_last_error", "nasrec_abi_version",
    "nasrec_desc_sizes", "nasrec_tsv_parse", "nasrec_alloc_uncached", "nasrec_free_uncached", "nasrec_persist_prepare", "nasrec_worklist_prepare",
]

_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """dlopen the engine, verify every declared symbol exists and every struct layout matches. Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("HIP engine library not found at %s — run `python __graft_entry__.py` (build()) first; "
                          "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise EngineError("libnasrec_hip.so does not export %s" % s)
    lib.nasrec_last_error.restype = C.c_char_p
    lib.nasrec_launch.argtypes = [vp, vp]
    lib.nasrec_program_run.argtypes = [vp, C.POINTER(vp), C.c_int]
    lib.nasrec_graph_create.argtypes = [vp, C.POINTER(vp), C.c_int, C.POINTER(vp)]
    lib.nasrec_graph_launch.argtypes = [vp, vp]
    lib.nasrec_graph_destroy.argtypes = [vp]
    lib.nasrec_event_create.argtypes = [C.POINTER(vp)]
    lib.nasrec_event_record.argtypes = [vp, vp]
    lib.nasrec_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(f32)]
    lib.nasrec_event_destroy.argtypes = [vp]
    lib.nasrec_desc_sizes.argtypes = [C.POINTER(i32), C.c_int]
    lib.nasrec_alloc_uncached.argtypes = [i64, C.POINTER(vp)]
    lib.nasrec_free_uncached.argtypes = [vp]
    lib.nasrec_persist_prepare.argtypes = [vp]
    lib.nasrec_worklist_prepare.argtypes = [vp, vp, vp]
    lib.nasrec_tsv_parse.argtypes = [vp, i64, i32, i32, vp, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i32)]
    lib.nasrec_tsv_parse.restype = i64
    for name in ("nasrec_gemm", "nasrec_embedding_gather", "nasrec_embedding_dedup", "nasrec_dot_tri", "nasrec_fm",
                 "nasrec_mha_ffn", "nasrec_layernorm", "nasrec_final_logit", "nasrec_bce_logits", "nasrec_adagrad_dense",
                 "nasrec_adagrad_rows", "nasrec_opt_reduce", "nasrec_opt_apply", "nasrec_worklist", "nasrec_dedup_ids", "nasrec_opt_reduce2", "nasrec_final_fused"):
        getattr(lib, name).argtypes = [vp, vp]
    if lib.nasrec_abi_version() != 17:
        raise EngineError("ABI version mismatch: library %d, binding 17" % lib.nasrec_abi_version())
    sizes = (i32 * 40)()
    n = lib.nasrec_desc_sizes(sizes, 40)
    for kind, cls in DESC_BY_KIND.items():
        if kind >= n or sizes[kind] != C.sizeof(cls):
            raise EngineError("struct layout mismatch for op kind %d: library %d bytes, binding %d bytes"
                              % (kind, sizes[kind] if kind < n else -1, C.sizeof(cls)))
    if n <= 37 or sizes[37] != C.sizeof(PersistItem):
        raise EngineError("struct layout mismatch for nasrec_persist_item_t: library %d bytes, binding %d bytes" % (sizes[37] if n > 37 else -1, C.sizeof(PersistItem)))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise EngineError("nasrec engine error %d: %s" % (rc, _lib.nasrec_last_error().decode(errors="replace")))
