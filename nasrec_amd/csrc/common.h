// Shared device/host helpers for the gfx950 NASRec engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nasrec_hip.h"

#define NASREC_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// address of element (r,k) of an operand under the four addressing modes of nasrec_hip.h
template <int MODE>
__device__ __forceinline__ long operand_offset(int r, int k, int ld) {
  if (MODE == NASREC_AM_KC) return (long)r * ld + k;
  if (MODE == NASREC_AM_RC) return (long)k * ld + r;
  if (MODE == NASREC_AM_TOKR) return (long)(r >> 4) * ld + (r & 15) + (long)k * 16;
  return (long)(k >> 4) * ld + (k & 15) + (long)r * 16;  // TOKK
}

template <int CMODE>
__device__ __forceinline__ long c_offset(int i, int j, int ldc) {
  if (CMODE == NASREC_CM_PLAIN) return (long)i * ldc + j;
  return (long)(j >> 4) * ldc + (j & 15) + (long)i * 16;
}

__device__ __forceinline__ float act_apply(float z, int act) {
  switch (act) {
    case NASREC_ACT_RELU: return z > 0.f ? z : 0.f;
    case NASREC_ACT_SILU: return z / (1.f + __expf(-z));
    case NASREC_ACT_SIGMOID: return 1.f / (1.f + __expf(-z));
    default: return z;
  }
}

// Pull the first BYTES of the kernel-argument segment through the scalar cache in ONE memory round trip, from the workgroup's first
// wave.  The descriptors of this engine are by-value kernel arguments of 0.4 - 4 KB, cold in every cache when a kernel starts; the
// compiler loads their fields lazily, next to the first use, so a prologue that branches on one field before it reads the next pays
// a round trip to memory (0.6 - 2 us on a cold L2) per level.  Called first in a kernel, this touches every 64-byte line at once;
// the lazily placed loads of all waves then find the lines in (or on their way into) the scalar cache.  One wave only: scalar loads
// are slow to issue (64 of them from every wave of every workgroup cost more than they saved: cfg 2 step 0.422 -> 0.447 ms).
// (One asm block with its own wait: the destination register is dead afterwards, never live across other code.)
template <int BYTES>
__device__ __forceinline__ void warm_kernarg() {
  static_assert(BYTES % 64 == 0 && BYTES >= 64 && BYTES <= 4096, "whole 64-byte lines");
  if (threadIdx.x >= 64) return;
  const unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  int sink;
  if (BYTES <= 1024) {
    asm volatile(
        ".set nasrec_wk_off, 0\n"
        ".rept %c2\n"
        "s_load_dword %0, %1, nasrec_wk_off\n"
        ".set nasrec_wk_off, nasrec_wk_off + 64\n"
        ".endr\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(sink)
        : "s"(p), "n"(BYTES / 64)
        : "memory");
  } else {  // more than 16 loads: the counter has 4 bits, drain in groups of 16
    asm volatile(
        ".set nasrec_wk_off, 0\n"
        ".rept %c2\n"
        ".rept 16\n"
        "s_load_dword %0, %1, nasrec_wk_off\n"
        ".set nasrec_wk_off, nasrec_wk_off + 64\n"
        ".endr\n"
        "s_waitcnt lgkmcnt(0)\n"
        ".endr"
        : "=&s"(sink)
        : "s"(p), "n"(BYTES / 1024)
        : "memory");
  }
}

// Sum over the 64 lanes, returned to every lane.  Data-parallel-primitive moves inside the vector ALU (two quad permutes, two row
// rotations, two row broadcasts; the total lands in lane 63) instead of six ds_bpermute round trips through the LDS crossbar: the
// Transformer backward does 40 of these per wave and sample (its bias-like gradients), each a dependent chain.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xb1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4e>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8   (every lane of a row of 16 now holds the row's sum)
  v += dpp_mov<0x142>(v);  // row_bcast:15
  v += dpp_mov<0x143>(v);  // row_bcast:31 (lane 63: rows 0 + 1 + 2 + 3)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// host-side launch-error helper (defined in api.hip)
int nasrec_set_error(int code, const char* fmt, ...);
int nasrec_check_launch(const char* what);

// A dynamic-LDS limit above the default 64 KB is a property of (kernel, DEVICE): `mask` remembers which devices of this process have
// been told for one kernel (a process may drive several GPUs: main_train.py --gpu N next to another engine on cuda:0).  True when
// the caller has to set the attribute on the current device.  (Setting it twice is harmless, so no lock.)
static inline bool nasrec_lds_attr_needed(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

// per-kind launchers (each .hip file defines its own)
int launch_gemm(hipStream_t s, const nasrec_gemm_desc_t* d);
int launch_embed_gather(hipStream_t s, const nasrec_embed_desc_t* d);
int launch_emb_dedup(hipStream_t s, const nasrec_emb_dedup_desc_t* d);
int launch_dot_tri(hipStream_t s, const nasrec_dot_tri_desc_t* d);
int launch_fm(hipStream_t s, const nasrec_fm_desc_t* d);
int launch_mha(hipStream_t s, const nasrec_mha_desc_t* d);
int launch_reduce_rows(hipStream_t s, const nasrec_reduce_rows_desc_t* d);
int launch_copy_segs(hipStream_t s, const nasrec_copy_segs_desc_t* d);
int launch_gate_bwd(hipStream_t s, const nasrec_gate_bwd_desc_t* d);
int launch_rowsum(hipStream_t s, const nasrec_rowsum_desc_t* d);
int launch_final(hipStream_t s, const nasrec_final_desc_t* d);
int final_fused_check(const nasrec_final_desc_t* d);
int launch_bce(hipStream_t s, const nasrec_bce_desc_t* d);
int launch_sumsq(hipStream_t s, const nasrec_sumsq_desc_t* d);
int launch_clip_coef(hipStream_t s, const nasrec_clip_coef_desc_t* d);
int launch_adagrad_dense(hipStream_t s, const nasrec_adagrad_dense_desc_t* d);
int launch_adagrad_rows(hipStream_t s, const nasrec_adagrad_rows_desc_t* d);
int launch_layernorm(hipStream_t s, const nasrec_layernorm_desc_t* d);
int launch_scale(hipStream_t s, const nasrec_scale_desc_t* d);
int launch_act_bwd(hipStream_t s, const nasrec_act_bwd_desc_t* d);
int launch_stage(hipStream_t s, const nasrec_stage_desc_t* d);
int launch_opt_reduce(hipStream_t s, const nasrec_opt_reduce_desc_t* d);
int launch_opt_apply(hipStream_t s, const nasrec_opt_apply_desc_t* d);
int launch_memset_chunks(hipStream_t s, const nasrec_memset_desc_t* d);
int launch_memset_flat(hipStream_t s, const nasrec_memset_desc_t* d);
int launch_const_i64(hipStream_t s, const nasrec_const_i64_desc_t* d);
int launch_splitk_epilogues(hipStream_t s, const nasrec_splitk_epilogues_desc_t* d);
int launch_worklist(hipStream_t s, const nasrec_worklist_desc_t* d);
int launch_worklist_dev(hipStream_t s, const nasrec_worklist_dev_desc_t* d);
int launch_dedup_ids(hipStream_t s, const nasrec_dedup_ids_desc_t* d);
int launch_opt_reduce2(hipStream_t s, const nasrec_opt_reduce2_desc_t* d);
int launch_persist(hipStream_t s, const nasrec_persist_desc_t* d);
