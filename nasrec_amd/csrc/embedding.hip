// Embedding stem kernels (reference: nasrec/supernet/supernet.py:404-430 forward; the backward +
// optimizer semantics of nn.Embedding(sparse=False) + clip_grad_norm_ + Adagrad, train_utils.py:283-286).
//
// Layout: Fs independent tables [rows_f, 16] fp32 (64-byte rows), indices int64 [B, Fs], one id per
// field per sample.  A row is moved by 4 lanes x float4, so a wavefront moves 16 rows = 1 KiB per
// instruction and the output [B,Fs,16] is written fully coalesced.
#include "common.h"
#include "optimizer_bodies.h"
#include "dedup_bodies.h"

// thread t of the gather: (pair = t >> 2 = (b, f), q = t & 3 = which float4 of the 64-byte row)
__device__ __forceinline__ void embed_gather_body(const nasrec_embed_desc_t& d, const int64_t* idx, int B, int Fs, long t) {
  const long pair = t >> 2;
  const int q = (int)(t & 3);
  if (pair >= (long)B * Fs) return;
  const int f = (int)(pair % Fs);
  long row = idx[pair];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row >= 0 && row < d.rows[f]) {
    v = *reinterpret_cast<const float4*>(d.table[f] + row * NASREC_EMB_DIM + q * 4);
  } else if (d.oob) {
    *d.oob = 1;  // torch raises IndexError; the host shim turns this flag into one
  }
  *reinterpret_cast<float4*>(d.out + pair * NASREC_EMB_DIM + q * 4) = v;
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const nasrec_embed_desc_t d) {
  embed_gather_body(d, d.idx, d.B, d.Fs, (long)blockIdx.x * 256 + threadIdx.x);
}

// per-step input staging + learning-rate store (one launch instead of three copies and a fill), optionally with the
// embedding gather of the step riding along — and, in `ids_blocks` more workgroups behind them, the id-only half of the optimizer's
// row dedup (dedup_bodies.h): the ids are in this launch's hands, and nothing reads that half's result before the backward is done
__global__ __launch_bounds__(256) void stage_inputs_kernel(const nasrec_stage_desc_t d, int base_blocks) {
  __shared__ __attribute__((aligned(16))) int dd_sidx[256];
  __shared__ int dd_sh[4];
  if ((int)blockIdx.x >= base_blocks) {
    dedup_ids_small_body(d.dedup_ids, d.cat_src, d.B, d.Fs, (int)blockIdx.x - base_blocks, dd_sidx, dd_sh);
    return;
  }
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int n_int = d.B * d.Fd, n_cat = d.B * d.Fs;
  if (t < n_int) d.int_dst[t] = d.int_src[t];
  if (t < n_cat) d.cat_dst[t] = d.cat_src[t];
  if (d.y_src != nullptr && t < d.B) d.y_dst[t] = d.y_src[t];
  if (t == 0 && d.lr_dst != nullptr) d.lr_dst[0] = d.lr;
  if (d.gather.out != nullptr) embed_gather_body(d.gather, d.cat_src, d.B, d.Fs, t);
}

static int dedup_ids_check(const nasrec_dedup_ids_desc_t* d, int B, int Fs, const char* who) {
  if (B < 1 || B > NASREC_DEDUP_IDS_MAX_B || Fs < 1 || Fs > NASREC_MAX_TABLES) return nasrec_set_error(-2, "%s: B=%d Fs=%d out of range", who, B, Fs);
  if (d->cap < 256 || d->cap > NASREC_DEDUP_IDS_MAX_B || (d->cap & (d->cap - 1)) || d->cap < B)
    return nasrec_set_error(-2, "%s: cap=%d must be a power of two in [max(256, B=%d), %d]", who, d->cap, B, NASREC_DEDUP_IDS_MAX_B);
  if (!d->leader || !d->order || !d->lists || !d->counts) return nasrec_set_error(-2, "%s: null output", who);
  return 0;
}

int launch_stage(hipStream_t st, const nasrec_stage_desc_t* d) {
  long n = (long)d->B * (d->Fd > d->Fs ? d->Fd : d->Fs);
  if (d->gather.out != nullptr) {
    if (d->Fs < 1 || d->Fs > NASREC_MAX_TABLES) return nasrec_set_error(-2, "stage: Fs=%d out of range", d->Fs);
    if ((long)d->B * d->Fs * 4 > n) n = (long)d->B * d->Fs * 4;
  }
  if (n < 1) n = 1;
  const int base = (int)((n + 255) / 256);
  int ids = 0;
  if (d->dedup_ids.order != nullptr) {
    if (d->B > 256 || d->dedup_ids.cap != 256) return nasrec_set_error(-2, "stage: the id half of the dedup rides along for B <= 256, cap 256 (B=%d cap=%d)", d->B, d->dedup_ids.cap);
    const int rc = dedup_ids_check(&d->dedup_ids, d->B, d->Fs, "stage.dedup_ids");
    if (rc) return rc;
    ids = d->Fs;
  }
  hipLaunchKernelGGL(stage_inputs_kernel, dim3((unsigned)(base + ids)), dim3(256), 0, st, *d, base);
  return nasrec_check_launch("stage_inputs");
}

int launch_embed_gather(hipStream_t st, const nasrec_embed_desc_t* d) {
  if (d->Fs < 1 || d->Fs > NASREC_MAX_TABLES) return nasrec_set_error(-2, "embed: Fs=%d out of range", d->Fs);
  long threads = (long)d->B * d->Fs * 4;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d);
  return nasrec_check_launch("embed_gather");
}

// Row-sparse backward: leader election + duplicate summation, per field.  "Leader" = first occurrence of a row id in
// sample order; its gsum row is the sum of the gradient rows of all occurrences in ascending sample order, so results are
// reproducible run to run and independent of how the batch was split over ranks.

// Batch <= 256 (one workgroup per field, thread = sample): every sample parks its id and its 64-byte gradient row in
// LDS with ONE parallel round of global loads; then each thread scans the ids in ascending order, four per
// ds_read_b128 (all lanes read the same address: broadcast): a match below b means "not the leader".  The LDS rows of a
// leader's later matches are added to its row in ascending b by sixteen lanes, one float of the row each, from an index list
// the group builds out of the leader's match mask (below): a 64-fold duplicate on a 4-row table costs 64 LDS reads per lane
// instead of 64 dependent global loads, and leaders without duplicates keep their row in registers.
// `chunk`: the body works on samples [256*chunk, 256*chunk + 256) (batches > 256 run it once per chunk and merge the
// chunk leaders afterwards, emb_dedup_merge_kernel); `final`: write the sum-of-squares partial of the leaders.
// LDS of dedup_small_body beyond sidx / rows / red (16-byte aligned): per 16-lane group the duplicate list of the leader it serves,
// the leaders' match masks, the leaders that have duplicates and their number
#define DEDUP_WORK_INTS (16 * 256 + 256 * 8 + 256 + 4)
__device__ __forceinline__ void dedup_small_body(const nasrec_emb_dedup_desc_t& d, int f, int chunk, bool final, int* sidx, float* rows,
                                                 float* red, int* work) {
  const int b = threadIdx.x;
  const long gb = (long)chunk * 256 + b;  // sample index in the batch
  const bool live = gb < d.B;
  // the id and the gradient row are loaded together (the id's LDS store waits for it: with the row loads behind that store they
  // were a second dependent round trip)
  const long gl = live ? gb : 0;
  f32x4 g[4];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(dd_row(const_cast<float*>(d.dout), (int)gl, f, d.Fs, d.rank_B, d.rank_stride));
#pragma unroll
    for (int v = 0; v < 4; ++v) g[v] = src[v];
  }
  const int my_in = (int)d.idx[gl * d.Fs + f];
  if (!live) {
#pragma unroll
    for (int v = 0; v < 4; ++v) g[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int my = live ? my_in : -1 - b;  // dead lanes get unique negative ids
  sidx[b] = my;
  if (b == 0) work[16 * 256 + 256 * 8 + 256] = 0;  // (number of leaders with duplicates, below)
#pragma unroll
  for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(&rows[b * 20 + 4 * v]) = g[v];
  __syncthreads();
  // branch-free scan: a 256-bit match mask per thread (the loop is fully unrolled, so the LDS reads pipeline)
  const int4* s4 = reinterpret_cast<const int4*>(sidx);
  unsigned mask[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) mask[w] = 0u;
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    const int4 v = s4[q];
    const unsigned m = (unsigned)(v.x == my) | ((unsigned)(v.y == my) << 1) | ((unsigned)(v.z == my) << 2) | ((unsigned)(v.w == my) << 3);
    mask[q >> 3] |= m << ((q & 7) * 4);
  }
  bool lead = live;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    // bits strictly below b in word w
    const unsigned below = (b >= 32 * (w + 1)) ? 0xffffffffu : (b <= 32 * w ? 0u : ((1u << (b - 32 * w)) - 1u));
    if (mask[w] & below) lead = false;
  }
  // Duplicate rows are added to their leader's row in ascending sample order, sixteen lanes per leader (lane = one float of the
  // row): a leader adding whole rows in its own lane spent ~200 clocks per duplicate on one wave — 6 of the launch's 12 us on the
  // bench's 4-row table, 21 us when all 256 ids are equal (the launch on distinct ids: 6.1 us).  Same additions in the
  // same order per element: same bits.
  int* glist = work;  // [16 groups][256]
  unsigned* dmask = reinterpret_cast<unsigned*>(work + 16 * 256);
  int* dlist = work + 16 * 256 + 256 * 8;
  int* ndup = dlist + 256;
  bool dup = false;
  if (lead) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      // bits strictly above b in word w
      const unsigned above = (b < 32 * w) ? 0xffffffffu : (b >= 32 * w + 31 ? 0u : ~((2u << (b - 32 * w)) - 1u));
      mask[w] &= above;
      dup = dup || mask[w] != 0u;
    }
    if (dup) {
      dlist[atomicAdd(ndup, 1)] = b;  // (which group of lanes serves which leader does not change any sum)
#pragma unroll
      for (int w = 0; w < 8; ++w) dmask[b * 8 + w] = mask[w];
    }
  }
  __syncthreads();
  const int nd = *ndup;
  if (nd > 0) {  // (uniform)
    const int e = b & 15, grp = b >> 4;
    int* gl = glist + grp * 256;
    for (int k = grp; k < nd; k += 16) {
      const int l = dlist[k];
      // the leader's duplicates as a list, ascending: lane e turns bits [16 e, 16 e + 16) of the mask into indices behind those of
      // the lanes below it (prefix sum over the group's 16 lanes = one DPP row); walking the mask bit by bit cost a dependent
      // ffs / LDS round trip per duplicate (125 clocks), the list is read four entries at a time (eight: no faster)
      unsigned bits = (dmask[l * 8 + (e >> 1)] >> (16 * (e & 1))) & 0xffffu;
      const int cnt = __popc(bits);
      int pre = cnt;
      pre += __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xf, 0xf, true);  // row_shr:1 (lanes without a source add 0)
      pre += __builtin_amdgcn_update_dpp(0, pre, 0x112, 0xf, 0xf, true);  // row_shr:2
      pre += __builtin_amdgcn_update_dpp(0, pre, 0x114, 0xf, 0xf, true);  // row_shr:4
      pre += __builtin_amdgcn_update_dpp(0, pre, 0x118, 0xf, 0xf, true);  // row_shr:8
      const int total = __shfl(pre, (threadIdx.x & 48) | 15, 64);
      int off = pre - cnt;
      while (bits) {
        gl[off++] = 16 * e + __ffs((int)bits) - 1;
        bits &= bits - 1;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the group's lanes are lanes of one wave: its LDS writes are in order
      float acc = rows[l * 20 + e];
      int i = 0;
      for (; i + 4 <= total; i += 4) {
        const int4 p = *reinterpret_cast<const int4*>(gl + i);
        const float r0 = rows[p.x * 20 + e], r1 = rows[p.y * 20 + e], r2 = rows[p.z * 20 + e], r3 = rows[p.w * 20 + e];
        acc += r0;
        acc += r1;
        acc += r2;
        acc += r3;
      }
      for (; i < total; ++i) acc += rows[gl[i] * 20 + e];
      rows[l * 20 + e] = acc;  // (a leader's row is nobody's duplicate: no other group reads it)
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the list is rewritten for the group's next leader)
    }
    __syncthreads();
    if (dup) {
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) g[vv] = *reinterpret_cast<const f32x4*>(&rows[b * 20 + 4 * vv]);
    }
  }
  float ss = 0.f;
  if (live) d.leader[gb * d.Fs + f] = lead ? 1 : 0;
  if (lead) {
    f32x4* dst = reinterpret_cast<f32x4*>(d.gsum + (gb * d.Fs + f) * 16);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      dst[v] = g[v];
      ss += g[v][0] * g[v][0] + g[v][1] * g[v][1] + g[v][2] * g[v][2] + g[v][3] * g[v][3];
    }
  }
  if (!final) return;
  red[b] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (b < o) red[b] += red[b + o];
    __syncthreads();
  }
  if (b == 0) d.sumsq_partial[f] = red[0];
}

__global__ __launch_bounds__(256) void emb_dedup_small_kernel(const nasrec_emb_dedup_desc_t d) {
  __shared__ __attribute__((aligned(16))) int sidx[256];
  __shared__ __attribute__((aligned(16))) float rows[256 * 20];  // 20-float rows: 16-byte aligned, bank-spread
  __shared__ float red[256];
  __shared__ __attribute__((aligned(16))) int work[DEDUP_WORK_INTS];
  dedup_small_body(d, blockIdx.x, 0, true, sidx, rows, red, work);
}

// NASREC_OP_OPT_REDUCE: workgroups [0, Fs) deduplicate one field each, the rest square-sum the dense gradient arena
__global__ __launch_bounds__(256) void opt_reduce_kernel(const nasrec_opt_reduce_desc_t d) {
  __shared__ __attribute__((aligned(16))) int sidx[256];
  __shared__ __attribute__((aligned(16))) float rows[256 * 20];
  __shared__ float red[256];
  __shared__ __attribute__((aligned(16))) int work[DEDUP_WORK_INTS];
  const int nd = d.dedup.Fs;
  if ((int)blockIdx.x < nd)
    dedup_small_body(d.dedup, blockIdx.x, 0, true, sidx, rows, red, work);
  else
    sumsq_body(d.sumsq, (int)blockIdx.x - nd, d.sumsq.nblocks, red);
}

int launch_opt_reduce(hipStream_t st, const nasrec_opt_reduce_desc_t* d) {
  if (d->dedup.B < 1 || d->dedup.B > 256) return nasrec_set_error(-2, "opt_reduce: B=%d outside [1,256]", d->dedup.B);
  if (d->sumsq.nblocks < 1) return nasrec_set_error(-2, "opt_reduce: sumsq.nblocks=%d", d->sumsq.nblocks);
  hipLaunchKernelGGL(opt_reduce_kernel, dim3(d->dedup.Fs + d->sumsq.nblocks), dim3(256), 0, st, *d);
  return nasrec_check_launch("opt_reduce");
}

// NASREC_OP_OPT_APPLY: clip coefficient (re-derived per workgroup) + Adagrad on the dense arena and on the touched rows
__global__ __launch_bounds__(256) void opt_apply_kernel(const nasrec_opt_apply_desc_t d) {
  __shared__ float sh_coef;
  const float lr = *d.dense.lr;  // (beside the partial sums' loads, not behind the barrier: one round trip less on every workgroup's path)
  if (threadIdx.x < 64) {
    float total;
    const float c = clip_coef_wave(d.clip, threadIdx.x, &total);
    if (threadIdx.x == 0) {
      sh_coef = c;
      if (blockIdx.x == 0) {
        d.clip.out[0] = c;
        d.clip.out[1] = total;
      }
    }
  }
  __syncthreads();
  const float coef = sh_coef;
  if ((int)blockIdx.x < d.dense_blocks)
    adagrad_dense_body(d.dense, blockIdx.x, d.dense_blocks, lr, coef);
  else
    adagrad_rows_body(d.rows, (int)blockIdx.x - d.dense_blocks, lr, coef);
}

int launch_opt_apply(hipStream_t st, const nasrec_opt_apply_desc_t* d) {
  const long threads = (long)d->rows.B * d->rows.Fs * 4;
  const int nrows = (int)((threads + 255) / 256);
  if (d->dense_blocks < 0 || d->dense_blocks + nrows < 1) return nasrec_set_error(-2, "opt_apply: empty launch");
  hipLaunchKernelGGL(opt_apply_kernel, dim3(d->dense_blocks + nrows), dim3(256), 0, st, *d);
  return nasrec_check_launch("opt_apply");
}

// Batches > 256 (the global batch of a data-parallel step, up to 8 x 8192 samples): O(B) instead of an all-pairs scan.
//   pass 1  grid (Fs, ceil(B/256)): the <= 256 kernel on every 256-sample chunk -> chunk leaders + their partial sums;
//   pass 2  grid (Fs, P): the chunk leaders are merged in chunk order through LDS hash tables keyed by row id.  The ids of a
//           field are PARTITIONED over P workgroups by a second hash, P = ceil(B / 4096), so a table of 16384 slots per
//           workgroup stays below 1/4 load at any batch size; every workgroup walks all chunk leaders of its field (ids and
//           flags only: B x 8 bytes from L2) and handles those of its partition.  Within one round (= one chunk) all candidates
//           have distinct ids (they lead their chunk), so a slot is claimed by exactly one thread (ds CAS) and an earlier
//           owner's row is updated by exactly one thread: no floating-point atomics, the sum runs in ascending chunk
//           (= sample) order, and the result does not depend on P.
#define DEDUP_SLOTS 16384
#define DEDUP_PART_ROWS 4096
#define DEDUP_UNROLL 8
__global__ __launch_bounds__(256) void emb_dedup_chunk_kernel(const nasrec_emb_dedup_desc_t d) {
  __shared__ __attribute__((aligned(16))) int sidx[256];
  __shared__ __attribute__((aligned(16))) float rows[256 * 20];
  __shared__ float red[256];
  __shared__ __attribute__((aligned(16))) int work[DEDUP_WORK_INTS];
  dedup_small_body(d, blockIdx.x, blockIdx.y, false, sidx, rows, red, work);
}

__global__ __launch_bounds__(256) void emb_dedup_merge_kernel(const nasrec_emb_dedup_desc_t d, int P) {
  __shared__ int keys[DEDUP_SLOTS];
  __shared__ int owner[DEDUP_SLOTS];
  __shared__ float red[256];
  const int f = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int nchunk = (d.B + 255) / 256;
  for (int i = tid; i < DEDUP_SLOTS; i += 256) keys[i] = -1;
  __syncthreads();
  float ss = 0.f;
  for (int c0 = 0; c0 < nchunk; c0 += DEDUP_UNROLL) {
    // this thread's candidates of the next rounds, fetched up front (one memory round trip per DEDUP_UNROLL rounds)
    int cid[DEDUP_UNROLL];
    bool mine[DEDUP_UNROLL];
#pragma unroll
    for (int u = 0; u < DEDUP_UNROLL; ++u) {
      const long b = (long)(c0 + u) * 256 + tid;
      const bool in = (c0 + u) < nchunk && b < d.B;
      cid[u] = in ? (int)d.idx[b * d.Fs + f] : 0;
      mine[u] = in && d.leader[b * d.Fs + f] != 0 && (int)((((unsigned)cid[u] * 2246822519u) >> 15) % (unsigned)P) == part;
    }
#pragma unroll
    for (int u = 0; u < DEDUP_UNROLL; ++u) {
      if (c0 + u < nchunk) {  // uniform
        const long b = (long)(c0 + u) * 256 + tid;
        if (mine[u]) {
          const int id = cid[u];
          unsigned h = ((unsigned)id * 2654435761u) >> 7;
          int ob = -1, probes = 0;
          for (;;) {
            h &= (unsigned)(DEDUP_SLOTS - 1);
            const int prev = atomicCAS(&keys[h], -1, id);
            if (prev == -1) {  // first occurrence of this row in the batch: stays the leader
              owner[h] = (int)b;
              break;
            }
            if (prev == id) {  // claimed in an earlier round (ids are distinct within a round)
              ob = owner[h];
              break;
            }
            ++h;
            if (++probes >= DEDUP_SLOTS) {  // table full: cannot happen below 16384 distinct ids per partition
              if (d.overflow) *d.overflow = 1;
              break;
            }
          }
          if (ob >= 0) {
            mine[u] = false;
            d.leader[b * d.Fs + f] = 0;
            const f32x4* src = reinterpret_cast<const f32x4*>(d.gsum + (b * d.Fs + f) * 16);
            f32x4* dst = reinterpret_cast<f32x4*>(d.gsum + ((long)ob * d.Fs + f) * 16);
#pragma unroll
            for (int v = 0; v < 4; ++v) dst[v] = dst[v] + src[v];
          }
        }
        __threadfence_block();
        __syncthreads();  // owner[] and the updated rows of this round are visible to the next one
      }
    }
  }
  // sum of squares of the rows that still lead after the merge (every leader belongs to exactly one partition)
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const long b = (long)c * 256 + tid;
    if (b < d.B && d.leader[b * d.Fs + f] != 0) {
      const int id = (int)d.idx[b * d.Fs + f];
      if ((int)((((unsigned)id * 2246822519u) >> 15) % (unsigned)P) == part) {
        const f32x4* g = reinterpret_cast<const f32x4*>(d.gsum + (b * d.Fs + f) * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const f32x4 t = g[v];
          ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
        }
      }
    }
  }
  red[tid] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  // the descriptor reserves ceil(B/256) >= P partials per field: partition p fills entry p, partition 0 zeroes the rest
  if (tid == 0) d.sumsq_partial[f * nchunk + part] = red[0];
  if (part == 0)
    for (int c = P + tid; c < nchunk; c += 256) d.sumsq_partial[f * nchunk + c] = 0.f;
}

int launch_emb_dedup(hipStream_t st, const nasrec_emb_dedup_desc_t* d) {
  if (d->B == 0) return 0;
  if (d->B <= 256) {
    hipLaunchKernelGGL(emb_dedup_small_kernel, dim3(d->Fs), dim3(256), 0, st, *d);
    return nasrec_check_launch("emb_dedup");
  }
  dim3 grid(d->Fs, (d->B + 255) / 256);
  hipLaunchKernelGGL(emb_dedup_chunk_kernel, grid, dim3(256), 0, st, *d);
  const int P = (d->B + DEDUP_PART_ROWS - 1) / DEDUP_PART_ROWS;
  hipLaunchKernelGGL(emb_dedup_merge_kernel, dim3(d->Fs, P), dim3(256), 0, st, *d, P);
  return nasrec_check_launch("emb_dedup");
}

__global__ __launch_bounds__(256) void adagrad_rows_kernel(const nasrec_adagrad_rows_desc_t d) {
  adagrad_rows_body(d, blockIdx.x, *d.lr, *d.coef);
}

int launch_adagrad_rows(hipStream_t st, const nasrec_adagrad_rows_desc_t* d) {
  long threads = (long)d->B * d->Fs * 4;
  if (threads == 0) return 0;
  hipLaunchKernelGGL(adagrad_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *d);
  return nasrec_check_launch("adagrad_rows");
}

// ---- the two-halves row-sparse backward (dedup_bodies.h) -----------------------------------------------------------
__global__ __launch_bounds__(256) void dedup_ids_small_kernel(const nasrec_dedup_ids_desc_t d) {
  __shared__ __attribute__((aligned(16))) int sidx[256];
  __shared__ int sh[4];
  dedup_ids_small_body(d, d.idx, d.B, d.Fs, blockIdx.x, sidx, sh);
}

// grid (chunks of 256 samples, Fs): the chunk index varies fastest, so the workgroups of one field (same ids) sit on neighbouring CUs
__global__ __launch_bounds__(256) void dedup_ids_pairs_kernel(const nasrec_dedup_ids_desc_t d) {
  __shared__ __attribute__((aligned(16))) int sidx[256];
  __shared__ int sh[4];
  __shared__ int htab[3 * DD_HASH];
  dedup_ids_pairs_body(d, d.idx, d.B, d.Fs, blockIdx.y, blockIdx.x, sidx, sh, htab, htab + DD_HASH, htab + 2 * DD_HASH);
}

int launch_dedup_ids(hipStream_t st, const nasrec_dedup_ids_desc_t* d) {
  const int rc = dedup_ids_check(d, d->B, d->Fs, "dedup_ids");
  if (rc) return rc;
  if (!d->idx) return nasrec_set_error(-2, "dedup_ids: null idx");
  if (d->B <= 256 && d->cap == 256) {
    hipLaunchKernelGGL(dedup_ids_small_kernel, dim3(d->Fs), dim3(256), 0, st, *d);
  } else {
    if (!d->heads) return nasrec_set_error(-2, "dedup_ids: B=%d > 256 needs the heads array", d->B);
    hipLaunchKernelGGL(dedup_ids_pairs_kernel, dim3((d->B + 255) / 256, d->Fs), dim3(256), 0, st, *d);
  }
  return nasrec_check_launch("dedup_ids");
}

// NASREC_OP_OPT_REDUCE2.  T threads per workgroup, RPT rows staged per thread: <256, 1> for B <= 256 (the one-GPU batch), <1024, 2>
// up to 2048 samples (the global batch of 8 ranks x 256): ALL rows of a field sit in LDS at once (2048 x 64 B = 128 KB), so both
// summation phases read LDS only.  A run / sub-run is summed by a QUAD of lanes (lane = one float4 of the row): 64 / 256 sums in
// flight per workgroup.  Dynamic LDS: rows [T RPT][16] floats | ord [T RPT] | lst [T RPT] | hds [T RPT] | red [T] | wave totals [T / 64][2].
#define OR2_LDS_BYTES(T, RPT) (4 * ((T) * (RPT) * 16 + 3 * (T) * (RPT) + (T) + 2 * ((T) / 64)))
extern __shared__ __attribute__((aligned(16))) float or2_lds[];

// sum of squares of x[0, n) (chunk tables as sumsq_body) with T threads -> partial[blk]; red: T floats
template <int T>
__device__ __forceinline__ void or2_dense_sumsq(const nasrec_sumsq_desc_t& d, int blk, int nblk, float* red) {
  if (T == 256) {
    sumsq_body(d, blk, nblk, red);
    return;
  }
  const int tid = threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (d.chunks) {
    for (long c = blk; c < d.nchunks; c += nblk) {
      const float* x = d.x + d.chunks[2 * c];
      const long n = d.chunks[2 * c + 1], n4 = n >> 2;
      for (long i = tid; i < n4; i += T) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * i);
        s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
      }
      for (long j = 4 * n4 + tid; j < n; j += T) s1 = fmaf(x[j], x[j], s1);
    }
  } else {
    const long stride = (long)nblk * T, n4 = d.n >> 2;
    long i = (long)blk * T + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d.x + 4 * i), b = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + stride));
      const f32x4 c = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + 2 * stride)), e = *reinterpret_cast<const f32x4*>(d.x + 4 * (i + 3 * stride));
      s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
      s1 += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
      s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
      s3 += (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
    }
    for (; i < n4; i += stride) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d.x + 4 * i);
      s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
    }
    for (long j = 4 * n4 + (long)blk * T + tid; j < d.n; j += stride) s1 = fmaf(d.x[j], d.x[j], s1);
  }
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) d.partial[blk] = red[0];
}

template <int T, int RPT>
__global__ __launch_bounds__(T) void opt_reduce2_kernel(const nasrec_opt_reduce2_desc_t d) {
  constexpr int ROWS = T * RPT;
  float* rows = or2_lds;
  int* ord = reinterpret_cast<int*>(or2_lds + ROWS * 16);
  int* lst = ord + ROWS;
  int* hds = lst + ROWS;
  float* red = reinterpret_cast<float*>(hds + ROWS);
  const int bid = blockIdx.x, tid = threadIdx.x;
  const int Fs = d.Fs, B = d.B;
  if (bid >= Fs + d.row_blocks) {  // dense gradient arena
    or2_dense_sumsq<T>(d.sumsq, bid - Fs - d.row_blocks, d.sumsq.nblocks, red);
    return;
  }
  float ss = 0.f;
  if (bid >= Fs) {
    // leaders without duplicates: their row is their own gradient.  4 lanes x float4 per row; flag and row are read together
    // (one round trip), rows of the other samples are read and dropped
    const long npair = (long)B * Fs;
    for (long t = (long)(bid - Fs) * T + tid; t < npair * 4; t += (long)d.row_blocks * T) {
      const long pair = t >> 2;
      const int q = (int)(t & 3);
      const int b = (int)(pair / Fs), f = (int)(pair - (long)b * Fs);
      const int lead = d.leader[pair];
      const f32x4 v = *reinterpret_cast<const f32x4*>(dd_row(d.rows, b, f, Fs, d.rank_B, d.rank_stride) + 4 * q);
      const float s4 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      ss += lead == 1 ? s4 : 0.f;
    }
  } else {
    // field f: every row of the field into LDS (one round of loads: RPT rows x 4 pieces per thread, beside the lists), then the sums
    const int f = bid, q = tid & 3, quad = tid >> 2;
    const int cap = d.cap;
    const int nA = d.counts[2 * f];
    f32x4 g[RPT][4];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int b = min(u * T + tid, B - 1);
      const f32x4* src = reinterpret_cast<const f32x4*>(dd_row(d.rows, b, f, Fs, d.rank_B, d.rank_stride));
#pragma unroll
      for (int v = 0; v < 4; ++v) g[u][v] = src[v];
    }
    int ov[RPT], lv[RPT], hv[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = u * T + tid;
      ov[u] = i < cap ? d.order[(long)f * cap + i] : 0;
      lv[u] = i < (RPT == 1 ? cap : min(cap, (B + 255) & ~255)) ? d.lists[(long)f * cap + i] : 0;  // (per-sample entries exist for the chunks of the batch only)
      hv[u] = (i < cap && d.heads) ? d.heads[(long)f * cap + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = u * T + tid;
      ord[i] = ov[u];
      lst[i] = lv[u];
      hds[i] = hv[u];
#pragma unroll
      for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(&rows[i * 16 + 4 * v]) = g[u][v];
    }
    __syncthreads();
    // B > 256: the per-sample entries (dedup_ids_pairs_body) are compacted first — a ballot per staged entry, wave totals through LDS:
    // a fixed order, so every thread sums the same squares in every run — into the sub-runs with >= 2 members (listA, in `red`'s
    // storage until the final reduction) and the leaders of multi-chunk runs (listB, over `ord` once phase 1 is done with it)
    int nA1 = nA, nB1 = 0, posB[RPT];
    int* listA = reinterpret_cast<int*>(red);
    if (RPT > 1) {
      int* wtot = reinterpret_cast<int*>(red + T);  // [T / 64][2]
      const int lane = tid & 63, wv = tid >> 6;
      const unsigned long long below = (1ull << lane) - 1ull;
      int pa[RPT], pb[RPT], ca = 0, cb = 0;
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const unsigned long long ma = __ballot(((unsigned)lv[u] & DD_A) != 0u), mb = __ballot(((unsigned)lv[u] & DD_MULTI) != 0u);
        pa[u] = ca + __popcll(ma & below);
        pb[u] = cb + __popcll(mb & below);
        ca += __popcll(ma);
        cb += __popcll(mb);
      }
      if (lane == 0) {
        wtot[2 * wv] = ca;
        wtot[2 * wv + 1] = cb;
      }
      __syncthreads();
      int ba = 0, bb = 0;
      nA1 = 0;
      for (int w = 0; w < T / 64; ++w) {
        const int xa = wtot[2 * w], xb = wtot[2 * w + 1];
        ba += w < wv ? xa : 0;
        bb += w < wv ? xb : 0;
        nA1 += xa;
        nB1 += xb;
      }
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        if ((unsigned)lv[u] & DD_A) listA[ba + pa[u]] = u * T + tid;
        posB[u] = ((unsigned)lv[u] & DD_MULTI) ? bb + pb[u] : -1;
      }
      __syncthreads();
    }
    // phase 1: every sub-run with >= 2 members, a quad per sub-run (lane = one float4 of the row), ascending sample order; the sum
    // lands in the LDS row of the sub-run's first sample, and in memory when the sub-run is the whole run
    for (int k0 = quad; k0 < nA1; k0 += T / 4) {
      const int k = RPT == 1 ? k0 : listA[k0];
      const unsigned en = (unsigned)lst[k];
      int s, len;
      if (RPT == 1) {
        s = (int)(en & 0xffffu);
        len = (int)((en >> 16) & 0x7fffu);
      } else {
        s = (k & ~255) + (int)(en & 0xffu);
        len = 1 + (int)((en >> 8) & 0xffu);
      }
      const int o0 = ord[s];
      f32x4 acc = *reinterpret_cast<const f32x4*>(&rows[o0 * 16 + 4 * q]);
      int i = 1;
      for (; i + 4 <= len; i += 4) {
        const int p0 = ord[s + i], p1 = ord[s + i + 1], p2 = ord[s + i + 2], p3 = ord[s + i + 3];
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(&rows[p0 * 16 + 4 * q]), r1 = *reinterpret_cast<const f32x4*>(&rows[p1 * 16 + 4 * q]);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(&rows[p2 * 16 + 4 * q]), r3 = *reinterpret_cast<const f32x4*>(&rows[p3 * 16 + 4 * q]);
        acc += r0;
        acc += r1;
        acc += r2;
        acc += r3;
      }
      for (; i < len; ++i) acc += *reinterpret_cast<const f32x4*>(&rows[ord[s + i] * 16 + 4 * q]);
      if (en & DD_WHOLE) {  // the run ends here: this is the leader's final row
        *reinterpret_cast<f32x4*>(dd_row(d.rows, o0, f, Fs, d.rank_B, d.rank_stride) + 4 * q) = acc;
        ss += (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
      } else {
        *reinterpret_cast<f32x4*>(&rows[o0 * 16 + 4 * q]) = acc;  // (a sub-run's first row is nobody else's operand in this phase)
      }
    }
    if (RPT > 1 && nB1 > 0) {  // (uniform) phase 2: runs that span chunks — the sub-run sums, in chunk order, into the leader's row
      __syncthreads();
#pragma unroll
      for (int u = 0; u < RPT; ++u)
        if (posB[u] >= 0) ord[posB[u]] = u * T + tid;
      __syncthreads();
      for (int k0 = quad; k0 < nB1; k0 += T / 4) {
        const int k = ord[k0];
        f32x4 acc = *reinterpret_cast<const f32x4*>(&rows[k * 16 + 4 * q]);
        for (int h = hds[k]; h >= 0; h = hds[h]) acc += *reinterpret_cast<const f32x4*>(&rows[h * 16 + 4 * q]);
        *reinterpret_cast<f32x4*>(dd_row(d.rows, k, f, Fs, d.rank_B, d.rank_stride) + 4 * q) = acc;
        ss += (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
      }
    }
  }
  __syncthreads();
  red[tid] = ss;
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) d.sumsq_partial[bid] = red[0];
}

int launch_opt_reduce2(hipStream_t st, const nasrec_opt_reduce2_desc_t* d) {
  if (d->B < 1 || d->B > NASREC_DEDUP_IDS_MAX_B || d->Fs < 1 || d->Fs > NASREC_MAX_TABLES) return nasrec_set_error(-2, "opt_reduce2: B=%d Fs=%d out of range", d->B, d->Fs);
  if (d->cap < 256 || d->cap > NASREC_DEDUP_IDS_MAX_B || (d->cap & (d->cap - 1)) || d->cap < d->B) return nasrec_set_error(-2, "opt_reduce2: cap=%d", d->cap);
  if (d->row_blocks < 1 || d->sumsq.nblocks < 0) return nasrec_set_error(-2, "opt_reduce2: row_blocks=%d sumsq.nblocks=%d", d->row_blocks, d->sumsq.nblocks);
  if (d->rank_B < 0 || (d->rank_B > 0 && d->rank_stride < (int64_t)d->rank_B * d->Fs * 16)) return nasrec_set_error(-2, "opt_reduce2: rank layout %d / %ld", d->rank_B, (long)d->rank_stride);
  if (!d->rows || !d->leader || !d->order || !d->lists || !d->counts || !d->sumsq_partial) return nasrec_set_error(-2, "opt_reduce2: null pointer");
  if (d->B > 256 && !d->heads) return nasrec_set_error(-2, "opt_reduce2: B=%d > 256 needs the heads array", d->B);
  const dim3 grid((unsigned)(d->Fs + d->row_blocks + d->sumsq.nblocks));
  if (d->cap <= 256) {
    hipLaunchKernelGGL((opt_reduce2_kernel<256, 1>), grid, dim3(256), OR2_LDS_BYTES(256, 1), st, *d);
  } else {
    static unsigned long long attr_mask = 0;
    if (nasrec_lds_attr_needed(attr_mask)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&opt_reduce2_kernel<1024, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, OR2_LDS_BYTES(1024, 2));
      if (e != hipSuccess) return nasrec_set_error((int)e, "opt_reduce2: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((opt_reduce2_kernel<1024, 2>), grid, dim3(1024), OR2_LDS_BYTES(1024, 2), st, *d);
  }
  return nasrec_check_launch("opt_reduce2");
}
