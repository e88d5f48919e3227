// Device bodies of the two-halves row-sparse embedding backward (include/nasrec_hip.h: NASREC_OP_DEDUP_IDS, NASREC_OP_OPT_REDUCE2).
// Reference semantics: nn.Embedding(sparse=False) backward + clip_grad_norm_ + Adagrad (nasrec/supernet/supernet.py:404-410,
// nasrec/utils/train_utils.py:283-286): a table row touched by several samples of the batch receives the SUM of their gradient rows.
//
// Which sample leads a row and which samples repeat it depends on the ids only, so that half runs long before the backward pass is
// done (on the staging launch of a one-GPU step; behind the ids all-gather of a data-parallel step) and leaves per field
//   leader     0 duplicate / 1 leader without duplicates / 2 leader with duplicates,
//   order      the samples of the rows with duplicates, run by run: a RUN = the samples of one id in ascending order; a SUB-RUN = the
//              part of a run inside one 256-sample chunk of the batch,
//   B <= 256 (every run is one sub-run): list A = the runs with >= 2 members (start position in `order` | length << 16 | DD_WHOLE);
//   above: per-sample entries and a next-sub-run link per sample instead of lists (dedup_ids_pairs_body below).
// What stays behind the backward pass is the arithmetic: every sub-run summed into its first row in ascending sample order, then the
// sub-run sums of a multi-chunk run into the leader's row in chunk order — exactly the order of the one-launch kernels
// (embedding.hip: emb_dedup_small / chunk + merge), so the summed rows are the same bits at every batch size and do not depend on how
// the batch was split over ranks — and the sum of squares for the clip.
#pragma once
#include "common.h"

#define DD_WHOLE 0x80000000u   // list A entry: the sub-run is its whole run (the leader's row is final after phase 1)

// exclusive prefix sum of one int per thread over a 256-thread workgroup (fixed order); *total = the sum.  sh: 4 ints of LDS.
__device__ __forceinline__ int dd_block_excl_scan(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // (sh may still be read from an earlier call)
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < w; ++q) base += sh[q];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  return base + inc - v;
}

// ---- B <= 256 (one chunk: every run is one sub-run): all-pairs match masks, no sort -----------------------------------------------
// Thread = sample.  Every thread compares its id with all 256 (four per ds_read_b128, all lanes read the same address: broadcast) into
// a 256-bit mask; no bit below the own position = leader; the bits above = its duplicates, ascending.  One global round trip (the
// ids), ~70 LDS reads, one block scan: short enough to ride on the step's staging launch.  sidx: 256 ints of LDS, sh: 4.
__device__ __forceinline__ void dedup_ids_small_body(const nasrec_dedup_ids_desc_t& d, const int64_t* idx, int B, int Fs, int f, int* sidx, int* sh) {
  const int b = threadIdx.x;
  const bool live = b < B;
  // ids are compared as 32-bit values (tables have < 2^31 rows: engine.py).  An id outside [0, 2^31) is out of range for every table —
  // the gather has raised the oob flag and the apply never writes outside a table — but narrowed it could collide with a valid id, with
  // a dead lane's synthetic id or with -1: such a sample gets a unique negative id of its own (a run of one that the apply skips).
  const int64_t raw = live ? idx[(long)b * Fs + f] : 0;
  const int my = !live ? -1 - b : ((raw >= 0 && raw <= 0x7fffffffL) ? (int)raw : (int)0x80000000u + b);  // dead lanes get unique negative ids
  sidx[b] = my;
  __syncthreads();
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
  int ndup = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const unsigned below = (b >= 32 * (w + 1)) ? 0xffffffffu : (b <= 32 * w ? 0u : ((1u << (b - 32 * w)) - 1u));
    const unsigned above = (b < 32 * w) ? 0xffffffffu : (b >= 32 * w + 31 ? 0u : ~((2u << (b - 32 * w)) - 1u));
    const unsigned valid = (B >= 32 * (w + 1)) ? 0xffffffffu : (B <= 32 * w ? 0u : ((1u << (B - 32 * w)) - 1u));  // (samples that exist)
    if (mask[w] & below) lead = false;
    mask[w] &= above & valid;
    ndup += __popc(mask[w]);
  }
  ndup = lead ? ndup : 0;
  if (live) d.leader[(long)b * Fs + f] = lead ? (ndup ? 2 : 1) : 0;
  int total;
  const int pos = dd_block_excl_scan(ndup ? ((1 + ndup) | (1 << 16)) : 0, sh, &total);  // order slots | list entries << 16
  if (ndup) {
    const int start = pos & 0xffff;
    int* o = d.order + (long)f * d.cap + start;
    d.lists[(long)f * d.cap + (pos >> 16)] = (int)((unsigned)start | ((unsigned)(1 + ndup) << 16) | DD_WHOLE);
    *o++ = b;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      unsigned m = mask[w];
      while (m) {
        *o++ = 32 * w + __ffs((int)m) - 1;
        m &= m - 1;
      }
    }
  }
  if (b == 0) {
    d.counts[2 * f] = total >> 16;
    d.counts[2 * f + 1] = 0;
  }
}

// ---- 256 < B <= NASREC_DEDUP_IDS_MAX_B: one workgroup per (field, 256-sample chunk) ----------------------------------------------------
// No workgroup needs another's result, so nothing is sorted, scanned or signalled across workgroups: every array is indexed by SAMPLE
// (or by the chunk's own region of `order`), and a thread decides everything about its sample from the ids of its field:
//   own chunk      256-bit match mask as in the small body -> head of its sub-run? the sub-run's members (ascending)
//   other chunks   through an LDS hash table of their ids (open addressing, 4096 slots for <= 1792 ids): per id, "occurs in an earlier
//                  chunk" (-> the sample does not lead its run) and the LOWEST sample of the later chunks (= the head of the run's next
//                  sub-run), kept with atomicMin — both independent of the order the table was filled in.  (All pairs against the
//                  other chunks, 2048 compares per thread, took 37 us at 2048 x 26: the broadcast LDS reads alone are ~8 clocks each.)
// Per field:  lists[b]  DD_A | DD_MULTI | DD_WHOLE | (members - 1) << 8 | position of the sub-run inside the chunk's region of `order`
//             heads[b]  the head of the next sub-run of b's run (-1: none): phase 2 of the sums walks leader -> next -> next ...
//             order[256 c + ..]  the members of chunk c's sub-runs with >= 2 members;   counts: unused (0)
#define DD_A 0x10000u      // the sample heads a sub-run with >= 2 members
#define DD_MULTI 0x20000u  // the sample leads a run with sub-runs in later chunks
#define DD_HASH 4096
__device__ __forceinline__ int dd_hash(int id) { return (int)(((unsigned)id * 2654435761u) >> 20); }

// sidx: 256 ints of LDS (the chunk's ids), sh: 4 ints, hkey / hlo / hearly: DD_HASH ints each
__device__ __forceinline__ void dedup_ids_pairs_body(const nasrec_dedup_ids_desc_t& d, const int64_t* idx, int B, int Fs, int f, int c, int* sidx, int* sh,
                                                     int* hkey, int* hlo, int* hearly) {
  const int t = threadIdx.x, b = c * 256 + t;
  const int nch = (B + 255) >> 8;
  int v[NASREC_DEDUP_IDS_MAX_B / 256];  // the field's ids, sample 256 u + t: every load first
#pragma unroll
  for (int u = 0; u < NASREC_DEDUP_IDS_MAX_B / 256; ++u) {
    const int i = u * 256 + t;
    const int64_t raw = i < B ? idx[(long)i * Fs + f] : -1;
    v[u] = (raw >= 0 && raw <= 0x7fffffffL) ? (int)raw : -1;  // (an id outside [0, 2^31) is in no table: treated like a sample that does not exist, see dedup_ids_small_body)
  }
  for (int i = t; i < DD_HASH; i += 256) {
    hkey[i] = -1;
    hlo[i] = 0x7fffffff;
    hearly[i] = 0;
  }
#pragma unroll
  for (int u = 0; u < NASREC_DEDUP_IDS_MAX_B / 256; ++u)
    if (u == c) sidx[t] = v[u] >= 0 ? v[u] : -1 - t;  // (samples that do not exist: unique negative ids)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NASREC_DEDUP_IDS_MAX_B / 256; ++u) {
    const int id = v[u];
    if (u < nch && u != c && id >= 0) {
      int slot = dd_hash(id);
      for (;;) {
        const int old = atomicCAS(&hkey[slot], -1, id);
        if (old == -1 || old == id) break;
        slot = (slot + 1) & (DD_HASH - 1);
      }
      if (u < c) hearly[slot] = 1;
      else atomicMin(&hlo[slot], u * 256 + t);
    }
  }
  __syncthreads();
  const bool live = b < B;
  const int my = sidx[t];
  bool early = false;
  int nxt = -1;
  if (live) {
    int slot = dd_hash(my);
    for (;;) {
      const int k = hkey[slot];
      if (k == my) {
        early = hearly[slot] != 0;
        const int lo = hlo[slot];
        nxt = lo == 0x7fffffff ? -1 : lo;
        break;
      }
      if (k == -1) break;
      slot = (slot + 1) & (DD_HASH - 1);
    }
  }
  const int4* s4 = reinterpret_cast<const int4*>(sidx);
  // own chunk
  unsigned mask[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) mask[w] = 0u;
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    const int4 w4 = s4[q];
    const unsigned m = (unsigned)(w4.x == my) | ((unsigned)(w4.y == my) << 1) | ((unsigned)(w4.z == my) << 2) | ((unsigned)(w4.w == my) << 3);
    mask[q >> 3] |= m << ((q & 7) * 4);
  }
  bool head = live;  // head of its sub-run
  int ndup = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const unsigned below = (t >= 32 * (w + 1)) ? 0xffffffffu : (t <= 32 * w ? 0u : ((1u << (t - 32 * w)) - 1u));
    const unsigned above = (t < 32 * w) ? 0xffffffffu : (t >= 32 * w + 31 ? 0u : ~((2u << (t - 32 * w)) - 1u));
    if (mask[w] & below) head = false;
    mask[w] &= above;
    ndup += __popc(mask[w]);
  }
  ndup = head ? ndup : 0;
  const bool lead = head && !early;
  if (live) d.leader[(long)b * Fs + f] = lead ? ((ndup || nxt >= 0) ? 2 : 1) : 0;
  d.heads[(long)f * d.cap + b] = head ? nxt : -1;
  int total;
  const int pos = dd_block_excl_scan(ndup ? 1 + ndup : 0, sh, &total);
  unsigned en = 0u;
  if (ndup) {
    en = DD_A | (unsigned)pos | ((unsigned)ndup << 8) | ((lead && nxt < 0) ? DD_WHOLE : 0u);
    int* o = d.order + (long)f * d.cap + c * 256 + pos;
    *o++ = b;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      unsigned m = mask[w];
      while (m) {
        *o++ = c * 256 + 32 * w + __ffs((int)m) - 1;
        m &= m - 1;
      }
    }
  }
  if (lead && nxt >= 0) en |= DD_MULTI;
  d.lists[(long)f * d.cap + b] = (int)en;
  if (t == 0 && c == 0) d.counts[2 * f] = d.counts[2 * f + 1] = 0;
}

// address of row (b, f) of the per-sample row gradients (contiguous, or the receive buffer of an all-gather: nasrec_adagrad_rows_desc_t)
__device__ __forceinline__ float* dd_row(float* rows, int b, int f, int Fs, int rank_B, long rank_stride) {
  if (rank_B > 0) {
    const int r = b / rank_B;
    return rows + (long)r * rank_stride + ((long)(b - r * rank_B) * Fs + f) * 16;
  }
  return rows + ((long)b * Fs + f) * 16;
}
