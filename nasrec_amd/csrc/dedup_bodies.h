// Device bodies of the two-halves row-sparse embedding backward (include/nasrec_hip.h: NASREC_OP_DEDUP_IDS, NASREC_OP_OPT_REDUCE2).
// Reference semantics: nn.Embedding(sparse=False) backward + clip_grad_norm_ + Adagrad (nasrec/supernet/supernet.py:404-410,
// nasrec/utils/train_utils.py:283-286): a table row touched by several samples of the batch receives the SUM of their gradient rows.
//
// Which sample leads a row and which samples repeat it depends on the ids only, so that half runs long before the backward pass is
// done (on the staging launch of a one-GPU step; behind the ids all-gather of a data-parallel step) and leaves per field
//   leader     0 duplicate / 1 leader without duplicates / 2 leader with duplicates,
//   order      the samples of the rows with duplicates, run by run: a RUN = the samples of one id in ascending order; a SUB-RUN = the
//              part of a run inside one 256-sample chunk of the batch,
//   list A     the sub-runs with >= 2 members (start position in `order` | length << 16 | DD_WHOLE: the sub-run is its whole run),
//   list B     the runs with >= 2 sub-runs (start position in `heads` | number of sub-runs << 16),
//   heads      for the runs of list B: the first sample of every sub-run, in chunk order.
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
  const int my = live ? (int)idx[(long)b * Fs + f] : -1 - b;  // dead lanes get unique negative ids
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

// ---- 256 < B <= NASREC_DEDUP_IDS_MAX_B: sort (id, sample) keys ---------------------------------------------------------------------
// first position p of the sorted keys [0, n) with key[p] >= t
__device__ __forceinline__ int dd_lower_bound(const unsigned long long* key, int n, unsigned long long t) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (key[mid] < t) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// One workgroup of 256 threads per field.  key: CAP 64-bit LDS words; sh: 4 ints.  d.cap (a power of two, 256 <= d.cap <= CAP)
// entries are sorted; B <= d.cap samples are real.  Here `order` is simply the sorted order (all B samples): runs are contiguous.
template <int CAP>
__device__ __forceinline__ void dedup_ids_sort_body(const nasrec_dedup_ids_desc_t& d, const int64_t* idx, int B, int Fs, int f, unsigned long long* key, int* sh) {
  constexpr int T = 256, PER = CAP / T;
  const int tid = threadIdx.x;
  const int n = d.cap, half = n >> 1;
  // keys (id, sample): ids are row numbers below 2^31 (engine.py), the sample index fits 16 bits; padding sorts last
  for (int i = tid; i < n; i += T) {
    unsigned long long k = ~0ull;
    if (i < B) k = ((unsigned long long)(unsigned)idx[(long)i * Fs + f] << 16) | (unsigned)i;
    key[i] = k;
  }
  // bitonic sort, ascending (a compare-exchange network: the result does not depend on timing)
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < half; t += T) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const unsigned long long a = key[i], b = key[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          key[i] = b;
          key[l] = a;
        }
      }
    }
  }
  __syncthreads();
  // every thread looks at PER consecutive sorted positions
  const int per = n / T;  // (>= 1)
  unsigned ea[PER];   // list A entry that ENDS at this position (0: none)
  int srun[PER];      // >= 0: a multi-chunk run ends at this position and started at srun
  int hb[PER];        // the sample, if this position is a sub-run head of a multi-chunk run (else -1)
  int cnt = 0, cnth = 0;  // list A entries | list B entries << 16; heads
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    ea[u] = 0u;
    srun[u] = hb[u] = -1;
    const int p = tid * per + u;
    if (u < per && p < B) {
      const unsigned long long k = key[p];
      const unsigned id = (unsigned)(k >> 16);
      const int b = (int)(k & 0xffffu);
      const unsigned long long kp = p > 0 ? key[p - 1] : ~k, kn = p + 1 < B ? key[p + 1] : ~k;
      const bool run_start = p == 0 || (unsigned)(kp >> 16) != id, run_end = p + 1 >= B || (unsigned)(kn >> 16) != id;
      const bool sub_start = run_start || (int)((kp & 0xffffu) >> 8) != (b >> 8), sub_end = run_end || (int)((kn & 0xffffu) >> 8) != (b >> 8);
      d.order[(long)f * n + p] = b;
      int s_run = p, e_run = p;
      if (!run_start) s_run = dd_lower_bound(key, B, (unsigned long long)id << 16);
      if (!run_end) e_run = dd_lower_bound(key, B, ((unsigned long long)id + 1) << 16) - 1;
      const bool multi = (int)((key[s_run] & 0xffffu) >> 8) != (int)((key[e_run] & 0xffffu) >> 8);  // the run spans chunks
      d.leader[(long)b * Fs + f] = run_start ? (e_run > s_run ? 2 : 1) : 0;
      if (multi && sub_start) {
        hb[u] = b;
        ++cnth;
      }
      if (multi && run_end) {
        srun[u] = s_run;
        cnt += 1 << 16;
      }
      if (sub_end && !sub_start) {  // a sub-run with >= 2 members ends here
        const int s_sub = dd_lower_bound(key, B, ((unsigned long long)id << 16) | (unsigned)(b & ~255));
        ea[u] = (unsigned)s_sub | ((unsigned)(p - s_sub + 1) << 16) | (multi ? 0u : DD_WHOLE);
        cnt += 1;
      }
    }
  }
  int total, totalh;
  const int pos = dd_block_excl_scan(cnt, sh, &total);
  const int ph = dd_block_excl_scan(cnth, sh, &totalh);
  // The heads of one run are consecutive in `heads` (sorted order): a run [s, e] owns heads [hx[s], hx[e] + head(e)), hx[p] = heads at
  // positions < p.  hx is parked over the keys (every thread is done with them: the scans above end with barriers).
  int* hx = reinterpret_cast<int*>(key);
  __syncthreads();
  {
    int run = ph;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int p = tid * per + u;
      if (u < per && p < n) hx[p] = run;
      if (hb[u] >= 0) {
        d.heads[(long)f * n + run] = hb[u];
        ++run;
      }
    }
  }
  __syncthreads();
  int pa = pos & 0xffff, pb = pos >> 16;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int p = tid * per + u;
    if (ea[u]) d.lists[(long)f * n + pa++] = (int)ea[u];
    if (srun[u] >= 0) {
      const int h0 = hx[srun[u]], nh = hx[p] + (hb[u] >= 0 ? 1 : 0) - h0;
      d.lists[(long)f * n + half + pb++] = (int)((unsigned)h0 | ((unsigned)nh << 16));
    }
  }
  if (tid == 0) {
    d.counts[2 * f] = total & 0xffff;
    d.counts[2 * f + 1] = total >> 16;
  }
}

// address of row (b, f) of the per-sample row gradients (contiguous, or the receive buffer of an all-gather: nasrec_adagrad_rows_desc_t)
__device__ __forceinline__ float* dd_row(float* rows, int b, int f, int Fs, int rank_B, long rank_stride) {
  if (rank_B > 0) {
    const int r = b / rank_B;
    return rows + (long)r * rank_stride + ((long)(b - r * rank_B) * Fs + f) * 16;
  }
  return rows + ((long)b * Fs + f) * 16;
}
