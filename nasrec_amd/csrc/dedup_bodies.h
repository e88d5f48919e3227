// Device bodies of the two-halves row-sparse embedding backward (include/nasrec_hip.h: NASREC_OP_DEDUP_IDS, NASREC_OP_OPT_REDUCE2).
// Reference semantics: nn.Embedding(sparse=False) backward + clip_grad_norm_ + Adagrad (nasrec/supernet/supernet.py:404-410,
// nasrec/utils/train_utils.py:283-286): a table row touched by several samples of the batch receives the SUM of their gradient rows.
//
// Which sample leads a row and which samples repeat it depends on the ids only, so that half runs long before the backward pass is
// done (on the staging launch of a one-GPU step; behind the ids all-gather of a data-parallel step) and leaves per field
//   order[p]   the samples sorted by (id, sample): a RUN = the samples of one id, ascending; a SUB-RUN = the part of a run inside one
//              256-sample chunk of the batch (bit 31 marks the first sample of a sub-run),
//   list A     the sub-runs with >= 2 members, list B the runs with >= 2 sub-runs (start position | length << 16),
//   leader     0 duplicate / 1 leader without duplicates / 2 leader with duplicates.
// What stays behind the backward pass is the arithmetic: every sub-run summed into its first row in ascending sample order, then the
// sub-run sums of a multi-chunk run into the leader's row in chunk order — exactly the order of the one-launch kernels
// (embedding.hip: emb_dedup_small / chunk + merge), so the summed rows are the same bits at every batch size and do not depend on how
// the batch was split over ranks — and the sum of squares for the clip.
#pragma once
#include "common.h"

#define DD_WHOLE 0x80000000u   // list A entry: the sub-run is its whole run (the leader's row is final after phase 1)
#define DD_HEAD 0x80000000u    // order entry: first sample of a sub-run

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

// The id-only half for field f, one workgroup of 256 threads.  key: CAP 64-bit LDS words; sh: 4 ints.  d.cap (a power of two,
// 256 <= d.cap <= CAP) entries are sorted; B <= d.cap samples are real.
template <int CAP>
__device__ __forceinline__ void dedup_ids_body(const nasrec_dedup_ids_desc_t& d, const int64_t* idx, int B, int Fs, int f, unsigned long long* key, int* sh) {
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
  unsigned ea[PER], eb[PER];
  int cnt = 0;  // list A entries | list B entries << 16 of this thread
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    ea[u] = eb[u] = 0u;
    const int p = tid * per + u;
    if (u < per && p < B) {
      const unsigned long long k = key[p];
      const unsigned id = (unsigned)(k >> 16);
      const int b = (int)(k & 0xffffu);
      const unsigned long long kp = p > 0 ? key[p - 1] : ~k, kn = p + 1 < B ? key[p + 1] : ~k;
      const bool run_start = p == 0 || (unsigned)(kp >> 16) != id, run_end = p + 1 >= B || (unsigned)(kn >> 16) != id;
      const bool sub_start = run_start || (int)((kp & 0xffffu) >> 8) != (b >> 8), sub_end = run_end || (int)((kn & 0xffffu) >> 8) != (b >> 8);
      d.order[(long)f * n + p] = (int)((unsigned)b | (sub_start ? DD_HEAD : 0u));
      if (!run_start) d.leader[(long)b * Fs + f] = 0;
      int s_run = p;
      if (run_end) {
        s_run = run_start ? p : dd_lower_bound(key, B, (unsigned long long)id << 16);
        const int bs = (int)(key[s_run] & 0xffffu);
        d.leader[(long)bs * Fs + f] = p > s_run ? 2 : 1;
        if ((bs >> 8) != (b >> 8)) {  // the run spans chunks
          eb[u] = (unsigned)s_run | ((unsigned)(p - s_run + 1) << 16);
          cnt += 1 << 16;
        }
      }
      if (sub_end && !sub_start) {  // a sub-run with >= 2 members ends here
        const int s_sub = dd_lower_bound(key, B, ((unsigned long long)id << 16) | (unsigned)(b & ~255));
        ea[u] = (unsigned)s_sub | ((unsigned)(p - s_sub + 1) << 16) | ((run_end && s_sub == s_run) ? DD_WHOLE : 0u);
        cnt += 1;
      }
    }
  }
  int total;
  int pos = dd_block_excl_scan(cnt, sh, &total);
  int pa = pos & 0xffff, pb = pos >> 16;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (ea[u]) d.lists[(long)f * n + pa++] = (int)ea[u];
    if (eb[u]) d.lists[(long)f * n + half + pb++] = (int)eb[u];
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
