// Go / no-go probe for VERDICT r5 item 1 ("make the batch-256 step one persistent launch with in-kernel dependencies"):
// what does a dependency SEAM cost inside one launch, against the kernel boundary it would replace, in the geometry of the cfg-2 step
// (256-thread workgroups, 3 - 4 per CU, ~512 workgroups per level, every level reads what OTHER workgroups — on other XCDs — wrote)?
//
// P phases.  Phase p, workgroup w: reads a 32 x 128 tile of X[p-1] whose two column halves were written by two different workgroups of
// phase p-1 (the row block is permuted from phase to phase, so producer and consumer sit on different XCDs), multiplies it with a
// 128 x 64 weight panel (read-only) and writes a 32 x 64 piece of X[p].  ~2 - 4 us of work per workgroup: a small level of the step.
//   launches   P kernel launches on one stream (plain loads / stores)                                   -- what the engine does today
//   level      ONE launch of P x G workgroups in phase-major order; a workgroup waits until the per-phase counter of phase p-1 has
//              reached G (sc1 poll by one lane + barrier), handed-off bytes are stored write-through (sc0 sc1) and loaded sc1
//   fine       ONE launch; a workgroup waits for the flags of ITS two producers only (a dependency DAG instead of levels)
//   fence      as `level`, with plain loads / stores and agent-scope release / acquire fences instead of sc1 accesses
//   item       as `level` (a workgroup waits for ALL of phase p-1), but nobody polls the counter: every finishing workgroup adds to it,
//              the one whose add returns G - 1 (the last arriver) stores the epoch into REPL replica flags on lines of their own, and
//              a waiting workgroup polls ONE replica (w % REPL): the completion mechanism a dependency DAG of ITEMS would use
//              (per-item arrival counter + replicated done flags; here with the coarsest possible dependencies)
//   inv        as `fine`, write-through (sc0 sc1) stores but PLAIN loads behind an agent-scope acquire (buffer_inv sc1: this CU's L1)
//   uc         as `fine` on UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached)): plain stores, plain loads behind
//              the acquire — the form that would need no change to the operator bodies at all
//   uc_sc1     uncached memory, plain stores, sc1 loads, no acquire
//   uc_plain   uncached memory, plain stores, PLAIN loads and NO acquire: correct only if this memory type is not held in the CU's L1 either
//              (the pre-read plants the lines: a stale hit shows as wrong bits)
//   flags      coarse dependencies like `item` (wait for ALL of phase p-1) without any atomic: every finishing workgroup stores ITS flag,
//              a waiting workgroup's wave 0 sweeps all G flags of the phase with 64-lane sc1 loads until every one carries the epoch
// Every one-launch form PRE-READS its input tile with plain loads before it waits (values discarded): the tile then still holds the
// poison of the memset, so this CU's L1 and this XCD's L2 hold stale lines of exactly the bytes the hand-off must deliver — a form that
// returns them shows up in `wrong bits`.
// `--skew S`: every 8th workgroup does S x the arithmetic (a level is as slow as its slowest item; only `fine` can run ahead of it).
// Forward progress of the one-launch forms relies on in-order dispatch of workgroup ids (a workgroup only waits for lower ids); every
// spin is bounded by a wall-clock budget and raises a flag instead of hanging.  All forms must produce the same bits.
//   hipcc -O3 --offload-arch=gfx950 -o seam_probe seam_probe.hip && ./seam_probe [--skew 3] [--phases 24] [--groups 512] [--modes 0,2,4,5,6,7]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { M_LAUNCHES = 0, M_LEVEL = 1, M_FINE = 2, M_FENCE = 3, M_ITEM = 4, M_INV = 5, M_UC = 6, M_UC_SC1 = 7, M_FLAGS = 8, M_UC_PLAIN = 9, M_COUNT = 10 };
#define REPL 32  // replica flags per phase in the `item` form, one 128-byte line each
#define TM 32
#define TK 128
#define TN 64
#define SPIN_BUDGET_TICKS 2000000ull  // 20 ms of the 100 MHz wall clock

struct Args {
  float* X;            // [P + 1][rows][TK]
  const float* W;      // [TK][TN]
  unsigned* cnt;       // [P + 1] per-phase arrival counters
  unsigned* flag;      // [P + 1][G] per-workgroup epochs
  unsigned* rep;       // [P + 1][REPL][32] replica flags of the `item` form
  unsigned* err;       // spin budget exceeded
  float* sink;         // pre-read results land here (never true)
  int P, G, rows, skew;
  unsigned epoch;      // run number + 1
  int phase0;          // M_LAUNCHES: the phase this launch runs
};

__device__ __forceinline__ int row_block(int p, int g, int nrb) { return (g * 37 + p * 11) % nrb; }

template <int MODE>
__global__ __launch_bounds__(256, 3) void seam_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float sx[TM * (TK + 4)];
  __shared__ __attribute__((aligned(16))) float sw[TK * TN];
  constexpr bool SC1_LD = MODE == M_LEVEL || MODE == M_FINE || MODE == M_ITEM || MODE == M_UC_SC1 || MODE == M_FLAGS;
  constexpr bool SC1_ST = MODE == M_LEVEL || MODE == M_FINE || MODE == M_ITEM || MODE == M_INV || MODE == M_FLAGS;
  constexpr bool ACQ = MODE == M_FENCE || MODE == M_INV || MODE == M_UC;
  constexpr bool PER_WG = MODE == M_FINE || MODE == M_INV || MODE == M_UC || MODE == M_UC_SC1 || MODE == M_UC_PLAIN;  // per-producer flags
  const int tid = threadIdx.x;
  const int G = a.G, nrb = G / 2;
  int p, w;
  if (MODE == M_LAUNCHES) {
    p = a.phase0;
    w = blockIdx.x;
  } else {
    p = 1 + blockIdx.x / G;
    w = blockIdx.x % G;
  }
  // output piece of this workgroup: row block w / 2, column half w % 2; input: a permuted row block of the previous phase, both halves
  const int orb = w >> 1, oh = w & 1;
  const int irb = row_block(p, orb, nrb);
  const float* xin = a.X + (size_t)(p - 1) * a.rows * TK + (size_t)irb * TM * TK;
  float* xout = a.X + (size_t)p * a.rows * TK + (size_t)orb * TM * TK + oh * TN;
  // the weight panel does not depend on anybody: stage it before waiting
  for (int i = tid; i < TK * TN / 4; i += 256) reinterpret_cast<f32x4*>(sw)[i] = reinterpret_cast<const f32x4*>(a.W)[i];
  if (MODE != M_LAUNCHES && p > 1) {
    {  // pre-read: plant (possibly stale) lines of the input tile in this CU's L1 and this XCD's L2
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM * TK / 4 / 256; ++i) s += reinterpret_cast<const f32x4*>(xin)[tid + 256 * i];
      if (s[0] == 123456.f && s[1] == 654321.f) a.sink[tid] = s[2] + s[3];
    }
    if (MODE == M_FLAGS && tid < 64) {  // wave 0 sweeps the G flags of phase p - 1, 64 at a time
      const unsigned long long t0 = wall_clock64();
      const unsigned* fl = a.flag + (size_t)(p - 1) * G;
      for (;;) {
        bool ok = true;
        for (int i = tid; i < G; i += 64) ok = ok && __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { if (tid == 0) atomicExch(a.err, 1u); break; }
      }
    }
    if (tid == 0 && MODE != M_FLAGS) {
      const unsigned long long t0 = wall_clock64();
      if (PER_WG) {
        const unsigned* f0 = a.flag + (size_t)(p - 1) * G + 2 * irb;
        while (__hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch ||
               __hip_atomic_load(f0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { atomicExch(a.err, 1u); break; }
        }
      } else if (MODE == M_ITEM) {
        const unsigned* f0 = a.rep + ((size_t)(p - 1) * REPL + (w % REPL)) * 32;
        while (__hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { atomicExch(a.err, 1u); break; }
        }
      } else {
        const unsigned want = a.epoch * (unsigned)G;
        while (__hip_atomic_load(a.cnt + (p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { atomicExch(a.err, 1u); break; }
        }
      }
      if (ACQ) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
  }
  // stage the input tile: 32 x 128 floats, 16-byte loads (sc1 in the sc1 forms: every load of handed-off bytes)
  {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, TM * TK * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM * TK / 4 / 256; ++i) {
      const int e = tid + 256 * i, row = e / (TK / 4), c4 = e % (TK / 4);
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 16 * e, 0, SC1_LD ? 16 : 0));
      *reinterpret_cast<f32x4*>(&sx[row * (TK + 4) + 4 * c4]) = v;
    }
  }
  __syncthreads();
  // 32 x 64 outputs, 8 per thread (row tid / 8, columns 8 (tid % 8) ..)
  const int r = tid >> 3, c = (tid & 7) * 8;
  float acc[8];
  const int reps = (a.skew > 1 && (w & 7) == 0) ? a.skew : 1;
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 8
    for (int k = 0; k < TK; ++k) {
      const float x = sx[r * (TK + 4) + k];
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(&sw[k * TN + c]), w1 = *reinterpret_cast<const f32x4*>(&sw[k * TN + c + 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = fmaf(x, w0[j], acc[j]);
        acc[4 + j] = fmaf(x, w1[j], acc[4 + j]);
      }
    }
    if (rep + 1 < reps) asm volatile("" ::"v"(acc[0]), "v"(acc[7]));
  }
  {
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(xout, 0, (TM * TK - oh * TN) * 4, 0x00020000);
    f32x4 v0, v1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v0[j] = tanhf(acc[j]);  // (bounded values through 24 phases)
      v1[j] = tanhf(acc[4 + j]);
    }
    constexpr int aux = SC1_ST ? 17 : 0;  // sc0 sc1: write-through
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v0), ro, 4 * (r * TK + c), 0, aux);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v1), ro, 4 * (r * TK + c + 4), 0, aux);
  }
  if (MODE != M_LAUNCHES) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE == M_FENCE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (PER_WG || MODE == M_FLAGS)
        __hip_atomic_store(a.flag + (size_t)p * G + w, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE != M_ITEM)
        __hip_atomic_fetch_add(a.cnt + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == M_ITEM && tid < 64) {  // (wave 0: lane 0 arrives; if it was the last one, lanes 0 .. REPL-1 publish the replicas with one store instruction)
      unsigned old = 0;
      if (tid == 0) old = __hip_atomic_fetch_add(a.cnt + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old == a.epoch * (unsigned)G - 1u && tid < REPL)
        __hip_atomic_store(a.rep + ((size_t)p * REPL + tid) * 32, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

template <int MODE>
static float run_once(Args a, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  if (MODE == M_LAUNCHES) {
    for (int p = 1; p <= a.P; ++p) {
      a.phase0 = p;
      hipLaunchKernelGGL(seam_kernel<MODE>, dim3(a.G), dim3(256), 0, st, a);
    }
  } else {
    hipLaunchKernelGGL(seam_kernel<MODE>, dim3(a.G * a.P), dim3(256), 0, st, a);
  }
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms * 1e3f;
}

int main(int argc, char** argv) {
  int P = 24, G = 512, skew = 1, reps = 200;
  bool on[M_COUNT];
  for (int m = 0; m < M_COUNT; ++m) on[m] = true;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--skew")) skew = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--phases")) P = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--groups")) G = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--modes")) {
      for (int m = 0; m < M_COUNT; ++m) on[m] = false;
      for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) on[atoi(t) % M_COUNT] = true;
      on[M_LAUNCHES] = true;  // (the reference bits)
    }
  }
  const int rows = G / 2 * TM;
  const size_t xn = (size_t)(P + 1) * rows * TK;
  Args a{};
  a.P = P, a.G = G, a.rows = rows, a.skew = skew;
  float *Xc = nullptr, *Xu = nullptr;
  CK(hipMalloc(&Xc, xn * 4));
  if (hipExtMallocWithFlags((void**)&Xu, xn * 4, hipDeviceMallocUncached) != hipSuccess) {
    printf("hipExtMallocWithFlags(hipDeviceMallocUncached) is refused on this box: the uc forms are skipped\n");
    Xu = nullptr;
    (void)hipGetLastError();
    on[M_UC] = on[M_UC_SC1] = on[M_UC_PLAIN] = false;
  }
  float* W;
  CK(hipMalloc(&W, TK * TN * 4));
  CK(hipMalloc(&a.cnt, (P + 1) * 4));
  CK(hipMalloc(&a.flag, (size_t)(P + 1) * G * 4));
  CK(hipMalloc(&a.err, 4));
  CK(hipMalloc(&a.sink, 256 * 4));
  CK(hipMalloc(&a.rep, (size_t)(P + 1) * REPL * 32 * 4));
  CK(hipMemset(a.rep, 0, (size_t)(P + 1) * REPL * 32 * 4));
  CK(hipMemset(a.cnt, 0, (P + 1) * 4));
  CK(hipMemset(a.flag, 0, (size_t)(P + 1) * G * 4));
  CK(hipMemset(a.err, 0, 4));
  std::vector<float> h0((size_t)rows * TK), hw(TK * TN);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : h0) v = rnd();
  for (auto& v : hw) v = rnd() * 0.3f;
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  a.W = W;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<float> ref, got((size_t)rows * TK);
  const char* names[M_COUNT] = {"launches", "level", "fine", "fence", "item", "inv", "uc", "uc_sc1", "flags", "uc_plain"};
  printf("seam probe: %d phases x %d workgroups of 256 threads, skew %d, %d timed runs each (one-launch forms pre-read their inputs)\n", P, G, skew, reps);
  unsigned epoch = 0;
  for (int mode = 0; mode < M_COUNT; ++mode) {
    if (!on[mode]) continue;
    a.X = (mode == M_UC || mode == M_UC_SC1 || mode == M_UC_PLAIN) ? Xu : Xc;
    // poison everything but phase 0, so that a stale or early read shows
    CK(hipMemset(a.X, 0xff, xn * 4));
    CK(hipMemcpy(a.X, h0.data(), h0.size() * 4, hipMemcpyHostToDevice));
    double sum = 0, best = 1e30;
    int bad_runs = 0;
    for (int r = 0; r < reps + 5; ++r) {
      a.epoch = ++epoch;
      if (mode == M_LEVEL || mode == M_FENCE || mode == M_ITEM) {  // the counters are per form: reset them so that epoch * G is the target
        CK(hipMemsetAsync(a.cnt, 0, (P + 1) * 4, st));
        if (mode == M_ITEM) CK(hipMemsetAsync(a.rep, 0, (size_t)(P + 1) * REPL * 32 * 4, st));
        a.epoch = 1;
      }
      CK(hipMemsetAsync(a.X + (size_t)rows * TK, 0xff, (xn - (size_t)rows * TK) * 4, st));  // poison every phase's output before every run: a stale or early read shows as NaN bits
      float us = 0;
      switch (mode) {
        case M_LAUNCHES: us = run_once<M_LAUNCHES>(a, st); break;
        case M_LEVEL: us = run_once<M_LEVEL>(a, st); break;
        case M_FINE: us = run_once<M_FINE>(a, st); break;
        case M_ITEM: us = run_once<M_ITEM>(a, st); break;
        case M_INV: us = run_once<M_INV>(a, st); break;
        case M_UC: us = run_once<M_UC>(a, st); break;
        case M_UC_SC1: us = run_once<M_UC_SC1>(a, st); break;
        case M_FLAGS: us = run_once<M_FLAGS>(a, st); break;
        case M_UC_PLAIN: us = run_once<M_UC_PLAIN>(a, st); break;
        default: us = run_once<M_FENCE>(a, st); break;
      }
      if (r >= 5) {
        sum += us;
        best = us < best ? us : best;
      }
      // every run's result is checked against the launches form (the hand-off may fail rarely, under load)
      CK(hipMemcpy(got.data(), a.X + (size_t)P * rows * TK, got.size() * 4, hipMemcpyDeviceToHost));
      if (mode == M_LAUNCHES && r == 0) ref = got;
      if (memcmp(ref.data(), got.data(), got.size() * 4) != 0) ++bad_runs;
    }
    unsigned err = 0;
    CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
    printf("%-9s %8.1f us per run  (best %8.1f)  = %6.2f us per phase;  runs with wrong bits: %d / %d;  spin budget exceeded: %u\n", names[mode],
           sum / reps, best, sum / reps / P, bad_runs, reps + 5, err);
    if (err) {
      printf("  (a spin ran out of budget: the one-launch forms need in-order dispatch of workgroup ids and co-resident producers)\n");
      CK(hipMemset(a.err, 0, 4));
    }
  }
  return 0;
}
