// probe: batch-256 single-pass GEMM with LDS-DMA staging (whole 512-B row pieces -> XOR-swizzled LDS image, 4 buffers, 3 chunks in flight
// across raw barriers).  y[256][768] = x[256][K] w[768][K]^T, ld = K (rows aligned to 4 bytes only when K is odd).  Checks the result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#ifndef DC
#define DC 128
#endif
#ifndef NB
#define NB 4
#endif
#define IPC (DC / 64)        // DMA instructions per wave and chunk
#define RPI (256 / DC)       // LDS rows per DMA instruction (1 KB)
#define SPR (DC / 4)         // 16-byte slots per row
#define KBS (DC / 64)        // 16-deep k-steps per wave and chunk (4 slices)
#define VMW_ ((NB - 2) * IPC)
#if NB == 4 && DC == 128
#define VMW 4
#elif NB == 2
#define VMW 0
#elif NB == 3 && DC == 128
#define VMW 2
#elif NB == 8 && DC == 64
#define VMW 6
#elif NB == 4 && DC == 64
#define VMW 2
#endif
#define STR2(x) #x
#define STR(x) STR2(x)
#define CH (64 * DC)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(4 * floats), 0x00020000);
}
__global__ __launch_bounds__(1024) void kdma(const float* x, const float* w, float* y, int M, int N, int K, int ld, int tiles_m, int tiles_n, int mode) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = wave & 3, slice = wave >> 2, tm = tile >> 1, tn = tile & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int nwg = gridDim.x, lin0 = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = lin0 & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin0 >> 3);
  const int PM = 4;
  const int panel = lin / (PM * tiles_n), within = lin - panel * PM * tiles_n;
  const int bx = within / PM, by = panel * PM + (within - bx * PM);
  const int m0 = by * 32, n0 = bx * 32;
  const int nchunks = (K + DC - 1) / DC;
  const __amdgpu_buffer_rsrc_t rA = mk_rsrc(x, (long)(M - 1) * ld + K), rB = mk_rsrc(w, (long)(N - 1) * ld + K);
  // this wave's two DMA instructions per chunk: instruction q = 2 wave + i covers LDS rows 2q, 2q + 1 (rows 0..31 = x tile, 32..63 = w tile)
  int voff[IPC];
  bool isB[IPC];
#pragma unroll
  for (int i = 0; i < IPC; ++i) {
    const int q = IPC * wave + i, row = RPI * q + lane / SPR, p = lane % SPR;
    const int kg = p ^ (row & 15);
    isB[i] = row >= 32;
    const int grow = isB[i] ? min(n0 + row - 32, N - 1) : min(m0 + row, M - 1);
    voff[i] = 4 * (grow * ld + 4 * kg);
  }
  auto issue = [&](int c) {
    const int k0 = c * DC;
#pragma unroll
    for (int i = 0; i < IPC; ++i) {
      const int q = IPC * wave + i, p = lane % SPR, row = RPI * q + lane / SPR;
      const int kg = p ^ (row & 15);
      const bool in = c < nchunks && k0 + 4 * kg < K;
      const int vo = in ? voff[i] + 4 * k0 : 0x7ffffff0;  // out of range: the DMA writes zeros
      float* dst = lds + (c % NB) * CH + q * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB[i] ? rB : rA, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
    }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NB - 1; ++c) issue(c);
  // fragments of chunk t are read right after the barrier and multiplied one iteration later (two register sets): the MFMAs of
  // chunk t - 1 run while the reads of chunk t are in flight, so a wave's only idle span per chunk is the barrier itself
  f32x4 fa[2][KBS], fb[2][KBS];
  auto readfrag = [&](int t, int set) {
    const float* buf = lds + (t % NB) * CH;
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb) {
      const int g = 4 * KBS * slice + 4 * kb + fg;
      f32x4 a = *reinterpret_cast<const f32x4*>(buf + (tm * 16 + fr) * DC + 4 * (g ^ fr));
      f32x4 b = *reinterpret_cast<const f32x4*>(buf + (32 + tn * 16 + fr) * DC + 4 * (g ^ fr));
      fa[set][kb] = a;
      fb[set][kb] = b;
    }
  };
  auto multiply = [&](int t, int set) {
    const int lim = K - t * DC;
    if (lim < DC) {  // last chunk: a 16-byte piece may straddle K
#pragma unroll
      for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool in = 4 * (4 * KBS * slice + 4 * kb + fg) + e < lim;
          fa[set][kb][e] = in ? fa[set][kb][e] : 0.f;
          fb[set][kb][e] = in ? fb[set][kb][e] : 0.f;
        }
    }
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][kb][j], fb[set][kb][j], acc, 0, 0, 0);
  };
  for (int t = 0; t < nchunks + 1; t += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = t + h;
      if (c < nchunks) {
        asm volatile("s_waitcnt vmcnt(" STR(VMW) ")" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(mode & 2) && !(slice & 1)) issue(c + NB - 1);
        if (!(mode & 4)) readfrag(c, h);
      }
      if (c >= 1 && c <= nchunks && !(mode & 1)) multiply(c - 1, h ^ 1);
      // half of the waves of every SIMD issue their DMA pieces after their MFMAs: the pieces of one half go out while the other half multiplies
      if (c < nchunks && !(mode & 2) && (slice & 1)) issue(c + NB - 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red = lds;
  if (slice > 0) *reinterpret_cast<f32x4*>(&red[(((slice - 1) * 4 + tile) * 64 + lane) * 4]) = acc;
  __syncthreads();
  if (slice == 0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = acc + *reinterpret_cast<const f32x4*>(&red[((s * 4 + tile) * 64 + lane) * 4]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = m0 + tm * 16 + 4 * fg + r, j = n0 + tn * 16 + fr;
      if (i < M && j < N) y[(long)i * N + j] = acc[r];
    }
  }
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int M = 256, N = 768;
  for (int K : {1565, 780, 3136}) {
    const int ld = K;
    std::vector<float> hx((size_t)M * ld), hw((size_t)N * ld), hy((size_t)M * N);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 1000.f;
    float *x, *w, *y;
    (void)hipMalloc(&x, hx.size() * 4);
    (void)hipMalloc(&w, hw.size() * 4);
    (void)hipMalloc(&y, hy.size() * 4);
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kdma), hipFuncAttributeMaxDynamicSharedMemorySize, NB * CH * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kdma, dim3(192), dim3(1024), NB * CH * 4, 0, x, w, y, M, N, K, ld, 8, 24, mode);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kdma, dim3(192), dim3(1024), NB * CH * 4, 0, x, w, y, M, N, K, ld, 8, 24, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 4000; ++s) {
      const int i = rand() % M, j = rand() % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hx[(size_t)i * ld + k] * hw[(size_t)j * ld + k];
      worst = fmax(worst, fabs(ref - hy[(size_t)i * N + j]));
    }
    printf("K=%4d: %.2f us per launch = %.1f TFLOP/s, max |err| over 4000 samples %.2e (hip %d)\n", K, ms * 1e3 / 200, 2.0 * M * N * K / (ms * 1e-3 / 200) / 1e12, worst, (int)err);
    (void)hipFree(x); (void)hipFree(w); (void)hipFree(y);
  }
  return 0;
}
