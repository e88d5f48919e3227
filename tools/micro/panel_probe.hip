// probe: what bounds a 1024-thread workgroup that consumes a (32 + 32)-row x K fp32 operand panel (the batch-256 single-pass GEMM of
// csrc/gemm_kslice.hip)?  Variants: fragment-shaped loads (16 rows x 64 B per wave-instruction) with / without the MFMAs, whole-line
// loads (1 KB of one row per wave-instruction) without MFMAs, MFMAs alone.  x[256][K], w[768][K]; 192 workgroups in the XCD-compact order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KD 4
template <int MODE>
__global__ __launch_bounds__(1024) void probe(const float* x, const float* w, float* out, int K, int tiles_m, int tiles_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nwg = gridDim.x, lin0 = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = lin0 & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin0 >> 3);
  const int PM = 4;
  const int panel = lin / (PM * tiles_n), within = lin - panel * PM * tiles_n;
  const int bx = within / PM, by = panel * PM + (within - bx * PM);
  const int m0 = by * 32, n0 = bx * 32;
  f32x4 acc[2][2] = {};
  float s = 0.f;
  if (MODE == 0 || MODE == 1) {
    const int steps = K / 16;
    f32x4 fa[KD][2], fb[KD][2];
    int g = wave;
    auto fetch = [&](int slot) {
      const bool live = g < steps;
      const int kk = live ? 16 * g + 4 * fg : 0;
      const float* pa = x + (long)(m0 + fr) * K + kk;
      const float* pb = w + (long)(n0 + fr) * K + kk;
      if (live) {
        fa[slot][0] = *(const f32x4*)pa;
        fa[slot][1] = *(const f32x4*)(pa + 16 * K);
        fb[slot][0] = *(const f32x4*)pb;
        fb[slot][1] = *(const f32x4*)(pb + 16 * K);
      }
      g += 16;
    };
#pragma unroll
    for (int r = 0; r < KD; ++r) fetch(r);
    const int mine = steps > wave ? (steps - wave + 15) >> 4 : 0;
    for (int t = 0; t < mine; t += KD) {
#pragma unroll
      for (int r = 0; r < KD; ++r) {
        if (t + r < mine) {
          if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][0][j], fb[r][0][j], acc[0][0], 0, 0, 0);
              acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][0][j], fb[r][1][j], acc[0][1], 0, 0, 0);
              acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][1][j], fb[r][0][j], acc[1][0], 0, 0, 0);
              acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][1][j], fb[r][1][j], acc[1][1], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s += fa[r][0][j] + fa[r][1][j] + fb[r][0][j] + fb[r][1][j];
          }
        }
        fetch(r);
      }
    }
  } else if (MODE == 2) {
    // whole lines: wave w streams rows w, w + 16 of x's panel and of w's panel, 1 KB per load
    const int pieces = K / 256;
    f32x4 f[KD][4];
    int g = 0;
    auto fetch = [&](int slot) {
      if (g < pieces) {
        const int kk = 256 * g + 4 * lane;
        f[slot][0] = *(const f32x4*)(x + (long)(m0 + wave) * K + kk);
        f[slot][1] = *(const f32x4*)(x + (long)(m0 + 16 + wave) * K + kk);
        f[slot][2] = *(const f32x4*)(w + (long)(n0 + wave) * K + kk);
        f[slot][3] = *(const f32x4*)(w + (long)(n0 + 16 + wave) * K + kk);
      }
      ++g;
    };
#pragma unroll
    for (int r = 0; r < KD; ++r) fetch(r);
    for (int t = 0; t < pieces; t += KD) {
#pragma unroll
      for (int r = 0; r < KD; ++r) {
        if (t + r < pieces) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += f[r][q][j];
        }
        fetch(r);
      }
    }
  } else {
    const int steps = K / 16;
    const int mine = steps > wave ? (steps - wave + 15) >> 4 : 0;
    float a = (float)lane, b = (float)fr;
    for (int t = 0; t < mine; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1][1], 0, 0, 0);
      }
    }
  }
  float v = s;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) v += acc[a][b][r];
  out[(long)blockIdx.x * 1024 + tid] = v;
}
template <int MODE>
float run(const float* x, const float* w, float* out, int K, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(192), dim3(1024), 0, 0, x, w, out, K, 8, 24);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(192), dim3(1024), 0, 0, x, w, out, K, 8, 24);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}
__global__ void empty_kernel(float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f; }
int main() {
  float *x, *w, *out;
  const int KMAX = 3328;
  hipMalloc(&x, 256L * KMAX * 4);
  hipMalloc(&w, 768L * KMAX * 4);
  hipMalloc(&out, 192L * 1024 * 4);
  hipMemset(x, 0, 256L * KMAX * 4);
  hipMemset(w, 0, 768L * KMAX * 4);
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(192), dim3(1024), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty 192 x 1024 launch: %.2f us\n", ms * 1e3f / 200);
  }
  for (int K : {256, 768, 1536, 3072}) {
    printf("K=%4d  fragment loads only %.2f us | fragment loads + MFMA %.2f us | whole-line loads only %.2f us | MFMA only %.2f us   (panel %d KB per workgroup)\n", K,
           run<0>(x, w, out, K, 200), run<1>(x, w, out, K, 200), run<2>(x, w, out, K, 200), run<3>(x, w, out, K, 200), 64 * K * 4 / 1024);
  }
  return 0;
}
