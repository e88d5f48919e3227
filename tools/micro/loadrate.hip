// Microbenchmark: how fast can ONE workgroup per CU stream a cold operand tile, by load width?
// Each workgroup reads `tiles` tiles of 64 rows x 64 floats (row stride ld floats) from its own region (never re-read),
// with a distance-1 register prefetch like the GEMM staging loop, and sums what it loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W, int NT>  // W floats per lane-load (1, 2, 4); NT threads
__global__ __launch_bounds__(NT) void stream_kernel(const float* __restrict__ src, float* __restrict__ out, int tiles, int ld,
                                                    long wg_stride) {
  const int tid = threadIdx.x;
  const float* base = src + (long)blockIdx.x * wg_stride;
  constexpr int PER = 64 * 64 / (NT * W);  // loads per thread and tile
  constexpr int LPR = 64 / W;              // lanes per row
  float acc = 0.f;
  float cur[PER][W], nxt[PER][W];
  auto fetch = [&](int t, float (*r)[W]) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int row = tid / LPR + (NT / LPR) * i, col = (tid % LPR) * W;
      const float* p = base + (long)row * ld + (long)t * 64 + col;
      if (W == 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        r[i][0] = v[0]; r[i][1 % W] = v[1]; r[i][2 % W] = v[2]; r[i][3 % W] = v[3];
      } else if (W == 2) {
        float2 v = *reinterpret_cast<const float2*>(p);
        r[i][0] = v.x; r[i][1 % W] = v.y;
      } else {
        r[i][0] = *p;
      }
    }
  };
  fetch(0, cur);
  for (int t = 0; t < tiles; ++t) {
    if (t + 1 < tiles) fetch(t + 1, nxt);
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int w = 0; w < W; ++w) acc += cur[i][w];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int w = 0; w < W; ++w) cur[i][w] = nxt[i][w];
  }
  out[(long)blockIdx.x * NT + tid] = acc;
}

template <int W, int NT>
void run(const char* name, const float* src, float* out, int wgs, int tiles, int ld, long wg_stride, char* flush, size_t flush_bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9, sum = 0;
  const int reps = 20;
  for (int r = 0; r < reps + 2; ++r) {
    hipMemsetAsync(flush, r, flush_bytes, 0);  // another kernel in between: L2 cold like inside the training step
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((stream_kernel<W, NT>), dim3(wgs), dim3(NT), 0, 0, src, out, tiles, ld, wg_stride);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  const double bytes = (double)wgs * tiles * 64 * 64 * 4;
  printf("%-22s wgs=%4d tiles=%3d  avg %7.2f us  best %7.2f us  -> %.2f us/tile/WG (best), %.0f GB/s\n", name, wgs, tiles,
         sum / reps * 1e3, best * 1e3, best * 1e3 / tiles, bytes / (best * 1e-3) / 1e9);
}

int main() {
  const int ld = 2048;                // floats per row
  const long wg_stride = 64L * ld;    // each WG owns 64 full rows
  const int max_wgs = 1024;
  size_t n = (size_t)max_wgs * wg_stride;
  float *src, *out;
  char* flush;
  hipMalloc(&src, n * 4);
  hipMalloc(&out, (size_t)max_wgs * 512 * 4);
  hipMalloc(&flush, 64 << 20);
  hipMemset(src, 0, n * 4);
  for (int wgs : {16, 64, 240, 512, 1024})
    for (int tiles : {1, 4, 16}) {
      run<1, 512>("dword   x512thr", src, out, wgs, tiles, ld, wg_stride, flush, 64 << 20);
      run<1, 256>("dword   x256thr", src, out, wgs, tiles, ld, wg_stride, flush, 64 << 20);
      run<2, 256>("dwordx2 x256thr", src, out, wgs, tiles, ld, wg_stride, flush, 64 << 20);
      run<4, 256>("dwordx4 x256thr", src, out, wgs, tiles, ld, wg_stride, flush, 64 << 20);
      run<4, 512>("dwordx4 x512thr", src, out, wgs, tiles, ld, wg_stride, flush, 64 << 20);
    }
  return 0;
}
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/loadrate tools/micro/loadrate.hip   (the binary is git-ignored; it travels
// to the GPU box with the snapshot)
