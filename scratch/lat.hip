#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const unsigned* __restrict__ next, int n, unsigned long long* out, int sc) {
  unsigned p = threadIdx.x * 16;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) p = next[p];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = p; }
}
__global__ void spin(float* x, int iters) {  // keep the whole chip busy
  float v = x[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  x[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
  for (size_t elems : {size_t(4096), size_t(1) << 20, size_t(1) << 26}) {
    std::vector<unsigned> h(elems);
    // random cyclic permutation with stride >= 16 elements (64 B) granularity
    size_t lines = elems / 16;
    std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
    for (size_t i = 0; i < lines; ++i) h[perm[i] * 16] = perm[(i + 1) % lines] * 16;
    unsigned* d; unsigned long long* o; float* sp;
    hipMalloc(&d, elems * 4); hipMemcpy(d, h.data(), elems * 4, hipMemcpyHostToDevice);
    hipMalloc(&o, 64); hipMalloc(&sp, 1 << 26);
    for (int busy = 0; busy < 2; ++busy) {
      hipStream_t s2; hipStreamCreate(&s2);
      if (busy) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s2, sp, 2000000);
      int n = 2000;
      hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, n, o, 0);
      hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, n, o, 0);
      hipStreamSynchronize(0);
      unsigned long long r[3]; hipMemcpy(r, o, 24, hipMemcpyDeviceToHost);
      double ns = r[1] * 10.0 / n; double clk = (double)r[0] / (r[1] * 10.0);
      printf("footprint %8zu KB busy=%d: %.0f ns per dependent load, %.0f cycles, shader clock %.2f GHz\n", elems * 4 / 1024, busy, ns, (double)r[0] / n, clk);
      hipDeviceSynchronize(); hipStreamDestroy(s2);
    }
    hipFree(d); hipFree(o); hipFree(sp);
  }
  return 0;
}
