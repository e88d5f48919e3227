// probe: do __builtin_amdgcn_kernarg_segment_ptr() and dynamic LDS work inside a __noinline__ device function?
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { int n; int pad; float* out; int vals[64]; };
extern __shared__ __attribute__((aligned(16))) float dyn[];
__device__ __forceinline__ const Args& uniform_ref(unsigned long long p) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  return *(const Args*)(const __attribute__((address_space(4))) Args*)(((unsigned long long)hi << 32) | lo);
}
__device__ __noinline__ void callee_ka(unsigned long long ap, float* out) {
  const Args& a = uniform_ref(ap);
  out[threadIdx.x] = (float)a.vals[a.n];
}
__device__ __noinline__ void callee_lds(float* out) {
  dyn[threadIdx.x] = (float)threadIdx.x * 2.f;
  __syncthreads();
  out[64 + threadIdx.x] = dyn[63 - threadIdx.x];
}
__global__ void k(const Args a, int which) {
  if (which & 1) callee_ka((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr(), a.out);
  if (which & 2) callee_lds(a.out);
}
#include <cstdlib>
int main(int argc, char** argv) {
  Args a; a.n = 5; for (int i = 0; i < 64; ++i) a.vals[i] = 100 + i;
  hipMalloc(&a.out, 128 * 4);
  int which = argc > 1 ? atoi(argv[1]) : 3;
  hipMemset(a.out, 0, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, a, which);
  float h[128]; hipError_t e = hipMemcpy(h, a.out, 512, hipMemcpyDeviceToHost);
  printf("which %d err %d ka-> %g (want 105)  lds-> %g (want 126) %g (want 0)\n", which, (int)e, h[3], h[64], h[127]);
  return 0;
}
