// probe: what does straight-line code on a kernel's executed path cost when every launch starts cold?  Kernel A executes NKB KB of
// s_nop laid out straight; kernel B executes the same number of s_nop in a loop (64 bytes of code).  Alternating with a different
// kernel (C) between launches shows whether the instruction cache keeps a kernel's code from one launch to the next.
#include <hip/hip_runtime.h>
#include <cstdio>
#define NOPS256 ".rept 256\n s_nop 0\n .endr\n"   // 1 KB of code
template <int NKB>
__global__ __launch_bounds__(1024) void straight(float* out) {
  if (NKB >= 1) asm volatile(NOPS256);
  if (NKB >= 2) asm volatile(NOPS256);
  if (NKB >= 4) asm volatile(NOPS256 NOPS256);
  if (NKB >= 8) asm volatile(NOPS256 NOPS256 NOPS256 NOPS256);
  if (NKB >= 16) asm volatile(NOPS256 NOPS256 NOPS256 NOPS256 NOPS256 NOPS256 NOPS256 NOPS256);
  if (threadIdx.x == 0) out[blockIdx.x] = 1.f;
}
__global__ __launch_bounds__(1024) void looped(float* out, int n) {
  for (int i = 0; i < n; ++i) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
  if (threadIdx.x == 0) out[blockIdx.x] = 1.f;
}
__global__ void other(float* out) { out[blockIdx.x * 256 + threadIdx.x] = 2.f; }
template <typename F>
float timeit(F f, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}
int main() {
  float* out;
  (void)hipMalloc(&out, 1 << 20);
  auto L = [&](int n) { return timeit([&] { hipLaunchKernelGGL(looped, dim3(192), dim3(1024), 0, 0, out, n); }, 300); };
  printf("looped nops (64 B of code): 0: %.2f us  256 (=1 KB worth): %.2f  1024: %.2f  4096: %.2f\n", L(0), L(32), L(128), L(512));
  printf("straight-line nops: 1 KB %.2f us  2 KB %.2f  4 KB %.2f  8 KB %.2f  16 KB %.2f\n",
         timeit([&] { hipLaunchKernelGGL(straight<1>, dim3(192), dim3(1024), 0, 0, out); }, 300),
         timeit([&] { hipLaunchKernelGGL(straight<2>, dim3(192), dim3(1024), 0, 0, out); }, 300),
         timeit([&] { hipLaunchKernelGGL(straight<4>, dim3(192), dim3(1024), 0, 0, out); }, 300),
         timeit([&] { hipLaunchKernelGGL(straight<8>, dim3(192), dim3(1024), 0, 0, out); }, 300),
         timeit([&] { hipLaunchKernelGGL(straight<16>, dim3(192), dim3(1024), 0, 0, out); }, 300));
  float o = timeit([&] { hipLaunchKernelGGL(other, dim3(1024), dim3(256), 0, 0, out); }, 300);
  printf("pairs (other kernel, then the kernel) minus the other kernel alone (%.2f us): straight 4 KB %.2f us  16 KB %.2f  looped 1024: %.2f\n", o,
         timeit([&] { hipLaunchKernelGGL(other, dim3(1024), dim3(256), 0, 0, out); hipLaunchKernelGGL(straight<4>, dim3(192), dim3(1024), 0, 0, out); }, 300) - o,
         timeit([&] { hipLaunchKernelGGL(other, dim3(1024), dim3(256), 0, 0, out); hipLaunchKernelGGL(straight<16>, dim3(192), dim3(1024), 0, 0, out); }, 300) - o,
         timeit([&] { hipLaunchKernelGGL(other, dim3(1024), dim3(256), 0, 0, out); hipLaunchKernelGGL(looped, dim3(192), dim3(1024), 0, 0, out, 128); }, 300) - o);
  return 0;
}
