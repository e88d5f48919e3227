// What a dependent launch costs when its workgroups read a 4 KB descriptor blob (a) passed by value in the kernel-argument segment (a fresh copy
// in the runtime's kernarg ring per launch: cold lines) or (b) kept in a device buffer uploaded once (read again every launch: warm lines), against
// (c) a launch that reads nothing.  hipcc -O3 --offload-arch=gfx950 -o kernarg_probe kernarg_probe.hip; ./kernarg_probe [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Blob { unsigned w[1024]; };
__device__ __forceinline__ unsigned sload(const unsigned* p) {
  const unsigned* q = (const unsigned*)(const __attribute__((address_space(4))) unsigned*)(unsigned long long)p;
  return *q;
}
__global__ __launch_bounds__(256) void by_value(Blob b, unsigned* out) {
  const unsigned* base = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned s = 0;
  for (int k = 0; k < 8; ++k) s += sload(base + ((blockIdx.x * 8 + k) % 60) * 16);  // eight 64-byte lines of "its descriptor"
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void by_pointer(const unsigned* blob, unsigned* out) {
  unsigned s = 0;
  for (int k = 0; k < 8; ++k) s += sload(blob + ((blockIdx.x * 8 + k) % 60) * 16);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void nothing(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x;
}
int main(int argc, char** argv) {
  const int wg = argc > 1 ? atoi(argv[1]) : 256, reps = 2000;
  unsigned *out, *dblob;
  hipMalloc(&out, 4 * 65536);
  hipMalloc(&dblob, 27 * sizeof(Blob));  // 27 blobs, one per "launch of the step"
  hipMemset(dblob, 1, 27 * sizeof(Blob));
  Blob h;
  for (int i = 0; i < 1024; ++i) h.w[i] = i;
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // a 230 MB sweep between "steps" so that the device blobs are not simply L2 hits (the step streams about that much)
  float* big;
  hipMalloc(&big, 230u << 20);
  for (int mode = 0; mode < 3; ++mode) {
    for (int sweep = 0; sweep < 2; ++sweep) {
      float total = 0;
      for (int rep = 0; rep < 40; ++rep) {  // 40 "steps" of 27 dependent launches
        if (sweep) hipMemsetAsync(big, rep, 230u << 20, st);
        hipEventRecord(e0, st);
        for (int l = 0; l < 27; ++l) {
          if (mode == 0) hipLaunchKernelGGL(by_value, dim3(wg), dim3(256), 0, st, h, out);
          else if (mode == 1) hipLaunchKernelGGL(by_pointer, dim3(wg), dim3(256), 0, st, dblob + l * 1024, out);
          else hipLaunchKernelGGL(nothing, dim3(wg), dim3(256), 0, st, out);
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 5) total += ms;
      }
      printf("%-42s %s: %.2f us per launch\n", mode == 0 ? "4 KB blob by value (kernel-argument segment)" : mode == 1 ? "blob in a device buffer (pointer argument)" : "no descriptor read",
             sweep ? "230 MB written between the steps" : "steps back to back            ", total / 35 * 1000 / 27);
    }
  }
  return 0;
}
