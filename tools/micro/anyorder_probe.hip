// Does hipExtAnyOrderLaunch let two INDEPENDENT kernels of one stream run side by side on gfx950 (the header says the flag is "not supported
// on AMD GFX9xx boards")?  K spins ~T us on 64 workgroups (a quarter of the CUs): two launches back to back take 2T in order; T if the second
// may start before the first has ended.   hipcc -O3 --offload-arch=gfx950 -o anyorder_probe anyorder_probe.hip && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  unsigned* out;
  hipMalloc(&out, 4096);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipEventRecord(e0, st);
      for (int k = 0; k < 8; ++k) {
        const unsigned flags = (mode == 1 && (k & 1)) ? hipExtAnyOrderLaunch : (mode == 2 ? hipExtAnyOrderLaunch : 0);
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, flags, 2000ull, out);
      }
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: 8 launches of a 20 us kernel on 64 workgroups: %.1f us\n", mode == 0 ? "in order            " : mode == 1 ? "every second any-order" : "all any-order       ", best * 1e3f);
  }
  return 0;
}
