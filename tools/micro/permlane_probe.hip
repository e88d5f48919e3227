// what v_permlane16_swap / v_permlane32_swap do on gfx950 (input: the lane id in both operands) — hipcc -O3 --offload-arch=gfx950 -o permlane_probe permlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned l = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
  o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* n[4] = {"permlane16_swap(vdst = l, src = l + 100) -> vdst", "                                         -> src ", "permlane32_swap(vdst = l, src = l + 100) -> vdst", "                                         -> src "};
  for (int q = 0; q < 4; ++q) { printf("%s:", n[q]); for (int l = 0; l < 64; l += 1) printf(" %u", h[64 * q + l]); printf("\n"); }
  return 0;
}
