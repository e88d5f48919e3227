// probe: a raw (stride 0) buffer_load_dwordx4 that straddles num_records — are the in-range dwords returned, or the whole access zeroed?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, int records_bytes, float* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, records_bytes, 0x00020000);
  const int off = 4 * threadIdx.x;  // lane l loads floats l .. l + 3
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
  // the same addresses with 16 bytes of the offset moved into the SCALAR offset: does the range check see it?
  const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off - 16, 16, 0));
  for (int e = 0; e < 4; ++e) out[64 + threadIdx.x * 4 + e] = v[e] == u[e] ? 1.f : 0.f;
}
int main() {
  float h[64], *p, *out, ho[128];
  for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
  (void)hipMalloc(&p, 256); (void)hipMalloc(&out, 512);
  (void)hipMemcpy(p, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(16), 0, 0, p, 4 * 10, out);  // 10 floats in range
  (void)hipMemcpy(ho, out, 512, hipMemcpyDeviceToHost);
  for (int l = 6; l < 12; ++l) printf("lane %2d (floats %2d..%2d, 10 in range): %g %g %g %g\n", l, l, l + 3, ho[4 * l], ho[4 * l + 1], ho[4 * l + 2], ho[4 * l + 3]);
  int same = 1; for (int l = 4; l < 12; ++l) for (int e = 0; e < 4; ++e) same &= ho[64 + 4 * l + e] == 1.f;
  printf("with 16 bytes of the offset in soffset the result is %s\n", same ? "identical (the range check includes soffset)" : "DIFFERENT (soffset is outside the range check)");
  return 0;
}
