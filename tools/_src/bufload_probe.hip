// probe: buffer_load_dwordx4 through a hand-built resource descriptor (the clang builtin
// __builtin_amdgcn_raw_buffer_load_b128 of this ROCm lowers to a ONE-dword load + splat): in-range /
// out-of-range byte offsets
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 llvm_raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const uint64_t a = (uint64_t)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));            // (stride 0)
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__global__ void k(const float* p, unsigned bytes, float* out) {
  const i32x4 r = make_rsrc(p, bytes);
  const unsigned t = threadIdx.x;
  const unsigned off = (t & 1) ? 0x80000000u : t * 16u;
  const f32x4 v = llvm_raw_buffer_load_f32x4(r, (int)off, 0, 0);
  out[4 * t + 0] = v.x; out[4 * t + 1] = v.y; out[4 * t + 2] = v.z; out[4 * t + 3] = v.w;
}
int main() {
  float *p, *o; const int n = 1024;
  (void)hipMalloc(&p, n * 4); (void)hipMalloc(&o, 64 * 16);
  float h[n]; for (int i = 0; i < n; ++i) h[i] = (float)i;
  (void)hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, (unsigned)(n * 4), o);
  float r[256]; (void)hipMemcpy(r, o, 64 * 16, hipMemcpyDeviceToHost);
  for (int t = 0; t < 6; ++t) printf("t=%d: %g %g %g %g\n", t, r[4 * t], r[4 * t + 1], r[4 * t + 2], r[4 * t + 3]);
  return 0;
}
