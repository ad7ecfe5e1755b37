// probe: does a global_load_dwordx4 from an address that is only 4-byte aligned return the right data?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out) {
  const unsigned t = threadIdx.x;
  unsigned byte_off = (t * 161u + (t & 3u)) * 4u;           // arbitrary dword alignment
  asm volatile("" : "+v"(byte_off));
  const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + (size_t)byte_off);
  out[4 * t + 0] = v.x; out[4 * t + 1] = v.y; out[4 * t + 2] = v.z; out[4 * t + 3] = v.w;
}
int main() {
  const int n = 64 * 161 + 16;
  float *p, *o;
  (void)hipMalloc(&p, n * 4); (void)hipMalloc(&o, 64 * 16);
  float* h = new float[n]; for (int i = 0; i < n; ++i) h[i] = (float)i;
  (void)hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o);
  float r[256]; (void)hipMemcpy(r, o, 64 * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) for (int j = 0; j < 4; ++j) bad += r[4 * t + j] != (float)(t * 161 + (t & 3) + j);
  printf("unaligned dwordx4 loads: %s (%d mismatches); t=1: %g %g %g %g\n", bad ? "WRONG" : "ok", bad, r[4], r[5], r[6], r[7]);
  return 0;
}
