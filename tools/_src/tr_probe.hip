// What does ds_read_b64_tr_b16 deliver?  LDS element e (16-bit) holds the value e; every lane passes
// an address and prints the four 16-bit values it receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void tr_probe(unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr_elems;
  if (mode == 0) addr_elems = (l & 15) * 4 + (l >> 4) * 64;            // contiguous 64-element blocks per 16-lane group
  else if (mode == 1) addr_elems = ((l & 15) >> 2) * 128 + (l & 3) * 4 + (l >> 4) * 16;   // rows of 128 elements: row (m>>2), cols 4(m&3).., group g at col offset 16g
  else addr_elems = (l & 15) * 128 + (l >> 4) * 4;                     // every lane its own row
  const unsigned byte_addr = (unsigned)(size_t)(lds) + addr_elems * 2; // (LDS pointers are 32-bit offsets in the low bits)
  unsigned lo, hi;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
  lo = (unsigned)v; hi = (unsigned)(v >> 32);
  out[l * 4 + 0] = lo & 0xffff; out[l * 4 + 1] = lo >> 16; out[l * 4 + 2] = hi & 0xffff; out[l * 4 + 3] = hi >> 16;
}

int main() {
  unsigned* d; hipMalloc(&d, 64 * 4 * 4);
  std::vector<unsigned> h(256);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
