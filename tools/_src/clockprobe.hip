// Shader-clock probe: one wave spins for a given wall time and reports how many shader cycles
// (s_memtime) went by per 100 MHz reference tick (s_memrealtime) - i.e. the engine clock the
// chip sustains WHILE another kernel (launched on a different stream) is running.  Plus a pure
// fp32-MFMA burn kernel (no memory traffic) as the "matrix pipe only" load.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/_src/clockprobe.hip -o tools/_bin/libclockprobe.so
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void clock_probe_kernel(long long* out, long long min_ticks) {
  if (threadIdx.x != 0) return;
  const long long w0 = wall_clock64(), c0 = clock64();
  long long w1 = w0;
  while (w1 - w0 < min_ticks) { __builtin_amdgcn_s_sleep(8); w1 = wall_clock64(); }
  const long long c1 = clock64();
  out[0] = c1 - c0; out[1] = w1 - w0;
}

__global__ __launch_bounds__(256) void mfma_burn_kernel(float* sink, int iters) {
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  for (int i = 0; i < iters; ++i) {
    #pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  if (s == 12345.678f) sink[0] = s;
}

extern "C" int clockprobe_launch(long long* out, long long min_ticks, hipStream_t st) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, out, min_ticks);
  return (int)hipGetLastError();
}

// blocks x 256 threads, each wave issues iters x 32 MFMAs (4096 FLOP x 64 lanes... = 32x32x2x2 per MFMA)
extern "C" int mfma_burn_launch(int blocks, int iters, float* sink, hipStream_t st) {
  hipLaunchKernelGGL(mfma_burn_kernel, dim3(blocks), dim3(256), 0, st, sink, iters);
  return (int)hipGetLastError();
}
