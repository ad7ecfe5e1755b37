// Micro-benchmark of the implicit-GEMM main-loop SKELETON on gfx950 (no real data): which element of
// the loop - LDS fragment reads, barriers, LDS stores, global loads, loader VALU - costs how much of
// the fp32 matrix pipe?  256-thread workgroups (4 waves, one per SIMD), wave tile 64x32 (TM=2, TN=1:
// 32 MFMAs per K chunk of 32) as the 128x64 block tile of csrc/conv.hip, LDS image 128x36 + 64x36 floats.
//   level 0: MFMAs only            1: + fragment ds_read_b128       2: + two barriers per chunk
//   level 3: + 6 ds_write_b128     4: + 6 global float4 loads       5: + ~160 VALU + ~110 SALU filler
// BLOCK_MFMA = MFMAs per barrier interval (32 = one K chunk of 32; 64 = a chunk of 64).
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MLD = 36;

template <int LEVEL, int KSTEPS>
__global__ __launch_bounds__(256) void skeleton_kernel(const float* __restrict__ g, float* __restrict__ sink, int iters,
                                                       int stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
  float* As = smem;
  float* Bs = smem + 128 * MLD * (KSTEPS / 16);
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float a0[KSTEPS], a1[KSTEPS], b0[KSTEPS];
  for (int s = 0; s < KSTEPS; ++s) { a0[s] = 1.f + lane * 1e-3f; a1[s] = 0.5f; b0[s] = 0.25f + s; }
  // init LDS
  for (int i = tid; i < (128 + 64) * MLD * (KSTEPS / 16); i += 256) smem[i] = (float)(i & 7);
  __syncthreads();
  float4 r[6];
  const float* gp = g + (size_t)(blockIdx.x * 256 + tid) * 4;
  unsigned filler = tid, fill1 = tid * 3u, fill2 = tid * 5u, fill3 = tid * 7u, sfill = blockIdx.x;
  const int i_ = lane & 31, h = lane >> 5;
  for (int it = 0; it < iters; ++it) {
    if (LEVEL >= 4) {
      #pragma unroll
      for (int k = 0; k < 6; ++k) r[k] = *reinterpret_cast<const float4*>(gp + (size_t)((it * 6 + k) % stride) * (1024 * 256 * 4));
    }
    if (LEVEL >= 5) {
      #pragma unroll
      for (int k = 0; k < 40; ++k) {          // four independent chains (the loader's address / affine math is parallel)
        filler = filler * 1664525u + (unsigned)k; fill1 = fill1 * 22695477u + 1u; fill2 = (fill2 ^ filler) + 7u; fill3 = fill3 * 3u + fill1;
      }
      #pragma unroll
      for (int k = 0; k < 110; ++k) sfill = __builtin_amdgcn_readfirstlane(sfill) * 3u + (unsigned)it;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (LEVEL >= 1) {
      #pragma unroll
      for (int c = 0; c < KSTEPS / 16; ++c) {
        const float* ra0 = As + c * 128 * MLD + (wm0 + i_) * MLD + 4 * h;
        const float* ra1 = As + c * 128 * MLD + (wm0 + 32 + i_) * MLD + 4 * h;
        const float* rb0 = Bs + c * 64 * MLD + (wn0 + i_) * MLD + 4 * h;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v0 = *reinterpret_cast<const float4*>(ra0 + 8 * q);
          const float4 v1 = *reinterpret_cast<const float4*>(ra1 + 8 * q);
          const float4 w0 = *reinterpret_cast<const float4*>(rb0 + 8 * q);
          a0[c * 16 + 4 * q] = v0.x; a0[c * 16 + 4 * q + 1] = v0.y; a0[c * 16 + 4 * q + 2] = v0.z; a0[c * 16 + 4 * q + 3] = v0.w;
          a1[c * 16 + 4 * q] = v1.x; a1[c * 16 + 4 * q + 1] = v1.y; a1[c * 16 + 4 * q + 2] = v1.z; a1[c * 16 + 4 * q + 3] = v1.w;
          b0[c * 16 + 4 * q] = w0.x; b0[c * 16 + 4 * q + 1] = w0.y; b0[c * 16 + 4 * q + 2] = w0.z; b0[c * 16 + 4 * q + 3] = w0.w;
        }
      }
    }
    #pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc1, 0, 0, 0);
    }
    if (LEVEL >= 2) __syncthreads();
    if (LEVEL >= 3) {
      const int col4 = tid & 7, r0 = tid >> 3;
      #pragma unroll
      for (int k = 0; k < 6 * (KSTEPS / 16); ++k) {
        float4 v = LEVEL >= 4 ? r[k % 6] : make_float4((float)it, 1.f, 2.f, 3.f);
        if (LEVEL >= 5) v.x += (float)((filler ^ fill2 ^ fill3) & 1u) + (float)(sfill & 1u);
        *reinterpret_cast<float4*>(smem + (r0 + 32 * k) * MLD + 4 * col4) = v;
      }
    }
    if (LEVEL >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q];
  if (s == 12345.678f) sink[0] = s + (float)filler + (float)sfill;
}

template <int LEVEL, int KSTEPS>
static int launch(const float* g, float* sink, int blocks, int iters, int stride, size_t extra_lds, hipStream_t st) {
  const size_t lds = (128 + 64) * MLD * (KSTEPS / 16) * sizeof(float) + extra_lds;
  hipFuncSetAttribute(reinterpret_cast<const void*>(skeleton_kernel<LEVEL, KSTEPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((skeleton_kernel<LEVEL, KSTEPS>), dim3(blocks), dim3(256), lds, st, g, sink, iters, stride);
  return (int)hipGetLastError();
}

// ksteps: 16 (BK 32: 32 MFMAs per barrier interval) or 32 (BK 64).  g: >= stride * 1024 * 256 * 4 floats.
extern "C" int skeleton_launch(int level, int ksteps, const float* g, float* sink, int blocks, int iters, int stride,
                               long long extra_lds, hipStream_t st) {
#define CASE(L, K) if (level == L && ksteps == K) return launch<L, K>(g, sink, blocks, iters, stride, (size_t)extra_lds, st);
  CASE(0, 16) CASE(1, 16) CASE(2, 16) CASE(3, 16) CASE(4, 16) CASE(5, 16)
  CASE(0, 32) CASE(1, 32) CASE(2, 32) CASE(3, 32) CASE(4, 32) CASE(5, 32)
#undef CASE
  return -1;
}
