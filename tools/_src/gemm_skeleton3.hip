// Which instruction class of the loader phase fails to hide under the co-resident waves' MFMAs?
// Same loop as gemm_skeleton2.hip MODE 0 / 3 with the loader mix as template knobs:
//   NV  VALU ops per slice (16 slices per chunk; 8 independent chains)      real kernel: ~10
//   NS  SALU ops per slice (4 chains)                                        real kernel: ~7
//   NL  global float4 loads per chunk (L2 resident)                          real kernel: 6
//   NW  ds_write_b128 per chunk                                              real kernel: 6
//   SLICED: the VALU / SALU slices sit between the MFMA pairs instead of in front of the block
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MLD = 36;
constexpr int IMG = (128 + 64) * MLD;

struct Fill { unsigned v[8]; int s[4]; };

template <int NV, int NS>
__device__ __forceinline__ void fill_slice(Fill& f, int it, int j) {
  #pragma unroll
  for (int c = 0; c < NV; ++c) f.v[c & 7] = f.v[c & 7] * 1664525u + (unsigned)(c + j);
  #pragma unroll
  for (int c = 0; c < NS; ++c) f.s[c & 3] = f.s[c & 3] * 3 + it + c;
}

template <int NV, int NS, int NL, int NW, bool SLICED>
__global__ __launch_bounds__(256) void skel3_kernel(const float* __restrict__ g, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  for (int i = tid; i < IMG; i += 256) smem[i] = (float)(i & 7);
  Fill f;
  for (int c = 0; c < 8; ++c) f.v[c] = tid * (2 * c + 1);
  for (int c = 0; c < 4; ++c) f.s[c] = blockIdx.x + c;
  const float* gp = g + (size_t)(blockIdx.x & 63) * 8192 + tid * 4;
  float4 r[6];
  #pragma unroll
  for (int k = 0; k < 6; ++k) r[k] = make_float4(1.f, 2.f, 3.f, 4.f);
  const int i_ = lane & 31, h = lane >> 5;
  const int col4 = tid & 7, r0 = tid >> 3;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    float a0[16], a1[16], b0[16];
    #pragma unroll
    for (int k = 0; k < NL; ++k) r[k] = *reinterpret_cast<const float4*>(gp + ((it + k) & 7) * 1024);
    if (!SLICED) {
      #pragma unroll
      for (int j = 0; j < 16; ++j) fill_slice<NV, NS>(f, it, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const float* ra0 = smem + (wm0 + i_) * MLD + 4 * h;
      const float* ra1 = smem + (wm0 + 32 + i_) * MLD + 4 * h;
      const float* rb0 = smem + 128 * MLD + (wn0 + i_) * MLD + 4 * h;
      #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v0 = *reinterpret_cast<const float4*>(ra0 + 8 * q);
        const float4 v1 = *reinterpret_cast<const float4*>(ra1 + 8 * q);
        const float4 w0 = *reinterpret_cast<const float4*>(rb0 + 8 * q);
        a0[4 * q] = v0.x; a0[4 * q + 1] = v0.y; a0[4 * q + 2] = v0.z; a0[4 * q + 3] = v0.w;
        a1[4 * q] = v1.x; a1[4 * q + 1] = v1.y; a1[4 * q + 2] = v1.z; a1[4 * q + 3] = v1.w;
        b0[4 * q] = w0.x; b0[4 * q + 1] = w0.y; b0[4 * q + 2] = w0.z; b0[4 * q + 3] = w0.w;
      }
    }
    #pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc1, 0, 0, 0);
      if (SLICED) { fill_slice<NV, NS>(f, it, s); __builtin_amdgcn_sched_barrier(0); }
    }
    __syncthreads();
    #pragma unroll
    for (int k = 0; k < NW; ++k) {
      float4 v = r[k]; v.x += (float)((f.v[k] ^ (unsigned)f.s[k & 3]) & 1u);
      *reinterpret_cast<float4*>(smem + (r0 + 32 * k) * MLD + 4 * col4) = v;
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q];
  unsigned fx = 0; for (int c = 0; c < 8; ++c) fx ^= f.v[c];
  if (s == 12345.678f) sink[0] = s + (float)fx + (float)(f.s[0] + f.s[1] + f.s[2] + f.s[3]) + r[0].x;
}

template <int NV, int NS, int NL, int NW, bool SLICED>
static int launch3(const float* g, float* sink, int blocks, int iters, size_t extra_lds, hipStream_t st) {
  const size_t lds = IMG * sizeof(float) + extra_lds;
  hipFuncSetAttribute(reinterpret_cast<const void*>(skel3_kernel<NV, NS, NL, NW, SLICED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((skel3_kernel<NV, NS, NL, NW, SLICED>), dim3(blocks), dim3(256), lds, st, g, sink, iters);
  return (int)hipGetLastError();
}

extern "C" int skel3_launch(int cfg, int sliced, const float* g, float* sink, int blocks, int iters, long long extra_lds,
                            hipStream_t st) {
#define CFG(id, NV, NS, NL, NW) \
  if (cfg == id) return sliced ? launch3<NV, NS, NL, NW, true>(g, sink, blocks, iters, (size_t)extra_lds, st) \
                               : launch3<NV, NS, NL, NW, false>(g, sink, blocks, iters, (size_t)extra_lds, st);
  CFG(0, 0, 0, 0, 0) CFG(1, 10, 0, 0, 0) CFG(2, 0, 7, 0, 0) CFG(3, 0, 0, 6, 6) CFG(4, 10, 7, 6, 6)
  CFG(5, 5, 0, 0, 0) CFG(6, 20, 0, 0, 0) CFG(7, 0, 14, 0, 0) CFG(8, 5, 3, 6, 6) CFG(9, 0, 0, 6, 0) CFG(10, 0, 0, 0, 6)
#undef CFG
  return -1;
}
