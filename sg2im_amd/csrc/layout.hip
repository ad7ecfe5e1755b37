// Scene layout (reference sg2im/layout.py) and object crops (reference sg2im/bilinear.py).
//
// The reference materialises, per object, img_in (O,D,M,M), a sampling grid (O,H,W,2) and
// the sampled tensor (O,D,H,W) - 426 MB at the COCO-64 batch - before a scatter_add into
// the 67 MB layout (layout.py:87-88,146-148).  Because the sampled value factorises,
//   sampled[o,d,y,x] = vecs[o,d] * S_o(y,x),   S_o = bilinear_zero_pad(mask_o)(grid_o(y,x)),
// the kernels below compute S_o on the fly in LDS and write the layout exactly once
// (HBM-bound: algorithmic bytes = N*H*W*D*4 written + O*(D+4+M*M)*4 read).  The
// factorised form differs from the reference's (v*m)*w rounding by ~1 ulp (SURVEY.md
// section 8a row 6: 4.8e-7 abs), so the layout is checked with a tolerance; the
// per-image accumulation order (objects in index order) is the reference's.
#include <algorithm>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "launch_count.h"
#include "sg2im_hip.h"

namespace sg2im {

// torch.linspace(0, 1, steps)[i] as computed by ATen on the CPU (symmetric halves)
__device__ __forceinline__ float lin01(int i, int steps) {
  if (steps == 1) return 0.f;
  const float step = 1.f / (float)(steps - 1);
  return i < steps / 2 ? (float)i * step : 1.f - (float)(steps - 1 - i) * step;
}

__device__ __forceinline__ float unnormalize(float g, int size, int align_corners) {
  // ATen grid_sampler_unnormalize
  return align_corners ? (g + 1.f) * 0.5f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) * 0.5f;
}

struct MaskRef { const float* f; const long long* i64; int M; };   // M == 0: all-ones 8x8 map

__device__ __forceinline__ float mask_at(const MaskRef& m, int o, int y, int x) {
  if (m.M == 0) return 1.f;
  const long long idx = ((long long)o * m.M + y) * m.M + x;
  return m.i64 ? (float)m.i64[idx] : m.f[idx];
}

// bilinear footprint of one output pixel in the object's M x M map
struct Foot { int x0, y0; float wx0, wx1, wy0, wy1, tx, ty; };   // weights already zeroed when out of bounds;
                                                                  // tx, ty: the raw fractions

__device__ __forceinline__ Foot footprint(const float* box, int y, int x, int H, int W, int Min,
                                          int align_corners) {
  // layout.py:94-128: X = (linspace(0,1,W) - x0) / (x1 - x0); grid = 2*X - 1
  const float bx0 = box[0], by0 = box[1], bx1 = box[2], by1 = box[3];
  const float gx = ((lin01(x, W) - bx0) / (bx1 - bx0)) * 2.f - 1.f;
  const float gy = ((lin01(y, H) - by0) / (by1 - by0)) * 2.f - 1.f;
  const float ix = unnormalize(gx, Min, align_corners), iy = unnormalize(gy, Min, align_corners);
  const float fx = floorf(ix), fy = floorf(iy);
  Foot f;
  // clamp the float before the int conversion: degenerate boxes give inf/NaN grids in the
  // reference; here such pixels simply fall outside the map
  f.x0 = (int)fminf(fmaxf(fx, -2.f), (float)Min + 1.f);
  f.y0 = (int)fminf(fmaxf(fy, -2.f), (float)Min + 1.f);
  const float tx = ix - fx, ty = iy - fy;
  f.tx = tx; f.ty = ty;
  f.wx0 = (f.x0 >= 0 && f.x0 < Min) ? 1.f - tx : 0.f;
  f.wx1 = (f.x0 + 1 >= 0 && f.x0 + 1 < Min) ? tx : 0.f;
  f.wy0 = (f.y0 >= 0 && f.y0 < Min) ? 1.f - ty : 0.f;
  f.wy1 = (f.y0 + 1 >= 0 && f.y0 + 1 < Min) ? ty : 0.f;
  if (!(ix == ix) || !(iy == iy) || fabsf(ix) > 1e8f || fabsf(iy) > 1e8f) { f.wx0 = f.wx1 = f.wy0 = f.wy1 = 0.f; f.x0 = f.y0 = -2; }
  return f;
}

__device__ __forceinline__ float sample_map(const MaskRef& m, int o, const Foot& f) {
  float s = 0.f;
  if (f.wy0 != 0.f) {
    if (f.wx0 != 0.f) s += mask_at(m, o, f.y0, f.x0) * (f.wx0 * f.wy0);
    if (f.wx1 != 0.f) s += mask_at(m, o, f.y0, f.x0 + 1) * (f.wx1 * f.wy0);
  }
  if (f.wy1 != 0.f) {
    if (f.wx0 != 0.f) s += mask_at(m, o, f.y0 + 1, f.x0) * (f.wx0 * f.wy1);
    if (f.wx1 != 0.f) s += mask_at(m, o, f.y0 + 1, f.x0 + 1) * (f.wx1 * f.wy1);
  }
  return s;
}

constexpr int LP = 32;       // pixels per workgroup
constexpr int LO = 32;       // objects per LDS pass

// grid: (ceil(H*W / LP), n_images); 256 threads = 32 pixels x 8 channel lanes
__global__ __launch_bounds__(256) void layout_fwd_kernel(const float* __restrict__ vecs, long long ld_vecs,
                                                         const float* __restrict__ boxes, MaskRef mk,
                                                         const int* __restrict__ img_row_ptr,
                                                         const int* __restrict__ img_entries, int D, int H,
                                                         int W, int align_corners, float* __restrict__ out,
                                                         long long ld_out) {
  __shared__ float S[LO][LP + 1];
  __shared__ int objs[LO];
  __shared__ int act[LO];      // does the object touch any of this workgroup's pixels?
  const int n = blockIdx.y, p0 = blockIdx.x * LP, HW = H * W;
  const int tid = threadIdx.x;
  const int p = tid >> 3, cl = tid & 7;
  const int ob = img_row_ptr[n], oe = img_row_ptr[n + 1];
  const int Min = mk.M == 0 ? 8 : mk.M;
  const bool v4 = (D % 4 == 0) && (ld_vecs % 4 == 0) && (ld_out % 4 == 0) && !(((uintptr_t)vecs | (uintptr_t)out) & 15);
  const int pix = p0 + p;
  float* orow = out + ((long long)n * HW + pix) * ld_out;
  const int nq = v4 ? D / 4 : D;              // work items per pixel (quads or scalars)
  for (int q0 = 0; q0 < nq; q0 += 8 * 4) {    // 4 register accumulators per thread per pass
    float4 acc[4];
    #pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cb = ob; cb < oe; cb += LO) {
      const int nobj = min(LO, oe - cb);
      __syncthreads();
      if (tid < nobj) objs[tid] = img_entries[cb + tid];
      __syncthreads();
      for (int e = tid; e < nobj * LP; e += 256) {
        const int oi = e / LP, pp = e - oi * LP;
        const int px = p0 + pp;
        float s = 0.f;
        if (px < HW) {
          const int o = objs[oi];
          const Foot f = footprint(boxes + 4LL * o, px / W, px % W, H, W, Min, align_corners);
          s = sample_map(mk, o, f);
        }
        S[oi][pp] = s;
        // (an object's LP = 32 pixels are one half of a wavefront: most boxes miss most pixel tiles, and
        // adding s = +0 products is a no-op - skip those objects.  Bit-identical for FINITE vectors; the one
        // deviation from the reference: 0 * NaN / 0 * Inf of a skipped object's vector does not reach these
        // pixels - a non-finite obj_vecs row still poisons the pixels its box does cover, box_net's and
        // mask_net's outputs and hence the total loss, so the NaN guard of train.py:553-555 fires all the same)
        const unsigned long long hit = __ballot(s != 0.f);
        if ((tid & 31) == 0) act[oi] = ((tid & 32) ? (unsigned)(hit >> 32) : (unsigned)hit) != 0u;
      }
      __syncthreads();
      if (pix < HW) {
        for (int oi = 0; oi < nobj; ++oi) {
          if (!act[oi]) continue;
          const float s = S[oi][p];
          const float* vrow = vecs + (long long)objs[oi] * ld_vecs;
          #pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int q = q0 + cl + 8 * k;
            if (q < nq) {
              if (v4) {
                const float4 v = *reinterpret_cast<const float4*>(vrow + 4 * q);
                acc[k].x += v.x * s; acc[k].y += v.y * s; acc[k].z += v.z * s; acc[k].w += v.w * s;
              } else {
                acc[k].x += vrow[q] * s;
              }
            }
          }
        }
      }
    }
    if (pix < HW) {
      #pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = q0 + cl + 8 * k;
        if (q < nq) {
          if (v4) *reinterpret_cast<float4*>(orow + 4 * q) = acc[k];
          else orow[q] = acc[k].x;
        }
      }
    }
  }
}

// ---- layout + noise channels + the refinement network's average-pool pyramid in ONE pass -------------------
// (crn.py:58-62 pools the full-resolution layout once per module; model.py:164-169 appends the noise channels.)
// A workgroup owns a 16 x 16 pixel tile of one image and a slab of 32 channels - layout channels, computed as in
// layout_fwd_kernel (objects in index order, one thread per pixel), or noise channels, read from the NCHW noise
// tensor - writes the full-resolution NHWC tile and reduces it through LDS to up to four coarser levels, each the
// 2 x 2 mean of the next finer one in avgpool_v4_kernel's summation order (bit-identical to the chain of
// sg2im_avgpool_forward launches it replaces, which re-read the 84 MB tensor).
constexpr int PT = 16;                 // tile edge
constexpr int PS = 32;                 // channels per slab
struct PyrLevels { float* p[5]; };     // [0] = full resolution; [l] = (H >> l) x (W >> l)
// grid.z = 2: z = 0 walks ALL layout slabs of its tile, z = 1 the noise slabs.  The per-pixel, per-object mask sample
// does not depend on the channel: the z = 0 workgroup evaluates it once (first slab) and keeps it in LDS for the
// other slabs (round 3: one workgroup per slab, i.e. the sampling - ~3x the arithmetic of a slab's multiply-adds - was
// repeated D / 32 times).  Same operations per element in the same order: bit-identical.
__global__ __launch_bounds__(256) void layout_pyramid_kernel(const float* __restrict__ vecs, long long ld_vecs,
                                                             const float* __restrict__ boxes, MaskRef mk,
                                                             const int* __restrict__ img_row_ptr,
                                                             const int* __restrict__ img_entries, int D,
                                                             const float* __restrict__ noise, int ND, int H, int W,
                                                             int align_corners, int n_levels, PyrLevels out, int ldc) {
  __shared__ float T[PT * PT][PS + 1];
  __shared__ float L1[64][PS + 1];
  __shared__ float vs[LO][PS];
  __shared__ float SV[LO][PT * PT];              // the samples of the (single) object pass, [object][pixel]
  __shared__ int objs[LO];
  __shared__ int vis[LO];                        // does the object's map reach any pixel of this tile?
  const int tiles_x = W / PT;
  const int ty0 = (blockIdx.x / tiles_x) * PT, tx0 = (blockIdx.x % tiles_x) * PT;
  const int n = blockIdx.y;
  const bool noise_group = blockIdx.z == 1;
  const int tid = threadIdx.x;
  const int y = ty0 + tid / PT, x = tx0 + tid % PT;
  const int nl = D / PS;                         // layout slabs; the rest are noise slabs
  const int slab_lo = noise_group ? nl : 0, slab_hi = noise_group ? nl + ND / PS : nl;
  const int ob = img_row_ptr[n], oe = img_row_ptr[n + 1];
  const int Min = mk.M == 0 ? 8 : mk.M;
  const bool one_pass = oe - ob <= LO;
  for (int slab = slab_lo; slab < slab_hi; ++slab) {
    const int cbase = slab * PS;                 // first destination channel
    float acc[PS];
    #pragma unroll
    for (int k = 0; k < PS; ++k) acc[k] = 0.f;
    if (!noise_group) {
      const bool sampled = one_pass && slab > slab_lo;      // (workgroup-uniform) SV holds this tile's samples
      for (int cb = ob; cb < oe; cb += LO) {
        const int nobj = min(LO, oe - cb);
        __syncthreads();
        if (tid < nobj) objs[tid] = img_entries[cb + tid];
        __syncthreads();
        for (int e = tid; e < nobj * PS; e += 256) {
          const int oi = e / PS, k = e - oi * PS;
          vs[oi][k] = vecs[(long long)objs[oi] * ld_vecs + cbase + k];
        }
        if (tid < nobj) {
          // Tile-level culling: the map coordinate of a pixel is monotonic along a row / column of the tile (affine in
          // linspace for any box), so when both end pixels fall off the SAME side of the map every pixel between them
          // does and every sample is exactly 0 (what the per-pixel path computes, too: bit-identical).  At 256 x 256 an
          // image has 10-30 objects and a 16 x 16 tile sees a handful of them.
          const float* bx = boxes + 4LL * objs[tid];
          const Foot a = footprint(bx, ty0, tx0, H, W, Min, align_corners);
          const Foot b = footprint(bx, ty0 + PT - 1, tx0 + PT - 1, H, W, Min, align_corners);
          const bool outx = (a.x0 < -1 && b.x0 < -1) || (a.x0 > Min - 1 && b.x0 > Min - 1);
          const bool outy = (a.y0 < -1 && b.y0 < -1) || (a.y0 > Min - 1 && b.y0 > Min - 1);
          vis[tid] = !(outx || outy);
        }
        __syncthreads();
        for (int oi = 0; oi < nobj; ++oi) {
          if (!vis[oi]) continue;                   // (workgroup-uniform)
          float sv;
          if (sampled) {
            sv = SV[oi][tid];
          } else {
            const int o = objs[oi];
            const Foot f = footprint(boxes + 4LL * o, y, x, H, W, Min, align_corners);
            sv = sample_map(mk, o, f);
            if (one_pass) SV[oi][tid] = sv;         // (read back by this thread only)
          }
          if (sv != 0.f) {                          // (a zero sample adds exactly +0: skipped, as in layout_fwd_kernel)
            #pragma unroll
            for (int k = 0; k < PS; ++k) acc[k] += vs[oi][k] * sv;
          }
        }
      }
    } else {
      const int c0 = cbase - D;
      #pragma unroll
      for (int k = 0; k < PS; ++k) acc[k] = noise[(((long long)n * ND + c0 + k) * H + y) * W + x];
    }
    // full resolution
    if (n_levels < 1) {
      float* dst = out.p[0] + (((long long)n * H + y) * W + x) * ldc + cbase;
      #pragma unroll
      for (int k = 0; k < PS; k += 4) *reinterpret_cast<float4*>(dst + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
    }
    if (n_levels >= 1) {
      __syncthreads();                             // (the previous slab's reads of T are over)
      #pragma unroll
      for (int k = 0; k < PS; ++k) T[tid][k] = acc[k];
      __syncthreads();
      // full resolution, written from the LDS tile: 8 consecutive lanes cover a pixel's 128-byte slab (a thread
      // writing its own pixel's 32 channels touches 64 lines per store instruction)
      for (int v = tid; v < PT * PT * (PS / 4); v += 256) {
        const int px = v >> 3, q = v & 7;
        const float* t = &T[px][4 * q];
        *reinterpret_cast<float4*>(out.p[0] + (((long long)n * H + ty0 + px / PT) * W + tx0 + px % PT) * ldc + cbase + 4 * q) =
          make_float4(t[0], t[1], t[2], t[3]);
      }
      // level 1: 8 x 8 pixels of the tile
      for (int v = tid; v < 64 * PS; v += 256) {
        const int px = v / PS, ch = v - px * PS;
        const int y1 = px / 8, x1 = px - y1 * 8;
        const int b = (2 * y1) * PT + 2 * x1;
        float sm = T[b][ch];
        sm += T[b + 1][ch]; sm += T[b + PT][ch]; sm += T[b + PT + 1][ch];
        const float r = sm * 0.25f;
        L1[px][ch] = r;
        out.p[1][(((long long)n * (H >> 1) + (ty0 >> 1) + y1) * (W >> 1) + (tx0 >> 1) + x1) * ldc + cbase + ch] = r;
      }
    }
    if (n_levels >= 2) {
      __syncthreads();
      // level 2: 4 x 4 (into T rows 0..15, free now), level 3: 2 x 2 (T rows 16..19), level 4: 1 (from level 3)
      for (int v = tid; v < 16 * PS; v += 256) {
        const int px = v / PS, ch = v - px * PS;
        const int y2 = px / 4, x2 = px - y2 * 4;
        const int b = (2 * y2) * 8 + 2 * x2;
        float sm = L1[b][ch];
        sm += L1[b + 1][ch]; sm += L1[b + 8][ch]; sm += L1[b + 9][ch];
        const float r = sm * 0.25f;
        T[px][ch] = r;
        out.p[2][(((long long)n * (H >> 2) + (ty0 >> 2) + y2) * (W >> 2) + (tx0 >> 2) + x2) * ldc + cbase + ch] = r;
      }
    }
    if (n_levels >= 3) {
      __syncthreads();
      if (tid < 4 * PS) {
        const int px = tid / PS, ch = tid - px * PS;
        const int y3 = px / 2, x3 = px - y3 * 2;
        const int b = (2 * y3) * 4 + 2 * x3;
        float sm = T[b][ch];
        sm += T[b + 1][ch]; sm += T[b + 4][ch]; sm += T[b + 5][ch];
        const float r = sm * 0.25f;
        T[16 + px][ch] = r;
        out.p[3][(((long long)n * (H >> 3) + (ty0 >> 3) + y3) * (W >> 3) + (tx0 >> 3) + x3) * ldc + cbase + ch] = r;
      }
    }
    if (n_levels >= 4) {
      __syncthreads();
      if (tid < PS) {
        float sm = T[16][tid];
        sm += T[17][tid]; sm += T[18][tid]; sm += T[19][tid];
        out.p[4][(((long long)n * (H >> 4) + (ty0 >> 4)) * (W >> 4) + (tx0 >> 4)) * ldc + cbase + tid] = sm * 0.25f;
      }
    }
  }
}

// ---- backward w.r.t. vecs: partial[tile][o_local][d] then reduce --------------------
constexpr int BP = 64;      // pixels per workgroup
constexpr int BO = 8;       // objects per register pass

// grid: (ceil(HW/BP), n_images).  part: [n_tiles][O][D]
__global__ __launch_bounds__(256) void layout_bwd_vecs_kernel(const float* __restrict__ dl, long long ld_dl,
                                                              const float* __restrict__ boxes, MaskRef mk,
                                                              const int* __restrict__ img_row_ptr,
                                                              const int* __restrict__ img_entries, int O, int D,
                                                              int H, int W, int align_corners,
                                                              float* __restrict__ part) {
  __shared__ float S[BO][BP + 1];
  __shared__ int objs[BO];
  extern __shared__ float red[];          // [BO][TR][D] reduction scratch
  const int n = blockIdx.y, p0 = blockIdx.x * BP, HW = H * W;
  const int tid = threadIdx.x;
  const int ob = img_row_ptr[n], oe = img_row_ptr[n + 1];
  const int Min = mk.M == 0 ? 8 : mk.M;
  // thread -> (channel c, pixel group pg): TC channel lanes
  const int TC = D < 256 ? D : 256, TR = 256 / TC;
  const int tx = tid % TC, pg = tid / TC;
  for (int cb = ob; cb < oe; cb += BO) {
    const int nobj = min(BO, oe - cb);
    __syncthreads();
    if (tid < nobj) objs[tid] = img_entries[cb + tid];
    __syncthreads();
    for (int e = tid; e < nobj * BP; e += 256) {
      const int oi = e / BP, pp = e - oi * BP;
      const int px = p0 + pp;
      float s = 0.f;
      if (px < HW) {
        const int o = objs[oi];
        const Foot f = footprint(boxes + 4LL * o, px / W, px % W, H, W, Min, align_corners);
        s = sample_map(mk, o, f);
      }
      S[oi][pp] = s;
    }
    __syncthreads();
    for (int c = tx; c < D; c += TC) {
      float acc[BO];
      #pragma unroll
      for (int k = 0; k < BO; ++k) acc[k] = 0.f;
      if (pg < TR) {
        for (int pp = pg; pp < BP; pp += TR) {
          const int px = p0 + pp;
          if (px >= HW) break;
          const float g = dl[((long long)n * HW + px) * ld_dl + c];
          #pragma unroll
          for (int k = 0; k < BO; ++k) acc[k] = fmaf(g, S[k][pp], acc[k]);
        }
      }
      if (TR > 1) {
        #pragma unroll
        for (int k = 0; k < BO; ++k) red[(k * TR + pg) * TC + tx] = acc[k];
        __syncthreads();
        if (pg == 0) {
          #pragma unroll
          for (int k = 0; k < BO; ++k) {
            float s = acc[k];
            for (int t = 1; t < TR; ++t) s += red[(k * TR + t) * TC + tx];
            acc[k] = s;
          }
        }
        __syncthreads();
      }
      if (pg == 0) {
        #pragma unroll
        for (int k = 0; k < BO; ++k)
          if (k < nobj) part[((long long)blockIdx.x * O + objs[k]) * D + c] = acc[k];
      }
    }
  }
}

// The same partial sums straight from the refinement network's PER-LEVEL layout gradients (crn.py:58-62: module i
// reads the layout average-pooled by 2^(L-1-i), so d layout = sum over levels of the level's gradient spread over
// its f x f pixels with weight 1 / f^2).  The full-resolution d layout (67 MB at the bench shape) is never written
// or re-read: sg2im_pyramid_backward + layout_bwd_vecs_kernel moved 89 + 67 + 67 MB, this pass reads the 89 MB of
// level gradients once (coarse levels are re-read by the 4^k fine pixels that share them: cache hits).  float4
// lanes along the channels, several pixels in flight per thread.
constexpr int kGradLevels = 6;         // (a six-module refinement network - BASELINE configs[3..4] - has six)
struct GradLevels { const float* p[kGradLevels]; int ld[kGradLevels]; int shift[kGradLevels]; float k[kGradLevels]; int n; };

// Round 5 form (the round-4 kernel - 256-pixel tiles, 16 float4 accumulators + 2 x 8 staged loads
// per thread = 196 registers, 512 workgroups - ran at 0.23 of 8 TB/s isolated and 172 us under the weight-gradient
// lane): many light workgroups.  A workgroup owns an 8 x 8 PIXEL TILE of one image and a slab of <= 128 channels:
//   1. S[object][pixel], the bilinear mask samples of the image's objects at the tile's pixels (<= 16 objects per pass);
//   2. per HALF tile (4 x 8 pixels): g[pixel][channel] = sum over levels of level[l][pixel >> l] / 4^l
//      (pyramid_bwd_v4_kernel's order, level 0 first) is formed once - every level's loads issued back to back - and
//      parked in LDS (16 KB);
//   3. the contraction d_vecs[o][c] += sum_pixel S[o][pixel] g[pixel][c] runs out of LDS: a thread owns one float4
//      of channels for one or two objects and walks the pixels in order (fixed order: reproducible), accumulating
//      over the two halves in registers;
//   4. one partial per (tile, object) - summed over the tiles by layout_bwd_reduce_kernel.
// ~90 registers, 20.6 KB of LDS: two workgroups fit into the 48 KB a CU has left next to the refinement network's
// background weight gradients (the kernel runs in their shadow), 2 048 workgroups at the bench shape.
constexpr int BOL = 16;     // objects per pass (an image rarely has more)
constexpr int TPX = 64;     // pixels per tile (8 x 8)
constexpr int HPX = 32;     // pixels per half tile (4 x 8): the unit that is staged in LDS
constexpr int TCH = 128;    // channels per slab

__global__ __launch_bounds__(256) void layout_bwd_vecs_levels_kernel(GradLevels lv, const float* __restrict__ boxes, MaskRef mk,
                                                                     const int* __restrict__ img_row_ptr,
                                                                     const int* __restrict__ img_entries, int O, int D,
                                                                     int H, int W, int tiles_x, int align_corners,
                                                                     float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float G[HPX][TCH];
  __shared__ float S[BOL][TPX + 1];
  __shared__ int objs[BOL];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int ty0 = (blockIdx.x / tiles_x) * 8, tx0 = (blockIdx.x % tiles_x) * 8;
  const int ob = img_row_ptr[n], oe = img_row_ptr[n + 1];
  if (ob == oe) return;                               // (workgroup-uniform: an image without objects)
  const int Min = mk.M == 0 ? 8 : mk.M;
  for (int c0 = 0; c0 < D; c0 += TCH) {
    const int DS = min(TCH, D - c0), D4 = DS >> 2;   // this slab's channels / float4 lanes (D % 4 == 0)
    const int c4 = tid % D4, trow = tid / D4;        // thread grid: float4 lane x (pixel row | object row)
    const int PR = 256 / D4;                          // pixel rows of the loader's thread grid
    const int KG = min(256 / D4, BOL);                // object rows of the contraction's thread grid (8 at 128 channels)
    for (int cb = ob; cb < oe; cb += BOL) {
      const int nobj = min(BOL, oe - cb);
      __syncthreads();                                // (the previous pass / slab is done with S, objs and G)
      if (tid < nobj) objs[tid] = img_entries[cb + tid];
      __syncthreads();
      // ---- 1. mask samples of this pass' objects at the tile's pixels ----
      for (int e = tid; e < BOL * TPX; e += 256) {
        const int oi = e / TPX, pp = e - oi * TPX;
        const int y = ty0 + (pp >> 3), x = tx0 + (pp & 7);
        float sv = 0.f;
        if (oi < nobj && y < H && x < W) {
          const int o = objs[oi];
          const Foot f = footprint(boxes + 4LL * o, y, x, H, W, Min, align_corners);
          sv = sample_map(mk, o, f);
        }
        S[oi][pp] = sv;
      }
      const int k = trow, k1 = trow + KG;             // this thread's objects (when trow < KG): k, k + KG
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
      #pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        __syncthreads();                              // (S complete; the previous half's contraction is done with G)
        // ---- 2. the summed level gradient of this half tile -> LDS ----
        if (trow < PR) {
          #pragma unroll 1
          for (int pb = trow; pb < HPX; pb += 4 * PR) {
            float4 g[4];
            #pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            #pragma unroll
            for (int l = 0; l < kGradLevels; ++l) {
              if (l < lv.n) {
                const int sh = lv.shift[l];
                const float kk = lv.k[l];
                const float* const base = lv.p[l] + c0 + 4 * c4;
                const int hl = H >> sh, wl = W >> sh, ldl = lv.ld[l];
                float4 v[4];
                #pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int pp = HPX * half + min(pb + i * PR, HPX - 1);
                  const int y = min(ty0 + (pp >> 3), H - 1), x = min(tx0 + (pp & 7), W - 1);
                  v[i] = *reinterpret_cast<const float4*>(base + ((long long)(n * hl + (y >> sh)) * wl + (x >> sh)) * ldl);
                }
                #pragma unroll
                for (int i = 0; i < 4; ++i) {          // (pyramid_bwd_v4_kernel's sum: level 0 first)
                  g[i].x += v[i].x * kk; g[i].y += v[i].y * kk; g[i].z += v[i].z * kk; g[i].w += v[i].w * kk;
                }
              }
            }
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int pl = pb + i * PR;
              if (pl < HPX) *reinterpret_cast<float4*>(&G[pl][4 * c4]) = g[i];     // (pixels outside the image: S is 0 there)
            }
          }
        }
        __syncthreads();
        // ---- 3. contraction out of LDS ----
        if (trow < KG && k < nobj) {
          const bool two = k1 < nobj;
          #pragma unroll 8
          for (int pl = 0; pl < HPX; ++pl) {
            const float4 gv = *reinterpret_cast<const float4*>(&G[pl][4 * c4]);
            const float s0 = S[k][HPX * half + pl], s1 = two ? S[k1][HPX * half + pl] : 0.f;
            a0.x = fmaf(gv.x, s0, a0.x); a0.y = fmaf(gv.y, s0, a0.y); a0.z = fmaf(gv.z, s0, a0.z); a0.w = fmaf(gv.w, s0, a0.w);
            a1.x = fmaf(gv.x, s1, a1.x); a1.y = fmaf(gv.y, s1, a1.y); a1.z = fmaf(gv.z, s1, a1.z); a1.w = fmaf(gv.w, s1, a1.w);
          }
        }
      }
      // ---- 4. the tile's partial for this thread's objects ----
      if (trow < KG && k < nobj) {
        *reinterpret_cast<float4*>(part + ((long long)blockIdx.x * O + objs[k]) * D + c0 + 4 * c4) = a0;
        if (k1 < nobj) *reinterpret_cast<float4*>(part + ((long long)blockIdx.x * O + objs[k1]) * D + c0 + 4 * c4) = a1;
      }
    }
  }
}

__global__ void layout_bwd_reduce_kernel(const float* __restrict__ part, int n_tiles, int O, int D,
                                         float* __restrict__ dvecs, long long ld_dvecs) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)O * D) return;
  float s = 0.f;
  #pragma unroll 8
  for (int t = 0; t < n_tiles; ++t) s += part[(long long)t * O * D + i];      // (independent loads, adds in tile order)
  dvecs[(i / D) * ld_dvecs + (i % D)] = s;
}

// ---- backward w.r.t. (soft) masks and boxes ------------------------------------------------
// With G_o(y,x) = <dL[n_o,:,y,x], v_o> the mask gradient is the bilinear transpose of G, and the
// box gradient follows from ix = unnormalize(2 (X - x0)/(x1 - x0) - 1):
//   dL/dx0 = sum_px G dS/dix (Min or Min-1)/2 * 2 (X - x1)/(x1 - x0)^2,   dL/dx1 = ... * -2 (X - x0)/(x1 - x0)^2
// (same in y), dS/dix being the x-difference of the map under the footprint with zero padding -
// exactly what grid_sample's backward gives the reference (layout.py:60-61,87-88,117-127).
// Two kernels, no atomics, fixed summation order:
//   layout_bwd_g_kernel    G_o for every pixel of the object's image -> workspace [O][H*W]
//   layout_bwd_masks_kernel one workgroup per object: every thread OWNS mask cells and gathers the
//                          pixels whose footprint touches them (y, x ascending); the four box
//                          partials are summed over pixels per thread, then over threads in order.
__global__ void layout_bwd_g_kernel(const float* __restrict__ dl, long long ld_dl, const float* __restrict__ vecs,
                                    long long ld_vecs, const long long* __restrict__ obj_to_img, int D, int HW,
                                    float* __restrict__ G) {
  const int o = blockIdx.y;
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= HW) return;
  const float* g = dl + (obj_to_img[o] * HW + px) * ld_dl;
  const float* v = vecs + (long long)o * ld_vecs;
  float ds = 0.f;
  for (int d = 0; d < D; ++d) ds = fmaf(g[d], v[d], ds);
  G[(long long)o * HW + px] = ds;
}

// The same G, one pass over the gradient (round 5; the VG-style steps' layout backward): the per-object kernel above
// re-reads its image's H x W x D gradient once per OBJECT, a thread walking one pixel's channels (stride-D accesses:
// 567 us of the 5.6 ms bfloat16 VG-64 step, and the weight-gradient lane's split-K finishes starved next to it).  Here a
// workgroup owns 64 pixels of ONE image, keeps their D <= 128 channels in registers - thread = (pixel, quarter q), its
// float4 columns are q, q + 4, q + 8, ...: the four lanes of a pixel read 64 contiguous bytes per load - and walks the
// image's objects, whose vectors pass through LDS: 67 MB read once instead of O / N times.  Per (object, pixel): four
// 32-term chains, then a fixed xor tree over the pixel's four lanes.
// LEVELS: the gradient is the refinement network's per-level gradients, summed on the fly in pyramid_bwd_v4_kernel's
// order (level 0 first) exactly as layout_bwd_vecs_levels_kernel forms it - the full-resolution tensor is never written.
constexpr int GOB = 16;     // objects per LDS pass
template <bool LEVELS>
__global__ __launch_bounds__(256) void layout_bwd_g_tiles_kernel(const float* __restrict__ dl, long long ld_dl, GradLevels lv,
                                                                 int H, int W,
                                                                 const float* __restrict__ vecs, long long ld_vecs,
                                                                 const int* __restrict__ img_row_ptr,
                                                                 const int* __restrict__ img_entries, int D, int HW,
                                                                 float* __restrict__ G) {
  __shared__ float4 vs[GOB][32];
  __shared__ int objs[GOB];
  const int n = blockIdx.y, tid = threadIdx.x, q = tid & 3;
  const int px = blockIdx.x * 64 + (tid >> 2);
  const bool live = px < HW;
  const int nj = D >> 2;
  float4 t[8];
  if constexpr (!LEVELS) {
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(dl + ((long long)n * HW + (live ? px : 0)) * ld_dl);
    #pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = q + 4 * k;
      t[k] = (live && j < nj) ? g4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    #pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int pl = live ? px : 0;
    const int y = pl / W, x = pl - y * W;
    #pragma unroll
    for (int l = 0; l < kGradLevels; ++l) {
      if (l < lv.n) {
        const int sh = lv.shift[l];
        const float kk = lv.k[l];
        const int hl = H >> sh, wl = W >> sh;
        const float4* __restrict__ g4 =
          reinterpret_cast<const float4*>(lv.p[l] + ((long long)(n * hl + (y >> sh)) * wl + (x >> sh)) * lv.ld[l]);
        float4 v[8];
        #pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = q + 4 * k;
          v[k] = (live && j < nj) ? g4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        #pragma unroll
        for (int k = 0; k < 8; ++k) {
          t[k].x += v[k].x * kk; t[k].y += v[k].y * kk; t[k].z += v[k].z * kk; t[k].w += v[k].w * kk;
        }
      }
    }
  }
  const int ob = img_row_ptr[n], oe = img_row_ptr[n + 1];
  for (int cb = ob; cb < oe; cb += GOB) {
    const int nobj = min(GOB, oe - cb);
    __syncthreads();
    if (tid < nobj) objs[tid] = img_entries[cb + tid];
    __syncthreads();
    for (int e = tid; e < nobj * 32; e += 256) {
      const int oi = e >> 5, j = e & 31;
      vs[oi][j] = j < nj ? *reinterpret_cast<const float4*>(vecs + (long long)objs[oi] * ld_vecs + 4 * j)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    for (int oi = 0; oi < nobj; ++oi) {
      float s = 0.f;
      #pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 v = vs[oi][q + 4 * k];
        s = fmaf(t[k].x, v.x, s); s = fmaf(t[k].y, v.y, s); s = fmaf(t[k].z, v.z, s); s = fmaf(t[k].w, v.w, s);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (q == 0 && live) G[(long long)objs[oi] * HW + px] = s;
    }
  }
}

// pixel range [lo, hi] along one axis whose footprint can touch map cell `cell`
__device__ __forceinline__ void layout_axis_range(float b0, float b1, int cell, int L, int Min, int align_corners,
                                                  int& lo, int& hi) {
  // map coordinate of pixel p: unnormalize(2 (lin01(p) - b0) / (b1 - b0) - 1) = A + B p (p in [0, L-1])
  const float A = unnormalize(2.f * (0.f - b0) / (b1 - b0) - 1.f, Min, align_corners);
  const float Z = unnormalize(2.f * (1.f - b0) / (b1 - b0) - 1.f, Min, align_corners);
  const float B = L > 1 ? (Z - A) / (float)(L - 1) : 0.f;
  lo = 0; hi = L - 1;
  if (B > 1e-6f && B == B && fabsf(A) < 1e8f) {
    const float l = floorf(((float)cell - 1.f - A) / B - 0.02f), h = ceilf(((float)cell + 1.f - A) / B + 0.02f);
    lo = (int)fminf(fmaxf(l, 0.f), (float)L);
    hi = (int)fminf(fmaxf(h, -1.f), (float)(L - 1));
  }
}

__global__ __launch_bounds__(256) void layout_bwd_masks_kernel(const float* __restrict__ G,
                                                               const float* __restrict__ boxes, MaskRef mk,
                                                               int H, int W, int align_corners,
                                                               float* __restrict__ dmasks, float* __restrict__ dboxes) {
  __shared__ float bp[4 * 256];
  const int M = mk.M, Min = M > 0 ? M : 8;
  const int o = blockIdx.x, tid = threadIdx.x, HW = H * W;
  const float* Go = G + (long long)o * HW;
  const float* box = boxes + 4LL * o;
  if (dmasks) {
    for (int cell = tid; cell < M * M; cell += 256) {
      const int ci = cell / M, cj = cell - ci * M;
      int xl, xh, yl, yh;
      layout_axis_range(box[0], box[2], cj, W, Min, align_corners, xl, xh);
      layout_axis_range(box[1], box[3], ci, H, Min, align_corners, yl, yh);
      float acc = 0.f;
      for (int y = yl; y <= yh; ++y)
        for (int x = xl; x <= xh; ++x) {
          const Foot f = footprint(box, y, x, H, W, Min, align_corners);
          const float wx = f.x0 == cj ? f.wx0 : (f.x0 + 1 == cj ? f.wx1 : 0.f);
          const float wy = f.y0 == ci ? f.wy0 : (f.y0 + 1 == ci ? f.wy1 : 0.f);
          if (wx != 0.f && wy != 0.f) acc += Go[y * W + x] * (wx * wy);
        }
      dmasks[(long long)o * M * M + cell] = acc;
    }
  }
  if (!dboxes) return;
  const float bw = box[2] - box[0], bh = box[3] - box[1];
  const float mult = align_corners ? 0.5f * (float)(Min - 1) : 0.5f * (float)Min;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  // only the pixels whose footprint can touch the map: the rectangle spanned by the ranges of its first and last cell
  // (conservative; the weights decide).  Outside it every term is exactly zero - the loop used to evaluate a footprint
  // for all H * W pixels of every object.
  int rx0, rx1, ry0, ry1, t0, t1;
  layout_axis_range(box[0], box[2], 0, W, Min, align_corners, rx0, t0);
  layout_axis_range(box[0], box[2], Min - 1, W, Min, align_corners, t1, rx1);
  layout_axis_range(box[1], box[3], 0, H, Min, align_corners, ry0, t0);
  layout_axis_range(box[1], box[3], Min - 1, H, Min, align_corners, t1, ry1);
  if (rx0 > rx1 || ry0 > ry1) { rx0 = 0; rx1 = W - 1; ry0 = 0; ry1 = H - 1; }      // (degenerate boxes: everything, as before)
  rx0 = max(rx0, 0); ry0 = max(ry0, 0); rx1 = min(rx1, W - 1); ry1 = min(ry1, H - 1);
  const int rw = rx1 - rx0 + 1, rn = rw * (ry1 - ry0 + 1);
  for (int q = tid; q < rn; q += 256) {
    const int y = ry0 + q / rw, x = rx0 + q % rw;
    const int px = y * W + x;
    const Foot f = footprint(box, y, x, H, W, Min, align_corners);
    if ((f.wx0 == 0.f && f.wx1 == 0.f) || (f.wy0 == 0.f && f.wy1 == 0.f)) continue;
    const float ds = Go[px];
    // corner values with zero padding; a corner outside the map contributes nothing (ATen's
    // grid_sampler backward skips it) while the other axis keeps its raw fraction
    const bool xa = f.x0 >= 0 && f.x0 < Min, xb = f.x0 + 1 >= 0 && f.x0 + 1 < Min;
    const bool ya = f.y0 >= 0 && f.y0 < Min, yb = f.y0 + 1 >= 0 && f.y0 + 1 < Min;
    const float fx = f.tx, fy = f.ty;
    float dix = 0.f, diy = 0.f;
    if (xa && ya) { const float m = mask_at(mk, o, f.y0, f.x0);         dix -= m * (1.f - fy); diy -= m * (1.f - fx); }
    if (xb && ya) { const float m = mask_at(mk, o, f.y0, f.x0 + 1);     dix += m * (1.f - fy); diy -= m * fx; }
    if (xa && yb) { const float m = mask_at(mk, o, f.y0 + 1, f.x0);     dix -= m * fy;         diy += m * (1.f - fx); }
    if (xb && yb) { const float m = mask_at(mk, o, f.y0 + 1, f.x0 + 1); dix += m * fy;         diy += m * fx; }
    const float X = lin01(x, W), Y = lin01(y, H);
    const float gx = ds * dix * mult, gy = ds * diy * mult;     // dL/d(grid x), dL/d(grid y)
    b0 += gx * (2.f * (X - box[2]) / (bw * bw));
    b2 += gx * (-2.f * (X - box[0]) / (bw * bw));
    b1 += gy * (2.f * (Y - box[3]) / (bh * bh));
    b3 += gy * (-2.f * (Y - box[1]) / (bh * bh));
  }
  bp[tid] = b0; bp[256 + tid] = b1; bp[512 + tid] = b2; bp[768 + tid] = b3;
  __syncthreads();
  if (tid < 4) {                           // fixed-order sum of the 256 thread partials
    float s = 0.f;
    for (int k = 0; k < 256; ++k) s += bp[tid * 256 + k];
    dboxes[4LL * o + tid] = s;
  }
}

// ---- crops ---------------------------------------------------------------------------
struct CropFoot { int x0, y0; float w00, w01, w10, w11; };

__device__ __forceinline__ CropFoot crop_foot(const float* box, int i, int j, int size, int H, int W,
                                              int align_corners) {
  // bilinear.py:123-130: bbox = 2*bbox-1; X = tensor_linspace(x0, x1, WW)
  const float x0 = 2.f * box[0] - 1.f, y0 = 2.f * box[1] - 1.f;
  const float x1 = 2.f * box[2] - 1.f, y1 = 2.f * box[3] - 1.f;
  const float wj = lin01(j, size), wi = lin01(i, size);
  // tensor_linspace: start * linspace(1,0) + end * linspace(0,1)  (bilinear.py:265-277)
  const float gx = (1.f - wj) * x0 + wj * x1;
  const float gy = (1.f - wi) * y0 + wi * y1;
  const float ix = unnormalize(gx, W, align_corners), iy = unnormalize(gy, H, align_corners);
  const float fx = floorf(ix), fy = floorf(iy);
  CropFoot f;
  f.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
  f.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
  const float tx = ix - fx, ty = iy - fy;
  const bool xa = f.x0 >= 0 && f.x0 < W, xb = f.x0 + 1 >= 0 && f.x0 + 1 < W;
  const bool ya = f.y0 >= 0 && f.y0 < H, yb = f.y0 + 1 >= 0 && f.y0 + 1 < H;
  f.w00 = (xa && ya) ? (1.f - tx) * (1.f - ty) : 0.f;
  f.w01 = (xb && ya) ? tx * (1.f - ty) : 0.f;
  f.w10 = (xa && yb) ? (1.f - tx) * ty : 0.f;
  f.w11 = (xb && yb) ? tx * ty : 0.f;
  if (!(ix == ix) || !(iy == iy)) { f.w00 = f.w01 = f.w10 = f.w11 = 0.f; f.x0 = f.y0 = -2; }
  return f;
}

__global__ void crop_fwd_kernel(const float* __restrict__ imgs, long long ld_img, int H, int W, int C,
                                const float* __restrict__ boxes, const long long* __restrict__ obj_to_img,
                                int O, int size, int align_corners, float* __restrict__ crops) {
  const long long total = (long long)O * size * size;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % size); const int i = (int)((t / size) % size); const int o = (int)(t / ((long long)size * size));
    const CropFoot f = crop_foot(boxes + 4LL * o, i, j, size, H, W, align_corners);
    const float* img = imgs + obj_to_img[o] * H * W * ld_img;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      if (f.w00 != 0.f) s += img[((long long)f.y0 * W + f.x0) * ld_img + c] * f.w00;
      if (f.w01 != 0.f) s += img[((long long)f.y0 * W + f.x0 + 1) * ld_img + c] * f.w01;
      if (f.w10 != 0.f) s += img[((long long)(f.y0 + 1) * W + f.x0) * ld_img + c] * f.w10;
      if (f.w11 != 0.f) s += img[((long long)(f.y0 + 1) * W + f.x0 + 1) * ld_img + c] * f.w11;
      crops[t * C + c] = s;
    }
  }
}

// Gather form of the crop backward: one thread per image pixel sums, over the image's objects
// (ascending) and the crop samples (i, j ascending) whose bilinear footprint touches the pixel,
// dcrop * weight - the transpose of crop_fwd_kernel with a FIXED summation order and no
// atomics (a scatter of the samples serialises on hot pixels: every sample of a small box lands
// on the same few pixels - 136 us vs 25 us at the bench shape).  The per-axis arithmetic repeats crop_foot's exactly.
struct CropAxis { int p0; float t; };

__device__ __forceinline__ CropAxis crop_axis(float b0, float b1, int k, int size, int L, int align_corners) {
  const float a = 2.f * b0 - 1.f, b = 2.f * b1 - 1.f;
  const float w = lin01(k, size);
  const float gpos = (1.f - w) * a + w * b;
  const float ip = unnormalize(gpos, L, align_corners);
  const float fp = floorf(ip);
  CropAxis r;
  r.p0 = (int)fminf(fmaxf(fp, -2.f), (float)L + 1.f);
  r.t = ip - fp;
  if (!(ip == ip)) { r.p0 = -2; r.t = 0.f; }
  return r;
}

// weight of sample k for pixel p along one axis (0 when the footprint misses p)
__device__ __forceinline__ float crop_axis_weight(const CropAxis& r, int p) {
  return r.p0 == p ? 1.f - r.t : (r.p0 + 1 == p ? r.t : 0.f);
}

// conservative sample range [lo, hi] whose footprint can touch pixel p
__device__ __forceinline__ void crop_axis_range(float b0, float b1, int p, int size, int L, int align_corners,
                                                int& lo, int& hi) {
  const float A = unnormalize(2.f * b0 - 1.f, L, align_corners);
  const float Z = unnormalize(2.f * b1 - 1.f, L, align_corners);
  const float B = size > 1 ? (Z - A) / (float)(size - 1) : 0.f;
  lo = 0; hi = size - 1;
  if (B > 1e-6f && B == B) {
    // sample k sits at A + B k up to rounding of a few ulp (crop_axis forms it as a blend of the
    // two box edges): a margin of 0.02 samples covers that without visiting dead samples
    const float l = floorf(((float)p - 1.f - A) / B - 0.02f), h = ceilf(((float)p + 1.f - A) / B + 0.02f);
    lo = (int)fminf(fmaxf(l, 0.f), (float)size);          // lo == size: empty range
    hi = (int)fminf(fmaxf(h, -1.f), (float)(size - 1));
  }
}

// Two passes, both free of atomics and with a fixed summation order:
//  1. one workgroup per OBJECT: its threads own the pixels of the object's footprint rectangle
//     and sum the crop samples (i, j ascending) that touch them - all threads of a workgroup
//     walk sample ranges of the same size (a per-pixel gather over all objects diverges badly:
//     a pixel inside a tiny box visits ~500 samples, its neighbours none) - into a per-object
//     image-sized plane of the workspace (only the rectangle is written);
//  2. one thread per image pixel adds the planes of the image's objects in ascending order.
constexpr int CB_MAXSIZE = 64;

struct CropRect { int x0, x1, y0, y1; };     // inclusive pixel rectangle touched by the crop

__device__ __forceinline__ CropRect crop_rect(const float* box, int size, int H, int W, int align_corners) {
  // sample positions are monotone in the sample index: the extreme samples bound the footprint
  const CropAxis xa = crop_axis(box[0], box[2], 0, size, W, align_corners);
  const CropAxis xb = crop_axis(box[0], box[2], size - 1, size, W, align_corners);
  const CropAxis ya = crop_axis(box[1], box[3], 0, size, H, align_corners);
  const CropAxis yb = crop_axis(box[1], box[3], size - 1, size, H, align_corners);
  CropRect r;
  r.x0 = max(0, min(xa.p0, xb.p0)); r.x1 = min(W - 1, max(xa.p0, xb.p0) + 1);
  r.y0 = max(0, min(ya.p0, yb.p0)); r.y1 = min(H - 1, max(ya.p0, yb.p0) + 1);
  return r;
}

__global__ void crop_bwd_object_kernel(const float* __restrict__ dcrops, int H, int W, int C,
                                       const float* __restrict__ boxes, int size, int align_corners,
                                       float* __restrict__ planes) {
  __shared__ int s_p0[2][CB_MAXSIZE];
  __shared__ float s_t[2][CB_MAXSIZE];
  const int o = blockIdx.x, tid = threadIdx.x;
  const float* box = boxes + 4LL * o;
  for (int k = tid; k < 2 * size; k += blockDim.x) {
    const int smp = k % size, ax = k / size;
    const CropAxis r = ax == 0 ? crop_axis(box[0], box[2], smp, size, W, align_corners)
                               : crop_axis(box[1], box[3], smp, size, H, align_corners);
    s_p0[ax][smp] = r.p0; s_t[ax][smp] = r.t;
  }
  __syncthreads();
  const CropRect R = crop_rect(box, size, H, W, align_corners);
  const int rw = R.x1 - R.x0 + 1, rh = R.y1 - R.y0 + 1;
  if (rw <= 0 || rh <= 0) return;
  const float* gobj = dcrops + (long long)o * size * size * C;
  float* plane = planes + (long long)o * H * W * C;
  // (gridDim.y workgroups share an object: one per object left the largest boxes - 4096 pixels on 256 threads -
  // as a 53 us tail on the generator's critical path)
  for (int q = blockIdx.y * blockDim.x + tid; q < rw * rh; q += blockDim.x * gridDim.y) {
    const int y = R.y0 + q / rw, x = R.x0 + q % rw;
    int jl, jh, il, ih;
    crop_axis_range(box[0], box[2], x, size, W, align_corners, jl, jh);
    crop_axis_range(box[1], box[3], y, size, H, align_corners, il, ih);
    for (int c0 = 0; c0 < C; c0 += 4) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int i = il; i <= ih; ++i) {
        const int py = s_p0[1][i];
        const float ty = s_t[1][i];
        const float wy = py == y ? 1.f - ty : (py + 1 == y ? ty : 0.f);
        if (wy == 0.f) continue;
        for (int j = jl; j <= jh; ++j) {
          const int px = s_p0[0][j];
          const float tx = s_t[0][j];
          const float wx = px == x ? 1.f - tx : (px + 1 == x ? tx : 0.f);
          if (wx == 0.f) continue;
          const float w = wx * wy;                         // crop_foot: (x factor) * (y factor)
          const float* gsrc = gobj + ((long long)i * size + j) * C + c0;
          a0 += gsrc[0] * w;
          if (c0 + 1 < C) a1 += gsrc[1] * w;
          if (c0 + 2 < C) a2 += gsrc[2] * w;
          if (c0 + 3 < C) a3 += gsrc[3] * w;
        }
      }
      float* dst = plane + ((long long)y * W + x) * C + c0;
      dst[0] = a0;
      if (c0 + 1 < C) dst[1] = a1;
      if (c0 + 2 < C) dst[2] = a2;
      if (c0 + 3 < C) dst[3] = a3;
    }
  }
}

// The same planes, SEPARABLY (round 5).  The crop is a tensor product of two 1-D interpolations, so its transpose is too:
//   T[i][x] = sum_j wx(j, x) g[i][j]   (x over the rectangle's columns),   plane[y][x] = sum_i wy(i, y) T[i][x].
// The kernel above walks (samples in y) x (samples in x) per pixel - ~150 dependent iterations for the pixels of a small
// box, each with a data-dependent global load - and was the longest kernel of the generator-loss backward (77-110 us
// inside the step, 1 120 workgroups).  Here the object's crop gradient is staged in LDS once, phase A runs one thread
// per (sample row i, column x) over the <= ~12 samples j that touch x, phase B one thread per pixel over the samples i
// that touch y, both out of LDS.  Same weights (crop_axis), fixed summation order (j ascending, then i ascending);
// products are grouped (g wx) wy instead of g (wx wy): equal up to fp32 rounding.  C <= 4, size <= 32, rectangles of
// <= 64 columns (anything else: the kernel above).
constexpr int CS_MAXSIZE = 32;
constexpr int CS_MAXW = 64;
__global__ __launch_bounds__(256) void crop_bwd_object_sep_kernel(const float* __restrict__ dcrops, int H, int W, int C,
                                                                  const float* __restrict__ boxes, int size,
                                                                  int align_corners, float* __restrict__ planes) {
  __shared__ int s_p0[2][CS_MAXSIZE];
  __shared__ float s_t[2][CS_MAXSIZE];
  __shared__ float s_g[CS_MAXSIZE * CS_MAXSIZE * 4];        // [i][j][c]
  __shared__ float s_T[CS_MAXSIZE * CS_MAXW * 4];           // [i][x - x0][c]
  __shared__ short s_jl[CS_MAXW], s_jh[CS_MAXW];
  const int o = blockIdx.x, tid = threadIdx.x;
  const float* box = boxes + 4LL * o;
  for (int k = tid; k < 2 * size; k += 256) {
    const int smp = k % size, ax = k / size;
    const CropAxis r = ax == 0 ? crop_axis(box[0], box[2], smp, size, W, align_corners)
                               : crop_axis(box[1], box[3], smp, size, H, align_corners);
    s_p0[ax][smp] = r.p0; s_t[ax][smp] = r.t;
  }
  const CropRect R = crop_rect(box, size, H, W, align_corners);
  const int rw = R.x1 - R.x0 + 1, rh = R.y1 - R.y0 + 1;
  if (rw <= 0 || rh <= 0) return;                            // (workgroup-uniform)
  // this workgroup's pixel rows (gridDim.y workgroups share an object) and the sample rows that can touch them
  const int rows_per = (rh + gridDim.y - 1) / gridDim.y;
  const int ya = R.y0 + blockIdx.y * rows_per, yb = min(R.y1, ya + rows_per - 1);
  if (ya > yb) return;
  int ilo, ihi, dummy;
  crop_axis_range(box[1], box[3], ya, size, H, align_corners, ilo, dummy);
  crop_axis_range(box[1], box[3], yb, size, H, align_corners, dummy, ihi);
  if (ihi < ilo) { ilo = 0; ihi = size - 1; }               // (degenerate boxes: crop_axis_range gave the full range)
  const float* gobj = dcrops + (long long)o * size * size * C;
  for (int e = tid; e < (ihi - ilo + 1) * size * C; e += 256) s_g[ilo * size * C + e] = gobj[ilo * size * C + e];
  for (int x = tid; x < rw; x += 256) {
    int jl, jh;
    crop_axis_range(box[0], box[2], R.x0 + x, size, W, align_corners, jl, jh);
    s_jl[x] = (short)jl; s_jh[x] = (short)jh;
  }
  __syncthreads();
  // ---- A: T[i][x][c] ----
  for (int e = tid; e < (ihi - ilo + 1) * rw; e += 256) {
    const int i = ilo + e / rw, x = e % rw, X = R.x0 + x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = s_jl[x]; j <= s_jh[x]; ++j) {
      const int px = s_p0[0][j];
      const float tx = s_t[0][j];
      const float wx = px == X ? 1.f - tx : (px + 1 == X ? tx : 0.f);
      const float* g = &s_g[(i * size + j) * C];
      a0 += g[0] * wx;
      if (C > 1) a1 += g[1] * wx;
      if (C > 2) a2 += g[2] * wx;
      if (C > 3) a3 += g[3] * wx;
    }
    float* t = &s_T[(i * CS_MAXW + x) * 4];
    t[0] = a0; t[1] = a1; t[2] = a2; t[3] = a3;
  }
  __syncthreads();
  // ---- B: plane[y][x][c] ----
  float* plane = planes + (long long)o * H * W * C;
  for (int q = tid; q < (yb - ya + 1) * rw; q += 256) {
    const int y = ya + q / rw, x = q % rw;
    int il, ih;
    crop_axis_range(box[1], box[3], y, size, H, align_corners, il, ih);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = max(il, ilo); i <= min(ih, ihi); ++i) {
      const int py = s_p0[1][i];
      const float ty = s_t[1][i];
      const float wy = py == y ? 1.f - ty : (py + 1 == y ? ty : 0.f);
      const float* t = &s_T[(i * CS_MAXW + x) * 4];
      a0 += t[0] * wy; a1 += t[1] * wy; a2 += t[2] * wy; a3 += t[3] * wy;
    }
    float* dst = plane + ((long long)y * W + R.x0 + x) * C;
    dst[0] = a0;
    if (C > 1) dst[1] = a1;
    if (C > 2) dst[2] = a2;
    if (C > 3) dst[3] = a3;
  }
}

__global__ void crop_bwd_sum_kernel(const float* __restrict__ planes, int H, int W, int C,
                                    const float* __restrict__ boxes, const long long* __restrict__ obj_to_img,
                                    int O, int size, int align_corners, float* __restrict__ dimgs, long long ld) {
  __shared__ int s_obj[256];
  __shared__ CropRect s_rect[256];
  __shared__ unsigned char s_flag[256];
  __shared__ int s_cnt;
  const int n = blockIdx.y, tid = threadIdx.x;
  const int pix = blockIdx.x * blockDim.x + tid;
  const bool live = pix < H * W;
  const int y = live ? pix / W : 0, x = live ? pix - y * W : 0;
  float acc[8];
  for (int c0 = 0; c0 < C; c0 += 8) {
    #pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int start = 0; start < O; start += 256) {
      // membership of the next 256 objects in parallel, compacted in ascending order from LDS
      s_flag[tid] = (start + tid < O && obj_to_img[start + tid] == n) ? 1 : 0;
      __syncthreads();
      if (tid == 0) {
        int cnt = 0;
        for (int k = 0; k < min(256, O - start); ++k)
          if (s_flag[k]) s_obj[cnt++] = start + k;
        s_cnt = cnt;
      }
      __syncthreads();
      const int cnt = s_cnt;
      if (tid < cnt) s_rect[tid] = crop_rect(boxes + 4LL * s_obj[tid], size, H, W, align_corners);
      __syncthreads();
      if (live) {
        for (int q = 0; q < cnt; ++q) {
          const CropRect R = s_rect[q];
          if (x < R.x0 || x > R.x1 || y < R.y0 || y > R.y1) continue;
          const float* src = planes + (((long long)s_obj[q] * H + y) * W + x) * C + c0;
          #pragma unroll
          for (int c = 0; c < 8; ++c)
            if (c0 + c < C) acc[c] += src[c];
        }
      }
      __syncthreads();
    }
    if (live) {
      float* dst = dimgs + ((long long)n * H * W + pix) * ld + c0;
      #pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c0 + c < C) dst[c] = acc[c];
    }
  }
}

static inline int ok_or(hipError_t e) { return e == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP; }


}  // namespace sg2im

using namespace sg2im;

extern "C" {

int sg2im_layout_forward(const float* vecs, long long ld_vecs, const float* boxes,
                         const float* masks, const long long* masks_i64, int mask_size,
                         const int* img_row_ptr, const int* img_entries, int n_images, int n_objs,
                         int dim, int height, int width, int align_corners, float* layout,
                         long long ld_layout, hipStream_t stream) {
  if (!vecs || !boxes || !img_row_ptr || !layout || dim < 1 || height < 1 || width < 1 || n_images < 0)
    return SG2IM_ERR_ARG;
  if ((masks || masks_i64) && mask_size < 1) return SG2IM_ERR_ARG;
  if (n_images == 0) return SG2IM_OK;
  (void)n_objs;
  const MaskRef mk{masks, masks_i64, (masks || masks_i64) ? mask_size : 0};
  dim3 grid((height * width + LP - 1) / LP, n_images);
  SG2IM_LAUNCH(layout_fwd_kernel, grid, dim3(256), 0, stream, vecs, ld_vecs, boxes, mk, img_row_ptr, img_entries,
                     dim, height, width, align_corners, layout, ld_layout);
  return ok_or(hipGetLastError());
}

int sg2im_layout_pyramid_forward(const float* vecs, long long ld_vecs, const float* boxes, const float* masks,
                                 const long long* masks_i64, int mask_size, const int* img_row_ptr,
                                 const int* img_entries, int n_images, int dim, const float* noise, int noise_dim,
                                 int height, int width, int align_corners, int n_levels, float* const* levels,
                                 long long ld_levels, hipStream_t stream) {
  if (!vecs || !boxes || !img_row_ptr || !levels || dim < 1 || height < 1 || width < 1 || n_images < 0 || noise_dim < 0)
    return SG2IM_ERR_ARG;
  if ((masks || masks_i64) && mask_size < 1) return SG2IM_ERR_ARG;
  if (dim % PS || noise_dim % PS || (noise_dim > 0 && !noise) || height % PT || width % PT || n_levels < 0 || n_levels > 4 ||
      ld_levels < dim + noise_dim || ld_levels % 4)
    return SG2IM_ERR_ARG;
  PyrLevels out;
  for (int l = 0; l < 5; ++l) {
    out.p[l] = l <= n_levels ? levels[l] : nullptr;
    if (l <= n_levels && (!levels[l] || ((uintptr_t)levels[l] & 15))) return SG2IM_ERR_ARG;
  }
  if (n_images == 0) return SG2IM_OK;
  const MaskRef mk{masks, masks_i64, (masks || masks_i64) ? mask_size : 0};
  dim3 grid((height / PT) * (width / PT), n_images, noise_dim > 0 ? 2 : 1);      // (z: layout slabs | noise slabs)
  SG2IM_LAUNCH(layout_pyramid_kernel, grid, dim3(256), 0, stream, vecs, ld_vecs, boxes, mk, img_row_ptr, img_entries, dim,
               noise, noise_dim, height, width, align_corners, n_levels, out, (int)ld_levels);
  return ok_or(hipGetLastError());
}

size_t sg2im_layout_backward_workspace(int n_objs, int dim, int height, int width) {
  const size_t vec_part = sizeof(float) * (size_t)((height * width + BP - 1) / BP) * (size_t)n_objs * (size_t)dim;
  const size_t g_planes = sizeof(float) * (size_t)n_objs * (size_t)height * (size_t)width;   // mask / box gradients
  // sg2im_layout_backward_vecs_levels: one partial per 8 x 8 pixel tile and object
  const size_t lev_part = sizeof(float) * (size_t)((height + 7) / 8) * (size_t)((width + 7) / 8) * (size_t)n_objs * (size_t)dim;
  return std::max(std::max(vec_part, g_planes), lev_part);
}

int sg2im_layout_backward(const float* dlayout, long long ld_dlayout, const float* vecs,
                          long long ld_vecs, const float* boxes, const float* masks,
                          const long long* masks_i64, int mask_size, const long long* obj_to_img,
                          const int* img_row_ptr, const int* img_entries, int n_images,
                          int n_objs, int dim, int height, int width, int align_corners,
                          float* d_vecs, long long ld_dvecs, float* d_masks, float* d_boxes,
                          float* workspace, hipStream_t stream) {
  if (!dlayout || !boxes || !img_row_ptr || dim < 1 || height < 1 || width < 1) return SG2IM_ERR_ARG;
  if (n_objs == 0) return SG2IM_OK;
  const MaskRef mk{masks, masks_i64, (masks || masks_i64) ? mask_size : 0};
  if (d_vecs) {
    if (!workspace) return SG2IM_ERR_ARG;
    const int n_tiles = (height * width + BP - 1) / BP;
    // objects that belong to no image (cannot happen with a valid obj_to_img) keep zero grads
    const size_t vec_part = sizeof(float) * (size_t)((height * width + BP - 1) / BP) * (size_t)n_objs * (size_t)dim;
    if (hipMemsetAsync(workspace, 0, vec_part, stream) != hipSuccess)
      return SG2IM_ERR_HIP;
    const int TC = dim < 256 ? dim : 256, TR = 256 / TC;
    const size_t lds = sizeof(float) * (size_t)BO * TR * TC;
    dim3 grid(n_tiles, n_images);
    SG2IM_LAUNCH(layout_bwd_vecs_kernel, grid, dim3(256), lds, stream, dlayout, ld_dlayout, boxes, mk,
                       img_row_ptr, img_entries, n_objs, dim, height, width, align_corners, workspace);
    const long long tot = (long long)n_objs * dim;
    SG2IM_LAUNCH(layout_bwd_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, workspace,
                       n_tiles, n_objs, dim, d_vecs, ld_dvecs);
  }
  if (d_masks || d_boxes) {
    if (!vecs || !obj_to_img || !workspace || (d_masks && (!masks || mask_size < 1))) return SG2IM_ERR_ARG;
    // G_o(y, x) for all objects: [O][H*W] floats at the start of the workspace (the d_vecs partials,
    // if any, were consumed by layout_bwd_reduce_kernel above - same stream)
    const int HW = height * width;
    static const bool g_tiles = [] { const char* e = getenv("SG2IM_LAYOUT_G_TILES"); return !(e && e[0] == '0'); }();   // (A/B knob)
    if (g_tiles && img_entries && dim <= 128 && !(dim & 3) && !(ld_dlayout & 3) && !(ld_vecs & 3) &&
        !((uintptr_t)dlayout & 15) && !((uintptr_t)vecs & 15)) {
      // (every object is an entry of exactly one image's row - padded objects belong to the last image - so every
      // G[o] plane is written)
      SG2IM_LAUNCH(layout_bwd_g_tiles_kernel<false>, dim3((HW + 63) / 64, n_images), dim3(256), 0, stream, dlayout,
                         ld_dlayout, GradLevels{}, height, width, vecs, ld_vecs, img_row_ptr, img_entries, dim, HW, workspace);
    } else {
      dim3 gg((HW + 255) / 256, n_objs);
      SG2IM_LAUNCH(layout_bwd_g_kernel, gg, dim3(256), 0, stream, dlayout, ld_dlayout, vecs, ld_vecs, obj_to_img,
                         dim, HW, workspace);
    }
    SG2IM_LAUNCH(layout_bwd_masks_kernel, dim3(n_objs), dim3(256), 0, stream, workspace, boxes, mk, height, width,
                       align_corners, d_masks, d_boxes);
  }
  return ok_or(hipGetLastError());
}

}  // extern "C"

// per-level gradient descriptors of sg2im_layout_backward_{vecs,maps}_levels; false: an argument does not qualify
static bool fill_grad_levels(sg2im::GradLevels& lv, const float* const* dlevels, const int* factors, const long long* lds,
                             int n_levels, int dim, int height, int width) {
  lv.n = n_levels;
  for (int l = 0; l < sg2im::kGradLevels; ++l) {
    lv.p[l] = nullptr; lv.ld[l] = 0; lv.shift[l] = 0; lv.k[l] = 0.f;
    if (l >= n_levels) continue;
    const int f = factors[l];
    if (f < 1 || (f & (f - 1)) || height % f || width % f || !dlevels[l] || (lds[l] & 3) || lds[l] < dim ||
        ((uintptr_t)dlevels[l] & 15))
      return false;
    lv.p[l] = dlevels[l]; lv.ld[l] = (int)lds[l]; lv.shift[l] = __builtin_ctz((unsigned)f); lv.k[l] = 1.f / (float)(f * f);
  }
  return true;
}

extern "C" {

int sg2im_layout_backward_vecs_levels(const float* const* dlevels, const int* factors, const long long* lds, int n_levels,
                                      const float* boxes, const float* masks, const long long* masks_i64, int mask_size,
                                      const int* img_row_ptr, const int* img_entries, int n_images, int n_objs, int dim,
                                      int height, int width, int align_corners, float* d_vecs, long long ld_dvecs,
                                      float* workspace, hipStream_t stream) {
  if (!dlevels || !factors || !lds || n_levels < 1 || n_levels > sg2im::kGradLevels || !boxes || !img_row_ptr || !d_vecs || !workspace ||
      dim < 4 || (dim & 3) || height < 1 || width < 1)
    return SG2IM_ERR_ARG;
  if (n_objs == 0) return SG2IM_OK;
  GradLevels lv;
  if (!fill_grad_levels(lv, dlevels, factors, lds, n_levels, dim, height, width)) return SG2IM_ERR_ARG;
  const MaskRef mk{masks, masks_i64, (masks || masks_i64) ? mask_size : 0};
  const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
  const int n_tiles = tiles_x * tiles_y;
  // (no zero fill: every object belongs to exactly one image, and the workgroup (tile, image) writes the partial of every
  // object of its image for every channel - each part[tile][object][channel] is written exactly once)
  dim3 grid(n_tiles, n_images);
  SG2IM_LAUNCH(layout_bwd_vecs_levels_kernel, grid, dim3(256), 0, stream, lv, boxes, mk, img_row_ptr, img_entries,
                     n_objs, dim, height, width, tiles_x, align_corners, workspace);
  const long long tot = (long long)n_objs * dim;
  SG2IM_LAUNCH(layout_bwd_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, workspace, n_tiles,
                     n_objs, dim, d_vecs, ld_dvecs);
  return ok_or(hipGetLastError());
}

int sg2im_layout_backward_maps_levels(const float* const* dlevels, const int* factors, const long long* lds, int n_levels,
                                      const float* vecs, long long ld_vecs, const float* boxes, const float* masks,
                                      const long long* masks_i64, int mask_size, const int* img_row_ptr,
                                      const int* img_entries, int n_images, int n_objs, int dim, int height, int width,
                                      int align_corners, float* d_masks, float* d_boxes, float* workspace,
                                      hipStream_t stream) {
  if (!dlevels || !factors || !lds || n_levels < 1 || n_levels > sg2im::kGradLevels || !vecs || !boxes || !img_row_ptr || !img_entries ||
      !workspace || (!d_masks && !d_boxes) || (d_masks && (!masks || mask_size < 1)) || dim < 4 || (dim & 3) || dim > 128 ||
      (ld_vecs & 3) || ((uintptr_t)vecs & 15) || height < 1 || width < 1)
    return SG2IM_ERR_ARG;
  if (n_objs == 0) return SG2IM_OK;
  GradLevels lv;
  if (!fill_grad_levels(lv, dlevels, factors, lds, n_levels, dim, height, width)) return SG2IM_ERR_ARG;
  const MaskRef mk{masks, masks_i64, (masks || masks_i64) ? mask_size : 0};
  const int HW = height * width;
  SG2IM_LAUNCH(layout_bwd_g_tiles_kernel<true>, dim3((HW + 63) / 64, n_images), dim3(256), 0, stream, nullptr, 0LL, lv,
                     height, width, vecs, ld_vecs, img_row_ptr, img_entries, dim, HW, workspace);
  SG2IM_LAUNCH(layout_bwd_masks_kernel, dim3(n_objs), dim3(256), 0, stream, workspace, boxes, mk, height, width,
                     align_corners, d_masks, d_boxes);
  return ok_or(hipGetLastError());
}

int sg2im_crop_forward(const float* imgs, long long ld_img, int n_images, int height, int width,
                       int channels, const float* boxes, const long long* obj_to_img, int n_objs,
                       int size, int align_corners, float* crops, hipStream_t stream) {
  if (!imgs || !boxes || !obj_to_img || !crops || size < 1 || channels < 1) return SG2IM_ERR_ARG;
  (void)n_images;
  const long long total = (long long)n_objs * size * size;
  if (total == 0) return SG2IM_OK;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  SG2IM_LAUNCH(crop_fwd_kernel, dim3(blocks), dim3(256), 0, stream, imgs, ld_img, height, width, channels, boxes,
                     obj_to_img, n_objs, size, align_corners, crops);
  return ok_or(hipGetLastError());
}

size_t sg2im_crop_backward_workspace(int n_objs, int height, int width, int channels) {
  return sizeof(float) * (size_t)n_objs * (size_t)height * (size_t)width * (size_t)channels;
}

int sg2im_crop_backward(const float* d_crops, int n_images, int height, int width, int channels,
                        const float* boxes, const long long* obj_to_img, int n_objs, int size,
                        int align_corners, float* d_imgs, long long ld_dimg, float* workspace,
                        hipStream_t stream) {
  if ((n_objs > 0 && (!d_crops || !boxes || !obj_to_img || !workspace)) || !d_imgs || size < 1 || size > CB_MAXSIZE ||
      channels < 1)
    return SG2IM_ERR_ARG;
  if (n_images < 1 || height < 1 || width < 1) return SG2IM_OK;
  if (n_objs > 0) {
    const int share = std::max(1, std::min(8, (1024 + n_objs - 1) / n_objs));
    static const bool g_sep = [] { const char* e = getenv("SG2IM_CROP_SEPARABLE"); return !(e && e[0] == '0'); }();   // (A/B knob)
    if (g_sep && channels <= 4 && size <= CS_MAXSIZE && width <= CS_MAXW)
      SG2IM_LAUNCH(crop_bwd_object_sep_kernel, dim3(n_objs, share), dim3(256), 0, stream, d_crops, height, width, channels,
                         boxes, size, align_corners, workspace);
    else
      SG2IM_LAUNCH(crop_bwd_object_kernel, dim3(n_objs, share), dim3(256), 0, stream, d_crops, height, width, channels,
                         boxes, size, align_corners, workspace);
  }
  // every pixel of d_imgs is WRITTEN (zero where no crop touches it): no pre-zeroing needed
  dim3 grid((height * width + 255) / 256, n_images);
  SG2IM_LAUNCH(crop_bwd_sum_kernel, grid, dim3(256), 0, stream, workspace, height, width, channels, boxes,
                     obj_to_img, n_objs, size, align_corners, d_imgs, ld_dimg);
  return ok_or(hipGetLastError());
}

}  // extern "C"
