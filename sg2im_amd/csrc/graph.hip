// Triple-indexed gather / scatter of GraphTripleConv and the embedding lookups.
//
// The reference pools with two torch scatter_add calls (sg2im/graph.py:98-99,106-107).
// On the CPU that is "for each destination row: add all subject-role rows in increasing
// triple index, then all object-role rows in increasing triple index, starting from +0"
// (probe-verified; restated in oracle.gconv_pool_sequential).  The
// kernels here reproduce exactly that order through a *stable* CSR over destination
// rows instead of atomics, so the pooled vectors are bit-identical and run-to-run
// deterministic.  These kernels are HBM/latency bound: rows are read with 16-byte
// coalesced loads, one workgroup per destination row.
#include <algorithm>
#include <hip/hip_runtime.h>
#include "launch_count.h"
#include "sg2im_hip.h"

namespace sg2im {

// `live` (optional, device): only the first live[0] keys of EACH key array are real - the rest is the
// padding of a bucketed batch (sg2im_amd/bucketing.py) and takes no part in the CSR, so no row grows
// a long tail of padding entries.
__device__ __forceinline__ bool csr_entry_live(int e, int na, const int* __restrict__ live) {
  if (!live) return true;
  return (e < na ? e : e - na) < live[0];
}

// The constant 100 MHz device clock into *slot: the Trainer's schedule marks (SG2IM_MARKS=1) - where the lanes of a
// replayed iteration really are in time, without a profiler serialising the host side of the replay.
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }


__global__ void csr_count_kernel(const long long* __restrict__ ka, int na, const long long* __restrict__ kb,
                                 int nb, int* __restrict__ counts, const int* __restrict__ live) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= na + nb || !csr_entry_live(e, na, live)) return;
  if (e < na) atomicAdd(&counts[(int)ka[e]], 1);
  else atomicAdd(&counts[(int)kb[e - na]], 1);
}

// exclusive scan of counts[0..n) -> row_ptr[0..n], single workgroup, then reset counts to 0
// so they can serve as fill cursors
__global__ void csr_scan_kernel(int* __restrict__ counts, int n, int* __restrict__ row_ptr) {
  __shared__ int warp_sums[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += blockDim.x) {
    const int i = base + tid;
    const int v = i < n ? counts[i] : 0;
    int x = v;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) warp_sums[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += warp_sums[w];
    const int c = carry;
    if (i < n) { row_ptr[i] = c + woff + x - v; counts[i] = 0; }
    __syncthreads();
    if (tid == blockDim.x - 1) carry = c + woff + x;
    __syncthreads();
  }
  if (tid == 0) row_ptr[n] = carry;
}

// The whole build in ONE launch for the sizes of a training batch (a few hundred rows, a few hundred to a few
// thousand keys): one workgroup stages the live keys in LDS; a row is owned by P = 2^lp threads, each of which walks
// ONE contiguous slice of the keys twice - count, then (after a block scan of the counts in (row, slice) order)
// append in entry order, which is the stable order by construction.  Replaces memset + count + scan + fill + sort
// (five dependent launches at the very head of the training step).
//   P: the walk of a thread is a chain of LDS reads with nothing else to do - at one thread per row and ~1000 keys
//      the 2 x 256 dependent-latency iterations were 35 us on the critical path of the step (profiles/
//      r6_step_bf16_kernel_sequence.txt); with P slices a walk is 1/P as long and the reads of a slice are unrolled.
//      The slice length is odd (in 16-byte units): the P distinct addresses of a wave's ds_read_b128 fall on
//      distinct bank groups.
//   strided keys (sa / sb elements between keys) + `split`: the keys of the pooling CSR are columns 0 and 2 of the
//      (T, 3) triples tensor (graph.py:73-75 chunks it); the same launch writes the three columns out as contiguous
//      arrays split[0..T) = s, [T..2T) = p, [2T..3T) = o, which the step used to get from three strided copies.
constexpr int kCsrSmallKeys = 8192;
constexpr int kCsrMaxLp = 5;
__global__ __launch_bounds__(1024) void csr_build_small_kernel(const long long* __restrict__ ka, int sa, int na,
                                                               const long long* __restrict__ kb, int sb, int nb,
                                                               int n_rows, int lp, int* __restrict__ row_ptr,
                                                               int* __restrict__ entries, const int* __restrict__ live,
                                                               long long* __restrict__ split) {
  __shared__ __attribute__((aligned(16))) int keys[kCsrSmallKeys];        // -1: padding entry (not live)
  __shared__ int warp_sums[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = na + nb;
  const int n4 = (n + 3) >> 2;               // (walked four keys per LDS read; the tail is padding)
  for (int e = tid; e < 4 * n4; e += blockDim.x) {
    int k = -1;
    if (e < n && csr_entry_live(e, na, live)) k = (int)(e < na ? ka[(long long)e * sa] : kb[(long long)(e - na) * sb]);
    keys[e] = k;
  }
  if (split)                                 // (ka is then the (na, 3) triples tensor itself)
    for (int e = tid; e < 3 * na; e += blockDim.x) {
      const int c = e / na, t = e - c * na;
      split[e] = ka[3LL * t + c];
    }
  if (tid == 0) carry = 0;
  __syncthreads();
  const int4* keys4 = reinterpret_cast<const int4*>(keys);
  const int P = 1 << lp, j = tid & (P - 1), rows_per_pass = blockDim.x >> lp;
  const int sl = ((n4 + P - 1) >> lp) | 1;
  const int q0 = min(j * sl, n4), q1 = min(q0 + sl, n4);
  for (int base = 0; base < n_rows; base += rows_per_pass) {
    const int r = base + (tid >> lp);
    int v = 0;
    if (r < n_rows) {
      #pragma unroll 4
      for (int q = q0; q < q1; ++q) {
        const int4 k = keys4[q];
        v += (k.x == r ? 1 : 0) + (k.y == r ? 1 : 0) + (k.z == r ? 1 : 0) + (k.w == r ? 1 : 0);
      }
    }
    int x = v;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) warp_sums[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += warp_sums[w];
    const int c = carry;
    const int begin = c + woff + x - v;
    if (r < n_rows) {
      if (j == 0) row_ptr[r] = begin;
      if (v > 0) {
        int pos = begin;
        #pragma unroll 2
        for (int q = q0; q < q1; ++q) {
          const int4 k = keys4[q];
          if (k.x == r) entries[pos++] = 4 * q;
          if (k.y == r) entries[pos++] = 4 * q + 1;
          if (k.z == r) entries[pos++] = 4 * q + 2;
          if (k.w == r) entries[pos++] = 4 * q + 3;
        }
      }
    }
    __syncthreads();
    if (tid == blockDim.x - 1) carry = c + woff + x;
    __syncthreads();
  }
  if (tid == 0) row_ptr[n_rows] = carry;
}

// (T, 3) -> three contiguous columns [3][T]: the large-batch path of sg2im_csr_build_triples
__global__ void split_triples_kernel(const long long* __restrict__ tri, int T, long long* __restrict__ split) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 3 * T) return;
  const int c = e / T, t = e - c * T;
  split[e] = tri[3LL * t + c];
}

__global__ void csr_fill_kernel(const long long* __restrict__ ka, int na, const long long* __restrict__ kb,
                                int nb, const int* __restrict__ row_ptr, int* __restrict__ cursor,
                                int* __restrict__ tmp, const int* __restrict__ live) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= na + nb || !csr_entry_live(e, na, live)) return;
  const int key = (int)(e < na ? ka[e] : kb[e - na]);
  const int pos = atomicAdd(&cursor[key], 1);
  tmp[row_ptr[key] + pos] = e;
}

// rank sort of every row segment (entry ids are unique): one wavefront per row.  The row is staged
// through LDS in chunks and compared four values per (broadcast) ds_read_b128 - the plain form walked
// `tmp[j]` with one wave-uniform scalar load + wait per comparison, 41 us for the 200-entry row of the
// __in_image__ predicate at the tail of the backward pass.
constexpr int RS_CHUNK = 1024;
__global__ void csr_ranksort_kernel(const int* __restrict__ row_ptr, const int* __restrict__ tmp,
                                    int n_rows, int* __restrict__ entries) {
  __shared__ __attribute__((aligned(16))) int stage[4][RS_CHUNK];
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int b = row_ptr[row], e = row_ptr[row + 1];
  int* const st = stage[wave];
  for (int i0 = b; i0 < e; i0 += 64) {              // (wave-uniform trip count: every lane takes part in the staging)
    const int i = i0 + lane;
    const int v = i < e ? tmp[i] : 0x7fffffff;
    int rank = 0;
    for (int c0 = b; c0 < e; c0 += RS_CHUNK) {
      const int n = min(RS_CHUNK, e - c0);
      for (int j = lane; j < ((n + 3) & ~3); j += 64) st[j] = j < n ? tmp[c0 + j] : 0x7fffffff;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0);                 // (one wavefront owns the row: LDS visibility within the wave)
      for (int j = 0; j < n; j += 4) {
        const int4 q = *reinterpret_cast<const int4*>(st + j);
        rank += (q.x < v) + (q.y < v) + (q.z < v) + (q.w < v);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (i < e) entries[b + rank] = v;
  }
}

// one workgroup per destination row; threads stride over the row in float4
__global__ void segment_sum_kernel(const float* __restrict__ src_a, long long ld_a, int n_a,
                                   const float* __restrict__ src_b, long long ld_b,
                                   const int* __restrict__ row_ptr, const int* __restrict__ entries,
                                   int width, int average, int accumulate, float* __restrict__ out,
                                   long long ld_out) {
  const int row = blockIdx.x;
  const int b = row_ptr[row], e = row_ptr[row + 1];
  const float cnt = (float)max(1, e - b);
  const bool v4 = (width % 4 == 0) && (ld_a % 4 == 0) && (ld_b % 4 == 0) && (ld_out % 4 == 0) &&
                  !(((uintptr_t)src_a | (uintptr_t)src_b | (uintptr_t)out) & 15);
  if (v4) {
    for (int c = threadIdx.x * 4; c < width; c += blockDim.x * 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      // eight entries in flight (ids, then rows), added in entry order: the same sum as the one-by-one
      // walk, without a dependent id -> row latency chain per entry (68 us for a 200-entry row)
      for (int i = b; i < e; i += 8) {
        int id[8];
        float4 v[8];
        #pragma unroll
        for (int k = 0; k < 8; ++k) id[k] = entries[min(i + k, e - 1)];
        #pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float* p = id[k] < n_a ? src_a + (long long)id[k] * ld_a : src_b + (long long)(id[k] - n_a) * ld_b;
          v[k] = *reinterpret_cast<const float4*>(p + c);
        }
        #pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (i + k < e) { acc.x = acc.x + v[k].x; acc.y = acc.y + v[k].y; acc.z = acc.z + v[k].z; acc.w = acc.w + v[k].w; }
        }
      }
      if (average) { acc.x = acc.x / cnt; acc.y = acc.y / cnt; acc.z = acc.z / cnt; acc.w = acc.w / cnt; }
      float4* dst = reinterpret_cast<float4*>(out + (long long)row * ld_out + c);
      if (accumulate) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
      *dst = acc;
    }
  } else {
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
      float acc = 0.f;
      for (int i = b; i < e; ++i) {
        const int id = entries[i];
        acc = acc + (id < n_a ? src_a[(long long)id * ld_a + c] : src_b[(long long)(id - n_a) * ld_b + c]);
      }
      if (average) acc = acc / cnt;
      if (accumulate) acc += out[(long long)row * ld_out + c];
      out[(long long)row * ld_out + c] = acc;
    }
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, long long ld_src,
                                   const long long* __restrict__ idx, int width,
                                   const int* __restrict__ row_ptr, float* __restrict__ dst,
                                   long long ld_dst) {
  const int i = blockIdx.x;
  const long long r = idx[i];
  float div = 1.f;
  if (row_ptr) div = (float)max(1, row_ptr[r + 1] - row_ptr[r]);
  const float* s = src + r * ld_src;
  float* d = dst + (long long)i * ld_dst;
  const bool v4 = (width % 4 == 0) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) &&
                  !(((uintptr_t)src | (uintptr_t)dst) & 15);
  if (v4) {
    for (int c = threadIdx.x * 4; c < width; c += blockDim.x * 4) {
      float4 v = *reinterpret_cast<const float4*>(s + c);
      if (row_ptr) { v.x = v.x / div; v.y = v.y / div; v.z = v.z / div; v.w = v.w / div; }
      *reinterpret_cast<float4*>(d + c) = v;
    }
  } else {
    for (int c = threadIdx.x; c < width; c += blockDim.x) d[c] = row_ptr ? s[c] / div : s[c];
  }
}

// Backward of the GraphTripleConv pooling (graph.py:98-114) in one launch: row t of d(new_t) is
//   [ d_pooled[s[t]] / cnt(s[t]) | g_pred[t] | d_pooled[o[t]] / cnt(o[t]) ]  *  relu'(new_t[t])
// - the two row gathers (divided by the row's entry count for 'avg' pooling), the copy of the predicate
// gradient into the middle column block and the activation backward of net1's last layer, which were
// four launches of the dependent chain.  Same operations per element (IEEE divide, multiply by 1 / slope).
__global__ void gconv_pool_bwd_kernel(const float* __restrict__ dpooled, long long ld_dp,
                                      const long long* __restrict__ s_idx, const long long* __restrict__ o_idx,
                                      const int* __restrict__ row_ptr, const float* __restrict__ g_pred,
                                      long long ld_gp, const float* __restrict__ new_t, long long ld_nt,
                                      int H, int Dout, float slope, float* __restrict__ out, long long ld_out) {
  const int t = blockIdx.x;
  const long long s = s_idx[t], o = o_idx[t];
  float ds = 1.f, dv = 1.f;
  if (row_ptr) {
    ds = (float)max(1, row_ptr[s + 1] - row_ptr[s]);
    dv = (float)max(1, row_ptr[o + 1] - row_ptr[o]);
  }
  const int NT = 2 * H + Dout;
  const float* ps = dpooled + s * ld_dp;
  const float* po = dpooled + o * ld_dp;
  const float* y = new_t + (long long)t * ld_nt;
  float* d = out + (long long)t * ld_out;
  for (int c = threadIdx.x; c < NT; c += blockDim.x) {
    float v;
    if (c < H) v = row_ptr ? ps[c] / ds : ps[c];
    else if (c < H + Dout) v = g_pred ? g_pred[(long long)t * ld_gp + (c - H)] : 0.f;
    else v = row_ptr ? po[c - H - Dout] / dv : po[c - H - Dout];
    d[c] = v * (y[c] > 0.f ? 1.f : slope);
  }
}

__global__ void copy_2d_kernel(const float* __restrict__ src, long long ld_src, float* __restrict__ dst,
                               long long ld_dst, long long rows, int width) {
  const long long total = rows * width;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / width; const int c = (int)(i - r * width);
    dst[r * ld_dst + c] = src[r * ld_src + c];
  }
}

// batch hand-over of a captured iteration: up to 16 (dst, src, 4-byte words) copies in ONE launch
struct StageJobs { unsigned* dst[16]; const unsigned* src[16]; unsigned words[16]; };
// (16-byte pieces where both ends allow it, up to 1 024 workgroups per job: the 25 MB image tensor of a 256 x 256 batch took
// 155 us at the head of the step as 4-byte copies on 64 workgroups)
__global__ void stage_batch_kernel(const StageJobs j) {
  const int job = blockIdx.y;
  unsigned* __restrict__ d = j.dst[job];
  const unsigned* __restrict__ s = j.src[job];
  const unsigned n = j.words[job];
  const unsigned stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
  if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {               // (job-uniform)
    const unsigned n4 = n >> 2;
    uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
    const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);
    for (unsigned i = i0; i < n4; i += stride) d4[i] = s4[i];
    for (unsigned i = 4 * n4 + i0; i < n; i += stride) d[i] = s[i];
  } else {
    for (unsigned i = i0; i < n; i += stride) d[i] = s[i];
  }
}

}  // namespace sg2im

using namespace sg2im;

extern "C" {

int sg2im_abi_version(void) { return 11; }

int sg2im_stage_batch(int n, void* const* dst, const void* const* src, const size_t* bytes, hipStream_t stream) {
  if (n < 0 || n > 16 || (n && (!dst || !src || !bytes))) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  StageJobs j;
  size_t most = 0;
  for (int i = 0; i < 16; ++i) {
    const bool live = i < n;
    if (live && (bytes[i] % 4 || ((uintptr_t)dst[i] & 3) || ((uintptr_t)src[i] & 3) || bytes[i] / 4 > 0xffffffffull ||
                 (bytes[i] && (!dst[i] || !src[i]))))
      return SG2IM_ERR_ARG;
    j.dst[i] = live ? (unsigned*)dst[i] : nullptr;
    j.src[i] = live ? (const unsigned*)src[i] : nullptr;
    j.words[i] = live ? (unsigned)(bytes[i] / 4) : 0u;
    if (live) most = bytes[i] / 4 > most ? bytes[i] / 4 : most;
  }
  const int bx = (int)std::max<size_t>(1, std::min<size_t>((most + 256 * 8 - 1) / (256 * 8), 1024));
  SG2IM_LAUNCH(stage_batch_kernel, dim3(bx, n), dim3(256), 0, stream, j);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

int sg2im_timestamp(unsigned long long* slot, hipStream_t stream) {
  if (!slot) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(sg2im::timestamp_kernel, dim3(1), dim3(1), 0, stream, slot);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

unsigned long long sg2im_launch_count(int which) { return which == 1 ? sg2im::g_gemm_launches : sg2im::g_launches; }

static int csr_small_lp(int n_rows) {
  int lp = 0;
  while (lp < sg2im::kCsrMaxLp && ((long long)n_rows << (lp + 1)) <= 1024) ++lp;
  return lp;
}
// the single-workgroup form: the keys fit the LDS image and the walks stay short
static bool csr_small_ok(int n, int n_rows) {
  const int lp = csr_small_lp(n_rows);
  const long long passes = ((long long)n_rows + (1024 >> lp) - 1) / (1024 >> lp);
  // (a thread walks n >> lp keys twice per pass, ~27 ns per key: beyond ~1 500 keys per walk - the 256 x 256 shape: 6 400 keys
  // over 700-960 rows took 171 us - the five-launch path is shorter)
  return n <= sg2im::kCsrSmallKeys && (n >> lp) <= 1536 && (long long)(n >> lp) * passes <= 65536;
}

int sg2im_csr_build(const long long* keys_a, int n_a, const long long* keys_b, int n_b, int n_rows,
                    int* row_ptr, int* entries, int* scratch, const int* live_keys, hipStream_t stream) {
  if (n_a < 0 || n_b < 0 || n_rows < 1 || !row_ptr || !scratch || (n_a && !keys_a) || (n_b && !keys_b))
    return SG2IM_ERR_ARG;
  const int n = n_a + n_b;
  if (csr_small_ok(n, n_rows) && (n == 0 || entries)) {
    SG2IM_LAUNCH(csr_build_small_kernel, dim3(1), dim3(1024), 0, stream, keys_a, 1, n_a, keys_b, 1, n_b, n_rows,
                 csr_small_lp(n_rows), row_ptr, entries, live_keys, (long long*)nullptr);
    return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
  }
  int* counts = scratch;            // [n_rows]
  int* tmp = scratch + n_rows;      // [n]
  if (hipMemsetAsync(counts, 0, sizeof(int) * n_rows, stream) != hipSuccess) return SG2IM_ERR_HIP;
  if (n > 0) {
    if (!entries) return SG2IM_ERR_ARG;
    SG2IM_LAUNCH(csr_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys_a, n_a, keys_b, n_b, counts, live_keys);
  }
  SG2IM_LAUNCH(csr_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, n_rows, row_ptr);
  if (n > 0) {
    SG2IM_LAUNCH(csr_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys_a, n_a, keys_b, n_b,
                       row_ptr, counts, tmp, live_keys);
    SG2IM_LAUNCH(csr_ranksort_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, row_ptr, tmp, n_rows, entries);
  }
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

int sg2im_csr_build_triples(const long long* triples, int n_triples, int n_rows, long long* split, int* row_ptr,
                            int* entries, int* scratch, const int* live_keys, hipStream_t stream) {
  if (n_triples < 0 || n_rows < 1 || !row_ptr || !scratch || (n_triples && (!triples || !split || !entries)))
    return SG2IM_ERR_ARG;
  const int T = n_triples;
  if (csr_small_ok(2 * T, n_rows)) {
    SG2IM_LAUNCH(csr_build_small_kernel, dim3(1), dim3(1024), 0, stream, triples, 3, T, triples + 2, 3, T, n_rows,
                 csr_small_lp(n_rows), row_ptr, entries, live_keys, split);
    return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
  }
  SG2IM_LAUNCH(split_triples_kernel, dim3((3 * T + 255) / 256), dim3(256), 0, stream, triples, T, split);
  if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
  return sg2im_csr_build(split, T, split + 2LL * T, T, n_rows, row_ptr, entries, scratch, live_keys, stream);
}

int sg2im_segment_sum(const float* src_a, long long ld_a, int n_a, const float* src_b, long long ld_b,
                      const int* row_ptr, const int* entries, int n_rows, int width, int average,
                      int accumulate, float* out, long long ld_out, hipStream_t stream) {
  if (n_rows < 0 || width < 1 || !row_ptr || !out || !src_a) return SG2IM_ERR_ARG;
  if (n_rows == 0) return SG2IM_OK;
  if (!src_b) { src_b = src_a; ld_b = ld_a; }
  const int threads = std::min(256, std::max(64, ((width + 3) / 4 + 63) / 64 * 64));
  SG2IM_LAUNCH(segment_sum_kernel, dim3(n_rows), dim3(threads), 0, stream, src_a, ld_a, n_a, src_b, ld_b,
                     row_ptr, entries, width, average, accumulate, out, ld_out);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

int sg2im_copy_2d(const float* src, long long ld_src, float* dst, long long ld_dst, long long rows,
                  int width, hipStream_t stream) {
  if (!src || !dst || width < 1 || rows < 0) return SG2IM_ERR_ARG;
  if (rows == 0) return SG2IM_OK;
  const int blocks = (int)std::min<long long>((rows * width + 255) / 256, 4096);
  SG2IM_LAUNCH(copy_2d_kernel, dim3(blocks), dim3(256), 0, stream, src, ld_src, dst, ld_dst, rows, width);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

int sg2im_gconv_pool_backward(const float* d_pooled, long long ld_dp, const long long* s_idx,
                              const long long* o_idx, int n_triples, const int* row_ptr, const float* g_pred,
                              long long ld_gp, const float* new_t, long long ld_nt, int hidden, int dout,
                              float slope, float* d_new_t, long long ld_out, hipStream_t stream) {
  if (n_triples == 0) return SG2IM_OK;
  if (n_triples < 0 || hidden < 1 || dout < 0 || !d_pooled || !s_idx || !o_idx || !new_t || !d_new_t)
    return SG2IM_ERR_ARG;
  const int NT = 2 * hidden + dout;
  const int threads = std::min(256, std::max(64, (NT + 63) / 64 * 64));
  SG2IM_LAUNCH(gconv_pool_bwd_kernel, dim3(n_triples), dim3(threads), 0, stream, d_pooled, ld_dp, s_idx, o_idx,
                     row_ptr, g_pred, ld_gp, new_t, ld_nt, hidden, dout, slope, d_new_t, ld_out);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

int sg2im_gather_rows(const float* src, long long ld_src, const long long* idx, int n, int width,
                      const int* row_ptr, float* dst, long long ld_dst, hipStream_t stream) {
  if (n < 0 || width < 1 || !src || !idx || !dst) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  const int threads = std::min(256, std::max(64, ((width + 3) / 4 + 63) / 64 * 64));
  SG2IM_LAUNCH(gather_rows_kernel, dim3(n), dim3(threads), 0, stream, src, ld_src, idx, width, row_ptr, dst, ld_dst);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

}  // extern "C"
