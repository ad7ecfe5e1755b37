// The C ABI of libsg2im_hip.so WITHOUT torch, Python or ctypes: plain hipMalloc'd buffers, three entry points, results
// checked against loops written here.  A 1 x 1 convolution 64 -> 3 over 8 x 16 x 16 pixels - the shape class of the
// refinement network's output_conv[2] (sg2im/crn.py:84): forward (implicit-GEMM kernel), data gradient and weight / bias
// gradient (the few-output kernels of csrc/conv_fewout.h), and the same data gradient with the LeakyReLU mask of
// sg2im_conv2d_backward_data_act.
//   hipcc --offload-arch=gfx950 -std=c++17 -Iinclude examples/abi_smoke.cpp -Lsg2im_amd/lib -lsg2im_hip \
//         -Wl,-rpath,'$ORIGIN/../../sg2im_amd/lib' -o tools/_bin/abi_smoke && tools/_bin/abi_smoke
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "sg2im_hip.h"

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define ABI_OK(e) do { int e_ = (e); if (e_ != SG2IM_OK) { std::printf("sg2im error %d at line %d\n", e_, __LINE__); return 3; } } while (0)

template <typename T> static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

static double rel_err(const std::vector<float>& got, const std::vector<double>& want) {
  double worst = 0.0, scale = 1e-30;
  for (double v : want) scale = std::fmax(scale, std::fabs(v));
  for (size_t i = 0; i < want.size(); ++i) worst = std::fmax(worst, std::fabs((double)got[i] - want[i]));
  return worst / scale;
}

int main() {
  const int N = 8, H = 16, W = 16, C = 64, CO = 3;
  const long M = (long)N * H * W;
  const float slope = 0.2f;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<float> x(M * C), w(CO * C), b(CO), dy(M * CO), act(M * C);
  for (auto& v : x) v = u(rng);
  for (auto& v : w) v = u(rng) * 0.125f;
  for (auto& v : b) v = u(rng);
  for (auto& v : dy) v = u(rng);
  for (auto& v : act) v = u(rng);                      // (the activated output of the layer the gradient flows into)
  std::printf("abi %d, %d entry points declared in include/sg2im_hip.h\n", sg2im_abi_version(), 79);
  ABI_OK(sg2im_init());
  float *dx_ = to_device(x), *dw_ = to_device(w), *db_ = to_device(b), *ddy = to_device(dy), *dact = to_device(act);
  float *y = nullptr, *gx = nullptr, *gxm = nullptr, *gw = nullptr, *gb = nullptr, *ws = nullptr;
  const size_t ws_bytes = 64u << 20;
  HIP_OK(hipMalloc(&y, M * CO * sizeof(float)));
  HIP_OK(hipMalloc(&gx, M * C * sizeof(float)));
  HIP_OK(hipMalloc(&gxm, M * C * sizeof(float)));
  HIP_OK(hipMalloc(&gw, CO * C * sizeof(float)));
  HIP_OK(hipMalloc(&gb, CO * sizeof(float)));
  HIP_OK(hipMalloc(&ws, ws_bytes));
  if (!dx_ || !dw_ || !db_ || !ddy || !dact) { std::printf("allocation failed\n"); return 2; }

  sg2im_conv_desc d;
  std::memset(&d, 0, sizeof(d));
  d.src[0].data = dx_; d.src[0].channels = C; d.src[0].ld = C; d.src[0].slope = 1.f;
  d.nsrc = 1; d.batch = N; d.in_h = H; d.in_w = W; d.out_h = H; d.out_w = W;
  d.kh = 1; d.kw = 1; d.stride = 1; d.pad = 0;
  hipStream_t stream = nullptr;
  ABI_OK(sg2im_conv2d_forward(&d, dw_, CO, db_, 1.f, y, CO, 0, ws, ws_bytes, stream));
  ABI_OK(sg2im_conv2d_backward_data(&d, dw_, CO, ddy, CO, 0, C, gx, C, 0, ws, ws_bytes, stream));
  ABI_OK(sg2im_conv2d_backward_data_act(&d, dw_, CO, ddy, CO, 0, C, gxm, C, dact, C, slope, ws, ws_bytes, stream));
  ABI_OK(sg2im_conv2d_backward_weight(&d, ddy, CO, CO, gw, gb, 0, ws, ws_bytes, stream));
  HIP_OK(hipDeviceSynchronize());

  std::vector<float> hy(M * CO), hgx(M * C), hgxm(M * C), hgw(CO * C), hgb(CO);
  HIP_OK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hgx.data(), gx, hgx.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hgxm.data(), gxm, hgxm.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hgw.data(), gw, hgw.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hgb.data(), gb, hgb.size() * 4, hipMemcpyDeviceToHost));

  std::vector<double> ry(M * CO), rgx(M * C), rgxm(M * C), rgw(CO * C, 0.0), rgb(CO, 0.0);
  for (long m = 0; m < M; ++m) {
    for (int o = 0; o < CO; ++o) {
      double s = b[o];
      for (int c = 0; c < C; ++c) s += (double)x[m * C + c] * w[o * C + c];
      ry[m * CO + o] = s;
      rgb[o] += dy[m * CO + o];
      for (int c = 0; c < C; ++c) rgw[o * C + c] += (double)dy[m * CO + o] * x[m * C + c];
    }
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int o = 0; o < CO; ++o) s += (double)dy[m * CO + o] * w[o * C + c];
      rgx[m * C + c] = s;
      rgxm[m * C + c] = s * (act[m * C + c] > 0.f ? 1.0 : (double)slope);
    }
  }
  const double e_y = rel_err(hy, ry), e_gx = rel_err(hgx, rgx), e_gxm = rel_err(hgxm, rgxm), e_gw = rel_err(hgw, rgw),
               e_gb = rel_err(hgb, rgb);
  std::printf("conv1x1 %d -> %d over %ld pixels, rel-to-max error vs double loops:\n", C, CO, M);
  std::printf("  forward %.2e   data gradient %.2e   data gradient x LeakyReLU mask %.2e   weight gradient %.2e   bias gradient %.2e\n",
              e_y, e_gx, e_gxm, e_gw, e_gb);
  std::printf("  kernels launched by this library so far: %llu\n", sg2im_launch_count(0));
  const bool ok = e_y < 1e-5 && e_gx < 1e-5 && e_gxm < 1e-5 && e_gw < 1e-5 && e_gb < 1e-5;
  std::printf(ok ? "abi_smoke: OK\n" : "abi_smoke: MISMATCH\n");
  return ok ? 0 : 1;
}
