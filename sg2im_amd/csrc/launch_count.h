// Kernel-launch counters behind sg2im_launch_count(): every launch site of the library goes through
// SG2IM_LAUNCH.  Plain process-wide counters (not atomic: statistics only, read by bench.py around one captured
// iteration to report the launches per training step of the graph-mode plan).
#pragma once
#include <hip/hip_runtime.h>

namespace sg2im {
inline unsigned long long g_launches = 0;        // all kernels
inline unsigned long long g_gemm_launches = 0;   // implicit-GEMM family incl. its split-K finishes (conv.hip)
}  // namespace sg2im

#ifdef SG2IM_GEMM_TU
#define SG2IM_LAUNCH(...) do { ++sg2im::g_launches; ++sg2im::g_gemm_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#else
#define SG2IM_LAUNCH(...) do { ++sg2im::g_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#endif
