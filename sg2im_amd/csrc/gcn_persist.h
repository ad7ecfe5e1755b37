// internal: the persistent GraphTripleConv-stack kernels (gcn_persist.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace sg2im {
namespace gcn {
// device attributes + dynamic-LDS limits of the persistent kernels; called by sg2im_init() so that a first launch may
// sit inside a stream capture (idempotent)
hipError_t prepare();
}  // namespace gcn
}  // namespace sg2im
