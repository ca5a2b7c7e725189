// rnde_bwd.h -- reverse pass (placeholder until the kernels land; returns an error, never a CPU fallback)
#pragma once
#include "rnde_device.h"
struct rnde_node;
namespace rnde {
struct BwdBuffers { int dummy; };
inline void bwd_free(BwdBuffers&) {}
}
static rnde_status bwd_run(rnde_node* h, const float*, const float*, float*, float*, float*, hipStream_t);
