// rnde_solve_sync.h -- parameter block of the one-launch solve kernel (rnde_stage_solve.h), shared with the host code in rnde.hip (which only
// launches that kernel through its translation unit's launcher).
#pragma once

namespace rnde {

struct SolveSync {
    unsigned long long* xch;   // [n_limit][3][256] granules {float value, uint tag}
    unsigned epoch;            // tag = epoch * 8192 + attempt + 1
    int n_limit;               // attempts this launch may run (< 8192)
    // X3 kernels (rnde_x3.h): the weights split into three bf16 planes, [tile][k-step < 4][plane < 3][64 lanes] fragments of 16 bytes
    const void* x3B;           // layer 2: tile = row tile (49)
    const void* x3D;           // layer 1: tile = hidden tile * 7 + row block
};

}  // namespace rnde
