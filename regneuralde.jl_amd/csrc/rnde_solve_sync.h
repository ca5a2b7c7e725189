// rnde_solve_sync.h -- parameter block of the one-launch solve kernel (rnde_stage_solve.h), shared with the host code in rnde.hip (which only
// launches that kernel through its translation unit's launcher).
#pragma once

namespace rnde {

struct SolveSync {
    unsigned long long* xch;   // [n_limit][3][256] granules {float value, uint tag}
    unsigned epoch;            // tag = epoch * 8192 + attempt + 1
    int n_limit;               // attempts this launch may run (< 8192)
};

}  // namespace rnde
