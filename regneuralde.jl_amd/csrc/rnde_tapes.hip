// rnde_tapes.hip -- several taped forwards alive at once behind ONE handle (include/rnde.h: rnde_tapes_*; SURVEY.md 8b sketches
// `tape_id_out` / `rnde_node_backward(h, tape_id, ...)`).
//
// An rnde_node holds one tape: its arena of max_attempts records IS the tape, and a second forward on it drops the first.  The
// reference's training loop needs more than one now and then -- the NFE probe on a fixed batch between a forward and its reverse
// (experiments/mnist_node.jl:245), two batches in flight -- and a Julia caller should not have to juggle handles for that.  A tape pool
// is a set of solver instances of one configuration, created on demand: a taped forward takes a free instance and returns its index
// as the tape id, the reverse pass (or rnde_tapes_release) frees it, untaped forwards run on one extra instance that never tapes
// (two scratch records instead of an arena).  Nothing here touches a kernel: every call is the corresponding rnde_node_* call.
#include "../../include/rnde.h"

#include <string>
#include <vector>

struct rnde_tapes {
    rnde_node_config cfg{};
    int max_tapes = 0;
    std::vector<rnde_node*> inst;      // [max_tapes] taped instances + [1] the untaped one, created on first use
    std::vector<char> busy;
    std::string err;
};

namespace {
thread_local std::string g_tapes_create_err;
rnde_status instance(rnde_tapes* t, int i) {
    if (t->inst[i]) return RNDE_OK;
    const rnde_status st = rnde_node_create(&t->cfg, &t->inst[i]);
    if (st != RNDE_OK) t->err = std::string("creating a solver instance failed: ") + rnde_last_error(nullptr);
    return st;
}
}  // namespace

extern "C" const char* rnde_tapes_last_error(const rnde_tapes* t) { return t ? t->err.c_str() : g_tapes_create_err.c_str(); }

extern "C" rnde_status rnde_tapes_create(const rnde_node_config* cfg, int32_t max_tapes, rnde_tapes** out) {
    if (!out) return RNDE_ERR_BAD_ARG;
    *out = nullptr;
    if (!cfg || max_tapes < 1 || max_tapes > 64) { g_tapes_create_err = "max_tapes: 1..64"; return RNDE_ERR_BAD_ARG; }
    rnde_tapes* t = new rnde_tapes();
    t->cfg = *cfg; t->max_tapes = max_tapes;
    t->inst.assign((size_t)max_tapes + 1, nullptr);
    t->busy.assign((size_t)max_tapes, 0);
    const rnde_status st = instance(t, 0);          // the configuration is validated (and the first arena allocated) now, not at the first solve
    if (st != RNDE_OK) { g_tapes_create_err = t->err; delete t; return st; }
    *out = t;
    return RNDE_OK;
}

extern "C" void rnde_tapes_destroy(rnde_tapes* t) {
    if (!t) return;
    for (rnde_node* h : t->inst) if (h) rnde_node_destroy(h);
    delete t;
}

extern "C" int32_t rnde_tapes_in_use(const rnde_tapes* t) {
    int n = 0;
    if (t) for (char b : t->busy) n += b ? 1 : 0;
    return n;
}

// the instance behind a tape id (tuning / statistics calls of the rnde_node_* family); NULL for a bad id
extern "C" rnde_node* rnde_tapes_node(rnde_tapes* t, int32_t tape_id) {
    if (!t || tape_id < -1 || tape_id >= t->max_tapes) return nullptr;
    return t->inst[tape_id < 0 ? t->max_tapes : tape_id];
}

extern "C" rnde_status rnde_tapes_forward(rnde_tapes* t, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, float* u_out_dev,
                                          int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream,
                                          int32_t* tape_id_out) {
    if (!t || !tape_id_out) return RNDE_ERR_BAD_ARG;
    *tape_id_out = -1;
    int i = t->max_tapes;                            // untaped: the instance that never holds a tape
    if (keep_tape) {
        for (i = 0; i < t->max_tapes && t->busy[i]; ++i) {}
        if (i == t->max_tapes) {
            t->err = "every tape of the pool is in use: run the reverse pass of (or release) an earlier forward, or create the pool with more tapes";
            return RNDE_ERR_BAD_ARG;
        }
    }
    rnde_status st = instance(t, i);
    if (st != RNDE_OK) return st;
    st = rnde_node_forward(t->inst[i], x_dev, p_dev, B, t0, t1, u_out_dev, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
    if (st != RNDE_OK) { t->err = rnde_last_error(t->inst[i]); return st; }
    if (keep_tape) { t->busy[i] = 1; *tape_id_out = i; }
    return RNDE_OK;
}

extern "C" rnde_status rnde_tapes_backward(rnde_tapes* t, int32_t tape_id, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                           float* p_bar_dev, float* tspan_bar_host, void* stream) {
    if (!t) return RNDE_ERR_BAD_ARG;
    if (tape_id < 0 || tape_id >= t->max_tapes || !t->busy[tape_id]) { t->err = "no such tape (already reversed or released?)"; return RNDE_ERR_NO_TAPE; }
    const rnde_status st = rnde_node_backward(t->inst[tape_id], u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, tspan_bar_host, stream);
    t->busy[tape_id] = 0;                            // the reverse pass consumes the tape whether it succeeded or not
    if (st != RNDE_OK) t->err = rnde_last_error(t->inst[tape_id]);
    return st;
}

extern "C" rnde_status rnde_tapes_release(rnde_tapes* t, int32_t tape_id) {
    if (!t) return RNDE_ERR_BAD_ARG;
    if (tape_id < 0 || tape_id >= t->max_tapes || !t->busy[tape_id]) return RNDE_OK;     // releasing twice is harmless
    t->busy[tape_id] = 0;
    return rnde_node_release_tape(t->inst[tape_id]);
}
