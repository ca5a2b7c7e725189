/* abi_check.c -- prints sizeof / offsetof of the two configuration structs of include/rnde.h as a C compiler lays them out.
 * tests/test_abi.py compiles this with `gcc -I include` (no GPU, no HIP) and compares the output with the ctypes mirrors
 * (regneuralde.jl_amd/_lib.py) and with the field lists of the Julia mirrors (bindings/julia/RNDE.jl): a field added to one side only,
 * or a width changed, fails the CPU suite instead of corrupting a handle at run time.
 * Output: one line per field, "<struct> <field> <offset> <size>", and "<struct> sizeof <bytes>". */
#include <stddef.h>
#include <stdio.h>

#include "rnde.h"

#define F(S, f) printf(#S " " #f " %zu %zu\n", offsetof(S, f), sizeof(((S*)0)->f))

int main(void) {
    F(rnde_node_config, n_layers); F(rnde_node_config, dims); F(rnde_node_config, act); F(rnde_node_config, time_dep);
    F(rnde_node_config, pre_act); F(rnde_node_config, max_batch); F(rnde_node_config, solver); F(rnde_node_config, reltol);
    F(rnde_node_config, abstol); F(rnde_node_config, regularize); F(rnde_node_config, cb_save_start); F(rnde_node_config, track_ctrl);
    F(rnde_node_config, track_initdt); F(rnde_node_config, max_attempts); F(rnde_node_config, device); F(rnde_node_config, col_tile);
    F(rnde_node_config, persist); F(rnde_node_config, wgrad_side_pct); F(rnde_node_config, stage_generic);
    printf("rnde_node_config sizeof %zu\n", sizeof(rnde_node_config));
    F(rnde_nsde_config, drift_layers); F(rnde_nsde_config, drift_dims); F(rnde_nsde_config, drift_act); F(rnde_nsde_config, diff_layers);
    F(rnde_nsde_config, diff_dims); F(rnde_nsde_config, diff_act); F(rnde_nsde_config, max_batch); F(rnde_nsde_config, solver);
    F(rnde_nsde_config, reltol); F(rnde_nsde_config, abstol); F(rnde_nsde_config, regularize); F(rnde_nsde_config, cb_save_start);
    F(rnde_nsde_config, max_attempts); F(rnde_nsde_config, device); F(rnde_nsde_config, beta1); F(rnde_nsde_config, beta2);
    F(rnde_nsde_config, gamma); F(rnde_nsde_config, qmin); F(rnde_nsde_config, qmax); F(rnde_nsde_config, qoldinit);
    F(rnde_nsde_config, delta); F(rnde_nsde_config, generic); F(rnde_nsde_config, stability_size);
    printf("rnde_nsde_config sizeof %zu\n", sizeof(rnde_nsde_config));
    printf("constants RNDE_MAX_LAYERS %d RNDE_COMM_ID_BYTES %d RNDE_COMM_WINDOW_BYTES %d\n", RNDE_MAX_LAYERS, RNDE_COMM_ID_BYTES, RNDE_COMM_WINDOW_BYTES);
    return 0;
}
