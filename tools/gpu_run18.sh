for x in 0 1; do echo "== RNDE_X3=$x"; RNDE_X3=$x RNDE_DIAG_BWD=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_bstage.py 2>&1 | grep -v amdgpu.ids | tail -12; done
