cd $GRAFT_REPO_ROOT
for v in base NODMA NOMFMA; do
  python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_$v.so 512 0
done
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_base.so 512 4
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_base.so 4096 0
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_base.so 64 0
