# Everything the bench line cites, collected in ONE gpurun call (each step bounded) -> gpurun_out/${R}/ ; copy what is to be judged into profiles/.
#   R=r04 RNDE_COMMIT=$(git rev-parse --short HEAD) bash tools/gpu_evidence.sh [stats] [pmc] [pmc4096] [sq] [ablation]      (default: all)
cd $GRAFT_REPO_ROOT
R=${R:-r04}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
WHAT=${@:-stats pmc pmc4096 sq ablation}
stats() {   # name, bench arguments: rocprofv3 --kernel-trace --stats summary
  N=$1; shift
  rm -rf $O/prof_$N
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -- python3 bench.py "$@" > $O/${N}_bench.log 2>&1
  echo "[$N] rocprofv3 rc=$?"; tail -1 $O/${N}_bench.log | cut -c1-300
  f=$(find $O/prof_$N -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $* ; collected $(date -u '+%Y-%m-%d %H:%M UTC') at ${RNDE_COMMIT}"; cat "$f"; } > $O/${R}_${N}_kernel_stats.csv
    head -9 "$f" | cut -c1-160
  fi
  rm -rf $O/prof_$N
}
pmc() {     # name, bench arguments: HBM traffic, two separate passes
  N=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${N}_$c
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${N}_$c -- python3 bench.py "$@" > $O/pmc_${N}_$c.log 2>&1
    echo "[$N $c] rc=$?"
  done
  PMC_CMD="python3 bench.py $*" python3 tools/pmc_summary.py $O/pmc_${N}_FETCH_SIZE $O/pmc_${N}_WRITE_SIZE > $O/${R}_pmc_hbm_traffic${N}.csv
  cat $O/${R}_pmc_hbm_traffic${N}.csv | cut -c1-200
  rm -rf $O/pmc_${N}_FETCH_SIZE $O/pmc_${N}_WRITE_SIZE
}
for w in $WHAT; do
  case $w in
    stats)
      stats bench --steps 20 --warmup 5 --no-cpu-baseline --no-extras
      stats latent --workload latent --steps 20 --warmup 5
      stats latent_e2e --workload latent_e2e --steps 20 --warmup 5
      stats nsde --workload nsde --steps 20 --warmup 5 ;;
    pmc) pmc "" --steps 3 --warmup 1 --no-cpu-baseline --no-extras ;;
    pmc4096) pmc _B4096 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-extras ;;
    sq) R=$R bash tools/gpu_pmc_sq.sh > $O/sq.log 2>&1; cp gpurun_out/${R}sq/${R}_pmc_sq_attempt.csv $O/ 2>/dev/null; cat $O/${R}_pmc_sq_attempt.csv | cut -c1-160 ;;
    ablation) timeout 600 python3 tools/experiments/attempt_ablation/ablate_attempt.py > $O/${R}_attempt_ablation.csv 2> $O/ablation.err; cat $O/${R}_attempt_ablation.csv ;;
  esac
done
ls -la $O | head -40
