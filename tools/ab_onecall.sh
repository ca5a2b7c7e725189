#!/bin/bash
# A/B on one GPU box: the training step as three library calls (RNDE_ONE_CALL=0) against rnde_node_classifier_grad (default)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in 0 1; do
    RNDE_ONE_CALL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('one_call=$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms | fixed weights', round(d['value_fixed_weights']), round(d['fixed_weights']['ms_per_step'], 3), 'ms nfe', d['mean_nfe'], d['fixed_weights']['mean_nfe'])"
  done
done
