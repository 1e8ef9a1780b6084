#!/bin/bash
# BASELINE configs[4] rehearsed on ONE GPU: the blocks of ranks R.. of 8 of the 100 000-column ensemble, one after another.
#   usage (inside gpurun): bash tools/eq100k_shards.sh R [R ...]
set -u
R0=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R0/gpurun_out
mkdir -p $O
cd $R0
for r in "$@"; do
  timeout -k 10 580 python examples/equilibrium_ensemble.py --ncol 100000 --shard $r/8 --tol 0.01 --max-steps 1500 \
    --progress 10 --save-state $O/eq100k_shard$r.npy > $O/eq100k_shard$r.json 2> $O/eq100k_shard$r.log
  rc=$?
  cut -c1-400 $O/eq100k_shard$r.json
  echo "shard $r rc=$rc"
  if [ $rc = 124 ] || [ $rc = 137 ]; then exit 1; fi
done
