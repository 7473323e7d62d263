#!/bin/bash
# The commit table's window width at the throughput setting (four threads x eight / sixteen proofs in lockstep).
# usage (GPU box): bash tools/rounds_window_sweep_wide.sh <out-file>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/rounds_window_sweep_wide.txt}
: > $OUT
for c in 1 14 15 16 17; do
  python $R/tools/write_chain_inputs.py /tmp/chain_$c 14 11 $c > /dev/null || exit 1
  for tb in "4 8" "4 16"; do
    set -- $tb
    echo "precompute=$c threads=$1 batch=$2" >> $OUT
    timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain_$c 12 $1 $2 2>&1 | grep proofs_per_s | cut -c1-150 >> $OUT
  done
done
cat $OUT
