#!/bin/bash
# Window width of the commit table (uzk_circuit_desc.precompute) x (threads, lockstep batch): proofs per second.
# usage (GPU box): bash tools/rounds_window_sweep.sh <out-file> [reps=15] [widths="1 12 13 14 15 16"]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/rounds_window_sweep.txt}
REPS=${2:-15}
WIDTHS=${3:-"1 12 13 14 15 16"}
: > $OUT
for c in $WIDTHS; do
  python $R/tools/write_chain_inputs.py /tmp/chain_$c 14 11 $c > /dev/null || exit 1
  for tb in "1 1" "1 8" "4 1" "4 4" "2 8"; do
    set -- $tb
    echo "precompute=$c threads=$1 batch=$2" >> $OUT
    timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain_$c $REPS $1 $2 >> $OUT 2>&1 || { echo "FAILED" >> $OUT; exit 1; }
  done
done
grep -E "^precompute|proofs_per_s|ms_per_chain\"" $OUT | cut -c1-200
