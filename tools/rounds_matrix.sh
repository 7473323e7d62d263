#!/bin/bash
# Proofs per second of one GPU through the round API: host threads (one context each) x lockstep batch.
# usage (on the GPU box): bash tools/rounds_matrix.sh <out-file> [reps=20]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/rounds_matrix.txt}
REPS=${2:-20}
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null || exit 1
: > $OUT
for tb in "1 1" "2 1" "4 1" "6 1" "8 1" "1 2" "1 4" "1 8" "2 2" "2 4" "4 2" "3 4" "4 4" "2 8"; do
  set -- $tb
  echo "threads=$1 batch=$2" >> $OUT
  timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain $REPS $1 $2 >> $OUT 2>&1 || { echo "FAILED threads=$1 batch=$2" >> $OUT; exit 1; }
done
grep -E "threads|ms_per_chain" $OUT | cut -c1-220
