#!/bin/bash
# Proofs per second of one GPU through tests/cpp/prover_rounds: the reference's call pattern (threads x shared provers of one
# proof on the default context), the explicit lockstep API, and private provers -- every proof its own witness, uploaded from
# pinned host memory.  usage: tools/rounds_shared_matrix.sh <out-file> [skew]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; SKEW=${2:-0}
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
: > $OUT
run() { echo "# $*" >> $OUT; timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain "$@" 2>&1 | grep -E "ms_per_chain\"|proofs_per_s|FAILED|->" | cut -c1-400 >> $OUT; }
run 20 1 1
for t in 4 8 16 32; do run 10 $t 8 shared $SKEW; done
run 10 16 4 shared $SKEW
run 10 32 16 shared $SKEW
run 10 4 8 lockstep $SKEW
run 10 4 4 lockstep $SKEW
run 10 4 1 private $SKEW
cat $OUT
