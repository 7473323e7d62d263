#!/bin/bash
# Shared provers over thread counts x pauses between proofs (UZK_THINK_US: every thread pauses a random time in [0, 2 x us) between
# two proofs).  usage: tools/rounds_think_matrix.sh "<threads list>" "<think_us list>"
R=$(cd "$(dirname "$0")/.." && pwd)
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
for th in ${2:-0 2000 5000}; do for t in ${1:-16 32 64}; do
  echo "think=$th threads=$t $(UZK_THINK_US=$th timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain 10 $t 1 shared | grep -o '"proofs_per_s": [0-9.]*\|"proofs_per_shared_round": [0-9.]*\|"moved_out": [0-9]*\|"threads_agree_with_single": [a-z]*' | tr '\n' ' ')"
done; done
