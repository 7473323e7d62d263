#!/bin/bash
# A few repetitions of the shared-prover throughput run at chosen thread counts and gathering waits (run-to-run spread).
# usage: tools/rounds_repeat.sh "<threads list>" "<gather_us list>" [reps]
R=$(cd "$(dirname "$0")/.." && pwd)
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
for i in $(seq ${3:-2}); do for t in ${1:-16 32}; do for w in ${2:-0}; do
  UZK_GATHER_US=$w timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain 10 $t 1 shared 0 2>&1 | grep -E "proofs_per_s|FAILED|->" | python3 -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('threads=$t gather=$w', d['proofs_per_s'], d['proofs_per_shared_round'], d['host_gap_us_between_shared_rounds'], d['gather_us_per_group'], d['groups_by_size'])"
done; done; done
