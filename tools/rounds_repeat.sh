#!/bin/bash
R=/root/repo
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
for i in 1 2 3 4; do for t in 8 16; do
  timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain 10 $t 1 shared 0 2>&1 | grep -E "proofs_per_s|FAILED|->" | python3 -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('threads=$t', d['proofs_per_s'], d['proofs_per_shared_round'])"
done; done
