#!/bin/bash
# Throughput of the prover rounds, a few repetitions of each form (run-to-run spread on one box):
#   lockstep8 = 4 threads x 8 witnesses per call, shared_32 = 32 threads x 1 proof (the reference's call pattern).
# usage: tools/rounds_ab.sh [reps=3] [forms="lockstep8 shared_32"]      (prints proofs/s per run)
R=$(cd "$(dirname "$0")/.." && pwd)
python3 $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
for i in $(seq ${1:-3}); do
  for f in ${2:-lockstep8 shared_32}; do
    case $f in
      lockstep8) a="10 4 8 lockstep 0";; lockstep8_skewed) a="10 4 8 lockstep 1";;
      shared_32) a="10 32 1 shared 0";; shared_32_skewed) a="10 32 1 shared 1";; single) a="20 1 1 private 0";;
    esac
    if [ $f = single ]; then
      timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain $a 2>&1 | grep -E "ms_per_chain" | head -1 | python3 -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('single ms_per_proof', d.get('ms_per_chain'), 'with witness upload', d.get('ms_per_chain_with_witness_upload'))"
      continue
    fi
    timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain $a 2>&1 | grep -E "proofs_per_s|FAILED" | tail -1 | python3 -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$f', d.get('proofs_per_s'), 'agree', d.get('threads_agree_with_single'), 'per_round', d.get('proofs_per_shared_round'))"
  done
done
