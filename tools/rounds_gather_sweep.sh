#!/bin/bash
# Shared provers (uzk_coalesce_config) against proofs per second: threads x gathering wait x groups, the defaults marked "-".
# usage: tools/rounds_gather_sweep.sh <out-file> [skew]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; SKEW=${2:-0}
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
: > $OUT
one() {  # threads gather groups
  UZK_GATHER_US=$2 UZK_GROUPS=$3 timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain 10 $1 1 shared $SKEW 2>&1 | grep -E "proofs_per_s|FAILED|->" | \
    python3 -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print('threads=$1 gather_us=$2 groups=$3', {k: d[k] for k in ('proofs_per_s','ms_per_proof_slowest_thread','proofs_per_shared_round','widest_shared_round','moved_out','threads_agree_with_single')})" >> $OUT
}
for t in 2 4 8 12 16 24 32; do one $t 0 0; done
for t in 8 16 32; do one $t 1000 0; one $t 250 0; done
for t in 16 32; do one $t 0 2; one $t 0 3; one $t 0 6; done
cat $OUT
