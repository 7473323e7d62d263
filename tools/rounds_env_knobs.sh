#!/bin/bash
# Runtime environment switches a host process may choose (the library never edits the environment): completion signals by
# polling instead of interrupts, kernel arguments in device memory.  usage (GPU box): bash tools/rounds_env_knobs.sh <out-file>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/rounds_env_knobs.txt}
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null || exit 1
: > $OUT
for rep in 1 2; do
for e in "X=0" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1"; do
  for tb in "1 1" "4 8"; do
    set -- $tb
    echo "env: $e threads=$1 batch=$2" >> $OUT
    env $e timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain 20 $1 $2 2>&1 | grep -E "ms_per_chain\"|proofs_per_s" | cut -c1-150 | tail -1 >> $OUT
  done
done
done
cat $OUT
