R=$GRAFT_REPO_ROOT
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null || exit 1
for q in 4 8 16; do
for tb in "4 8" "5 4" "6 4" "8 4" "6 8"; do
  set -- $tb
  echo "GPU_MAX_HW_QUEUES=$q threads=$1 batch=$2"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain 15 $1 $2 | grep -E "proofs_per_s" | cut -c1-140
done
done
