R=$GRAFT_REPO_ROOT
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null || exit 1
for tb in "4 4" "4 8" "3 8" "6 4" "5 4" "4 6" "8 2" "4 16" "2 16"; do
  set -- $tb
  echo "threads=$1 batch=$2"
  timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain 15 $1 $2 | grep -E "proofs_per_s" | cut -c1-170
done
