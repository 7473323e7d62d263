R=$GRAFT_REPO_ROOT
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null || exit 1
for t in "" "msm_class_reduce=0" "msm_quad_reduce=0" "msm_scan_reduce=2" "msm_reduce_seg=4" "msm_reduce_seg=16" "msm_task_len=32" "msm_task_len=64" "msm_task_len=128" "msm_scan_nb_log=13" "msm_direct=0" "msm_fold_big=0"; do
  echo "UZK_TUNE=$t"
  UZK_TUNE=$t timeout -k 10 240 $R/tests/cpp/prover_rounds /tmp/chain 20 4 4 | grep -E "proofs_per_s" | cut -c1-170
done
