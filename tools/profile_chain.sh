set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R && bash tools/profile_round.sh r03b > gpurun_out/r03b_profile_round.log 2>&1; tail -3 gpurun_out/r03b_profile_round.log | cut -c1-200
python tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
tests/cpp/prover_rounds /tmp/chain 30 4 > gpurun_out/r03b_prover_rounds.txt 2>&1; cat gpurun_out/r03b_prover_rounds.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- $R/tests/cpp/prover_rounds /tmp/chain 20 1 > $R/gpurun_out/r03b_rounds_prof.log 2>&1
cp "$(find /tmp/pr -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r03b_kernel_stats_prover_rounds.csv && head -5 $R/gpurun_out/r03b_kernel_stats_prover_rounds.csv | cut -c1-160
