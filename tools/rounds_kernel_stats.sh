#!/bin/bash
# rocprofv3 kernel statistics of the throughput section of tests/cpp/prover_rounds (per-kernel device time, all streams).
# usage: tools/rounds_kernel_stats.sh <out-prefix> <threads> <lanes> <mode> [skew]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
export TMPDIR=/tmp
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
rm -rf /tmp/prs
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prs -- $R/tests/cpp/prover_rounds /tmp/chain 10 "$@" > ${OUT}_under_rocprof.txt 2>&1
cp "$(find /tmp/prs -name '*kernel_stats.csv' | head -1)" ${OUT}_kernel_stats.csv
grep proofs_per_s ${OUT}_under_rocprof.txt | cut -c1-200
head -30 ${OUT}_kernel_stats.csv | cut -d, -f1,2,3,4,5 | cut -c1-170
