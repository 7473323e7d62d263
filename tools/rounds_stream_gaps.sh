#!/bin/bash
# Per-stream timeline of the throughput section of tests/cpp/prover_rounds: which share of the time each queue runs a kernel, how
# many kernels run at once, and between which kernels the queues stand idle (tools/rounds_stream_gaps.py).
# usage: tools/rounds_stream_gaps.sh <out-prefix> <threads> <lanes> <mode> [skew]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
export TMPDIR=/tmp
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 ${PRECOMPUTE:-1} > /dev/null   # PRECOMPUTE: 1 = the automatic tables, else one table of that window width
rm -rf /tmp/prg
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prg -- $R/tests/cpp/prover_rounds /tmp/chain 6 "$@" > ${OUT}_under_rocprof.txt 2>&1
K=$(find /tmp/prg -name '*kernel_trace.csv' | head -1)
M=$(find /tmp/prg -name '*memory_copy_trace.csv' | head -1)
grep -o '"proofs_per_s": [0-9.]*' ${OUT}_under_rocprof.txt
python $R/tools/rounds_stream_gaps.py $K $M > ${OUT}.txt
cat ${OUT}.txt
