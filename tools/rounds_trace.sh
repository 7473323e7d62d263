#!/bin/bash
# usage: tools/rounds_trace.sh [output file]   (default: gpurun_out/rounds_trace16.txt under the repo root)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/rounds_trace16.txt}
mkdir -p "$(dirname "$OUT")"
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
UZK_COALESCE_TRACE=1 timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain 4 16 1 shared 0 2> "$OUT" | tail -2 | cut -c1-120
wc -l "$OUT"
