#!/bin/bash
R=/root/repo
python $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
UZK_COALESCE_TRACE=1 timeout -k 10 200 $R/tests/cpp/prover_rounds /tmp/chain 4 16 1 shared 0 2> $R/gpurun_out/r05a_trace16.txt | tail -2 | cut -c1-120
wc -l $R/gpurun_out/r05a_trace16.txt
