#!/bin/bash
# Per-kernel counter sums of the prover rounds (rocprofv3 --pmc, counters only; kernels are serialised under counter collection, so
# this gives instruction COUNTS per kernel -- who consumes the chip's vector issue slots -- not timings).
# The driver's single-proof latency phase is skipped (UZK_ROUNDS_NO_SINGLE=1): the counts are those of the throughput phase and of the
# threads x lanes single proofs it is checked against.
# usage: tools/rounds_pmc.sh <out.csv> <threads> <lanes> <mode> <skew> [counters...]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; T=$2; L=$3; M=$4; S=$5; shift 5
export TMPDIR=/tmp
python3 $R/tools/write_chain_inputs.py /tmp/chain 14 11 1 > /dev/null
rm -rf /tmp/prpmc
export UZK_ROUNDS_NO_SINGLE=1
cd /tmp && rocprofv3 --pmc ${@:-SQ_INSTS_VALU SQ_WAVES} --output-format csv -d /tmp/prpmc -- $R/tests/cpp/prover_rounds /tmp/chain 2 $T $L $M $S > /tmp/prpmc.log 2>&1
python3 $R/tools/pmc_summary.py "$(find /tmp/prpmc -name '*counter_collection.csv' | head -1)" > $OUT
head -3 $OUT
