set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03h
mkdir -p $out
pass() { name=$1; shift; rm -rf /tmp/pp; rocprofv3 --pmc "$@" --output-format csv -d /tmp/pp -- python3 $R/tools/pmc_ntt.py > /dev/null 2> $out/$name.err && python3 $R/tools/pmc_summary.py "$(find /tmp/pp -name '*counter_collection.csv' | head -1)" > $out/$name.csv && echo "$name ok" || echo "$name FAILED: $(tail -2 $out/$name.err)"; }
pass lds1 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pass lds2 SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD
pass misc SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cat $out/*.csv
