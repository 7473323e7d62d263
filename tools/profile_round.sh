#!/bin/bash
# Round-end measurement set, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r02a
# 1. bench.py (default flags)                         -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py     -> gpurun_out/<tag>_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE    -> gpurun_out/<tag>_pmc_{fetch,write}_size.csv
# 4. rocprofv3 --pmc <8 SQ counters> / GRBM           -> gpurun_out/<tag>_pmc_sq.csv, _pmc_grbm.csv
#    (separate passes, counters only, program directly after `--`)
# 5. tools/make_counter_files.py                      -> gpurun_out/<tag>_pmc_traffic.json, <tag>_sq_counters.json
# 6. bench.py again with those files in profiles/      -> gpurun_out/<tag>_bench.json (the line of record; the first run: <tag>_bench_first.json)
#    (stamped with the hash of the kernel sources they were measured on; copy to profiles/pmc_traffic.json and
#     profiles/sq_counters.json -- bench.py refuses them when the sources have changed since)
set -o pipefail
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
echo "bench done"; cut -c1-400 $out/${tag}_bench.json
rm -rf $out/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err || { tail -5 $out/${tag}_rocprof.err; exit 1; }
cp "$(find $out/prof_stats -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
rm -rf $out/prof_stats
echo "kernel stats done"
pmc_pass() {   # name, counters...
    local name=$1; shift
    rm -rf $out/prof_pmc
    rocprofv3 --pmc "$@" --output-format csv -d $out/prof_pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $out/${tag}_pmc_${name}.err || { tail -5 $out/${tag}_pmc_${name}.err; return 1; }
    python3 tools/pmc_summary.py "$(find $out/prof_pmc -name '*counter_collection.csv' | head -1)" > $out/${tag}_pmc_${name}.csv
    rm -rf $out/prof_pmc
    echo "pmc $name done"
}
pmc_pass fetch_size FETCH_SIZE || exit 1
pmc_pass write_size WRITE_SIZE || exit 1
pmc_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY || echo "SQ pass failed (see $out/${tag}_pmc_sq.err)"
pmc_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || echo "GRBM pass failed"
python3 tools/make_counter_files.py $tag && echo "counter files written" || exit 1
# 6. the bench line of record, on THIS box, with the counter files just measured in place (roofline.traffic, sq_counters) -- so that
#    <tag>_bench.json, <tag>_kernel_stats.csv and the counter passes all come from one machine and one binary
cp $out/${tag}_pmc_traffic.json profiles/pmc_traffic.json && cp $out/${tag}_sq_counters.json profiles/sq_counters.json || exit 1
mv $out/${tag}_bench.json $out/${tag}_bench_first.json
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
echo "bench of record done"; cut -c1-300 $out/${tag}_bench.json
