#!/bin/bash
# Round-end measurement set, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r01i
# 1. bench.py (default flags)                         -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py     -> gpurun_out/<tag>_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE    -> gpurun_out/<tag>_pmc_{fetch,write}_size.csv
#    (separate passes, counters only, program directly after `--`)
set -o pipefail
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
echo "bench done"; cut -c1-400 $out/${tag}_bench.json
rm -rf $out/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err || { tail -5 $out/${tag}_rocprof.err; exit 1; }
cp "$(find $out/prof_stats -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
rm -rf $out/prof_stats
echo "kernel stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
    lc=$(echo $ctr | tr 'A-Z' 'a-z')
    rm -rf $out/prof_pmc
    rocprofv3 --pmc $ctr --output-format csv -d $out/prof_pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $out/${tag}_pmc_${lc}.err || { tail -5 $out/${tag}_pmc_${lc}.err; exit 1; }
    python3 tools/pmc_summary.py "$(find $out/prof_pmc -name '*counter_collection.csv' | head -1)" > $out/${tag}_pmc_${lc}.csv
    rm -rf $out/prof_pmc
    echo "pmc $ctr done"
done
