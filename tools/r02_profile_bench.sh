#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object (round 2): kernel statistics of the bench command, then the HBM
# traffic of the verify kernel from two separate PMC passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 320 --warmup 32 --no-cpu-baseline --no-other-workloads --no-loop"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof_stats -o b -- python3 $R/bench.py $ARGS > $O/r02_bench_under_rocprof.json 2> $O/r02_bench_under_rocprof.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r02_prof_fetch -o b -- python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-workloads --no-loop --mode eager > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r02_prof_write -o b -- python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-workloads --no-loop --mode eager > /dev/null 2>&1
cd $R
python3 tools/kstats.py $O/r02_prof_stats --top 6
python3 tools/parse_pmc.py --fetch $(ls $O/r02_prof_fetch/*counter_collection.csv | head -1) --write $(ls $O/r02_prof_write/*counter_collection.csv | head -1) --kernel k_verify --out $O/r02_traffic.json
tail -c 600 $O/r02_bench_under_rocprof.json
