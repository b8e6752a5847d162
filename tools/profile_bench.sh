#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object: kernel statistics of the bench command, then the HBM traffic of the
# verify kernel (plain and FUSED instantiation) from two separate PMC passes (FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md prescribes; then the per-kernel statistics of the sampling kernels (tools/bench_sampling.py).
#   ROUND=r03 bash tools/profile_bench.sh
set -e -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=${ROUND:-r03}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 320 --warmup 32 --no-cpu-baseline --no-other-workloads --no-loop"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_stats -o b -- python3 $R/bench.py $ARGS > $O/${T}_bench_under_rocprof.json 2> $O/${T}_bench_under_rocprof.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${T}_prof_fetch -o b -- python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-workloads --no-loop --mode eager > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${T}_prof_write -o b -- python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-workloads --no-loop --mode eager > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_sampling -o s -- python3 $R/tools/bench_sampling.py --out $O/${T}_sampling_under_rocprof.json > /dev/null 2>&1
cd $R
python3 tools/kstats.py $O/${T}_prof_stats --top 8
python3 tools/kstats.py $O/${T}_prof_sampling --top 8
F=$(find $O/${T}_prof_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/${T}_prof_write -name "*counter_collection.csv" | head -1)
python3 tools/parse_pmc.py --fetch $F --write $W --kernel "512, 3, true, false, false" --out $O/${T}_traffic.json || python3 tools/parse_pmc.py --fetch $F --write $W --kernel k_verify --out $O/${T}_traffic.json
python3 tools/parse_pmc.py --fetch $F --write $W --kernel "512, 3, true, true, false" --out $O/${T}_traffic_fused.json || true
cp $(find $O/${T}_prof_stats -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
cp $(find $O/${T}_prof_sampling -name "*kernel_stats.csv" | head -1) $O/${T}_sampling_kernel_stats.csv
tail -c 600 $O/${T}_bench_under_rocprof.json
