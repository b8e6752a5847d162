set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r03_gpu_all_c.log 2>&1; tail -3 $O/r03_gpu_all_c.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/r03_smoke_c.log 2>&1; tail -1 $O/r03_smoke_c.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_default_c.json 2> $O/r03_bench_default_c.err; tail -c 400 $O/r03_bench_default_c.json; echo
ROUND=r03c timeout -k 10 1000 bash tools/profile_bench.sh > $O/r03c_profile.log 2>&1; tail -5 $O/r03c_profile.log
timeout -k 10 600 python tools/bench_linear.py --rows 32,99,208,288 --out $O/r03_linear_final.json > $O/r03_linear_final.txt 2>&1; tail -2 $O/r03_linear_final.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r03_prof_pass7b_hip3 -o p -- python3 $GRAFT_REPO_ROOT/tools/profile_pass.py --model 7b --batch 32 --tokens 1 --passes 8 --hip-layers > $GRAFT_REPO_ROOT/$O/r03_pass7b_hip3.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -1 $O/r03_pass7b_hip3.txt; python3 tools/kstats.py $O/r03_prof_pass7b_hip3 --top 10
