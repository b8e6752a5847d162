set -e
cd $GRAFT_REPO_ROOT
F=adaptive-speculative-decoding_amd/csrc/residual_sample.hip
cp $F /tmp/rs.orig
for T in 64 100000; do
  sed "s/constexpr int kRsRowMinBatch = 64;/constexpr int kRsRowMinBatch = $T;/" /tmp/rs.orig > $F
  python adaptive-speculative-decoding_amd/build.py > /dev/null 2>&1
  echo "== threshold $T"
  timeout -k 10 300 python tools/bench_sampling.py --batches 48,64,96,128,256 --reps 100 --out /tmp/s.json | python3 -c "
import sys,ast
for l in sys.stdin:
    l=l.strip()
    if l and l[0].isdigit():
        b,rest=l.split(' ',1); d=ast.literal_eval(rest); print(b, round(d['asd_residual_sample_us'],1))"
done
cp /tmp/rs.orig $F
