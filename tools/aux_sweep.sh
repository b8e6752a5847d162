#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
F=adaptive-speculative-decoding_amd/csrc/lse_device.hpp
cp $F /tmp/lse_device.hpp.orig
for AUX in 2 3 18 19 16 17; do
  sed "s/rsrc, byte_off, 0, NT ? 2 : 0)/rsrc, byte_off, 0, NT ? $AUX : 0)/" /tmp/lse_device.hpp.orig > $F
  python adaptive-speculative-decoding_amd/build.py > /dev/null 2>&1
  echo "== aux $AUX"
  timeout -k 10 200 python tools/sweep_verify.py --workload c3 --threads 512 --unroll 4 --splits 1 --nt 1 --reps 400 --out /tmp/s.json | grep "T=" 
  timeout -k 10 200 python tools/sweep_verify.py --workload c5 --threads 512 --unroll 4 --splits 1 --nt 1 --reps 200 --out /tmp/s.json | grep "T="
done
cp /tmp/lse_device.hpp.orig $F
