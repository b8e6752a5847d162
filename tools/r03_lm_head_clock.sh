#!/bin/bash
# VERDICT r2 item 7, the decisive measurement for N2 at M = 256: does the chip CLOCK DOWN when the MFMA pipe and the HBM
# stream run together?  Three builds of csrc/lm_head_verify.hip (tools/lm_head_lab.py: full kernel / no loads in the loop /
# no math in the loop) run back to back in ONE process at the 72B head under rocprofv3 --kernel-trace --pmc; the effective
# shader clock of every k_lm_head_tile dispatch = GRBM_GUI_ACTIVE / (End - Start), cross-checked with the SQ wave-cycle
# method of round 1/2.  Build the variants in the container first:
#   python tools/lm_head_lab.py --build base: noload:-DASD_LMHEAD_LAB=1 nomath:-DASD_LMHEAD_LAB=2
set -e -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_lmh_clock
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/lm_head_lab.py --run --shapes 72b --reps 10 --out $O/lab_unprofiled.json > $O/unprofiled.log 2>&1 || echo "unprofiled lab run failed"
i=0
for P in "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o c -- python3 $R/tools/lm_head_lab.py --run --shapes 72b --reps 6 --out $O/lab_p$i.json > $O/p$i.log 2>&1 || echo "pass $i failed (see $O/p$i.log)"
done
cd $R
python3 tools/parse_lm_head_clock.py $O > $O/summary.txt 2>&1 || true
cat $O/summary.txt
