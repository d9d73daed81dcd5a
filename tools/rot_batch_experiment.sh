#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the three shears of the FFT derotation back to back on Infinity-Cache-sized batches of frames
# (round-2 VERDICT item 4): time per batch size, and HBM fetch / write bytes of the shear kernels (separate --pmc passes,
# --kernel-trace only) for the whole cube in one batch against batches of 24 and 12 frames (A1r + A2r = 8.4 MB per frame at 512 px).
# usage: tools/rot_batch_experiment.sh OUTDIR
set -u
REPO=$(pwd)
OUT=$REPO/${1:-gpurun_out/rot_batch}
mkdir -p $OUT
for b in 0 200 100 48 24 12; do
  timeout 120 python tools/time_rot.py 512 400 rot_batch=$b >> $OUT/times.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for b in 0 24 12; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$b -o p -- python $REPO/tools/time_rot.py 512 400 rot_batch=$b > $OUT/pmc_${c}_$b.log 2>&1 || echo "pass $c $b failed" >> $OUT/failed.txt
  done
done
cd $REPO
for b in 0 24 12; do
  python tools/pmc_summary.py $OUT/pmc_hbm_batch$b.json $OUT/pmc_FETCH_SIZE_$b $OUT/pmc_WRITE_SIZE_$b > $OUT/pmc_summary_$b.log 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/times.txt
