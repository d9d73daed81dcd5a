#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats of bench.py + separate PMC passes of one pca() call.
# usage: tools/profile_round.sh rNN
set -u
R=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $REPO/tools/prof_stage.py pca 400 512 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $REPO/tools/prof_stage.py pca 400 512 1 > $OUT/pmc_write.log 2>&1
cd $REPO
python tools/pmc_summary.py $OUT/pmc_hbm.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT
