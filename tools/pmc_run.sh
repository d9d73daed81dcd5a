#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh TAG "CTR1 CTR2 ..." -- cmd...   (one rocprofv3 pass per quoted counter group)
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${GROUPS_[@]}"; do
  (cd $REPO && rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/g$i -o p -- "$@" > $OUT/g$i.log 2>&1)
  i=$((i+1))
done
cd $REPO && python tools/pmc_table.py $OUT ${PMC_FILTER:-} 
