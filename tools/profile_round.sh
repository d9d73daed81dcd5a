#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats of bench.py + separate PMC passes of one pca() call.
# usage: tools/profile_round.sh rNN        (every rocprofv3 run under its own timeout; counters never mixed with traces
#                                           other than --kernel-trace)
set -u
R=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters_available.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_prof.log 2>&1
CMD="python $REPO/tools/prof_stage.py pca 400 512 2"
pass() {   # pass NAME "COUNTERS"
  timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/pmc_$1 -o p -- $CMD > $OUT/pmc_$1.log 2>&1 || echo "pass $1 failed rc=$?" >> $OUT/failed_passes.txt
}
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
pass sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
pass sq3 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"
pass grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
pass tcc "TCC_HIT_sum TCC_MISS_sum"
cd $REPO
python tools/pmc_summary.py $OUT/pmc_hbm.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.log 2>&1
python tools/pmc_sq_summary.py $OUT/pmc_sq.json $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_grbm $OUT/pmc_tcc > $OUT/pmc_sq_summary.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# per-configuration kernel stats (round 4's single CSV mixed the plans of four configurations): one rocprofv3 --kernel-trace --stats
# run per BASELINE config through the public API
cd /tmp
for cfg in c2 c3 c4 odd; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o k -- python $REPO/tools/time_configs.py $cfg > $OUT/stats_$cfg.log 2>&1
  find $OUT/stats_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$cfg.csv \;
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o k -- python $REPO/tools/run_c5.py > $OUT/stats_c5.log 2>&1
find $OUT/stats_c5 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_c5.csv \;
# (round 6) HBM traffic of the other configurations' kernels -- the `traffic` of the strong legs' roofline objects in bench.py:
# FETCH_SIZE and WRITE_SIZE in separate passes of one configuration each -> pmc_hbm_<cfg>.json
for cfg in c3 c4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${cfg}_$ctr -o p -- python $REPO/tools/time_configs.py $cfg > $OUT/pmc_${cfg}_$ctr.log 2>&1 || echo "pass $cfg $ctr failed rc=$?" >> $OUT/failed_passes.txt
  done
  (cd $REPO && python tools/pmc_summary.py $OUT/pmc_hbm_$cfg.json $OUT/pmc_${cfg}_FETCH_SIZE $OUT/pmc_${cfg}_WRITE_SIZE "python tools/time_configs.py $cfg" > $OUT/pmc_summary_$cfg.log 2>&1)
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_c5_$ctr -o p -- python $REPO/tools/run_c5.py > $OUT/pmc_c5_$ctr.log 2>&1 || echo "pass c5 $ctr failed rc=$?" >> $OUT/failed_passes.txt
done
(cd $REPO && python tools/pmc_summary.py $OUT/pmc_hbm_c5.json $OUT/pmc_c5_FETCH_SIZE $OUT/pmc_c5_WRITE_SIZE "python tools/run_c5.py" > $OUT/pmc_summary_c5.log 2>&1)
cd $REPO
# socket power / shader clock under each chip-filling stage (sysfs hwmon)
for w in rot512 rot256 rot1024 gram; do timeout 60 python tools/power_probe.py $w 2>&1 | grep -v amdgpu.ids >> $OUT/power.txt; done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT
