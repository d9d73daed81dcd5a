#!/bin/bash
# round 5, GPU call 1: packed-FP32 issue probe, power test of the shears (half / quarter of the CUs), Le = 1024 counters at 3 waves per SIMD
O=gpurun_out/r5a; mkdir -p $O
timeout 200 tools/bin/pkp > $O/pkp.txt 2>&1
for args in "512 400" "512 400 reserve_cus=128" "512 400 reserve_cus=192" "256 1600" "256 1600 reserve_cus=128" "1024 100"; do
  timeout 120 python tools/time_rot.py $args 2>&1 | grep -v amdgpu.ids >> $O/rot.txt
done
PMC_FILTER=rs_shear timeout 400 tools/pmc_run.sh le1024 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- python tools/time_rot.py 256 1600 > $O/pmc_le1024.txt 2>&1
PMC_FILTER=rs_shear timeout 400 tools/pmc_run.sh le2048 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- python tools/time_rot.py 512 400 > $O/pmc_le2048.txt 2>&1
rm -rf gpurun_out/pmc_le1024/g*/ gpurun_out/pmc_le2048/g*/ 2>/dev/null
cat $O/pkp.txt $O/rot.txt; tail -30 $O/pmc_le1024.txt; tail -30 $O/pmc_le2048.txt
