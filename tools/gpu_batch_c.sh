mkdir -p gpurun_out/r02c
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "rotate or derotat or c2_ or c4_" 2>&1 | tail -4
python tools/prof_stage.py pca 400 512 3 2>&1 | tail -2
OUT=$(pwd)/gpurun_out/r02c; REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for grp in "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- python $REPO/tools/prof_stage.py pca 400 512 2 > $OUT/pmc_$name.log 2>&1
done
cd $REPO; python tools/pmc_sq_summary.py $OUT/pmc_sq.json $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm > $OUT/pmc_sq_summary.log 2>&1
find $OUT -name "*.db" -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/pmc_sq.json'))['kernels']
for k,e in sorted(d.items(), key=lambda kv:-max(kv[1].get('dur_us_profiled',[0]))):
    du=e.get('dur_us_profiled',[0]); g=e.get
    if max(du)<100: continue
    gui=g('GRBM_GUI_ACTIVE',0)/8
    print("%-50s dur %7.1f us waves %6d VALU/wave %8.0f LDS/wave %6.0f clk %.2f VALUbusy %.1f%% waves/SIMD %.2f wait_any %.2f wait_inst %.2f" % (k[:50], sum(du)/len(du), g('SQ_WAVES',0), g('SQ_INSTS_VALU',0)/max(1,g('SQ_WAVES',1)), g('SQ_INSTS_LDS',0)/max(1,g('SQ_WAVES',1)), gui/(du[-1]*1e3) if gui else 0, 100*4*g('SQ_ACTIVE_INST_VALU',0)/(1024*gui) if gui else 0, 4*g('SQ_WAVE_CYCLES',0)/(1024*gui) if gui else 0, g('frac_of_wave_cycles:SQ_WAIT_ANY',0), g('frac_of_wave_cycles:SQ_WAIT_INST_ANY',0)))
PY
