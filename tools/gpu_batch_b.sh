mkdir -p gpurun_out/r02b
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "median or collapse or trimmean" > gpurun_out/r02b/pytest_median.log 2>&1; echo "rc=$?" >> gpurun_out/r02b/pytest_median.log); tail -5 gpurun_out/r02b/pytest_median.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_f64.json 2> gpurun_out/r02b/bench_f64.err; python -c "
import json;d=json.loads(open('gpurun_out/r02b/bench_f64.json').read().strip().splitlines()[-1]);print('f64 gram:',d['value'],d['value_serial'],d['stages_serial_ms'])"
for opt in "gram_f32=1" "gram_f32=1,gram_slices=512" "gram_f32=1,gram_slices=1024"; do
  VIPMI_OPTS=$opt timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_$opt.json 2> gpurun_out/r02b/bench_$opt.err; python -c "
import json,sys;d=json.loads(open('gpurun_out/r02b/bench_$opt.json').read().strip().splitlines()[-1]);print('$opt:',d['value'],d['value_serial'],d['stages_serial_ms'])"
  (VIPMI_OPTS=$opt timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "c2_against or c3_annular or c4_per or c5_principal or c2_projection" > "gpurun_out/r02b/pytest_$opt.log" 2>&1; echo "rc=$?" >> "gpurun_out/r02b/pytest_$opt.log"); tail -4 "gpurun_out/r02b/pytest_$opt.log"
done
VIPMI_OPTS=gram_f32=1,gram_slices=512 timeout 300 python tools/gram_parity.py > gpurun_out/r02b/gram_parity.log 2>&1; cat gpurun_out/r02b/gram_parity.log
