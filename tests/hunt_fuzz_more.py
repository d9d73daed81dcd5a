"""The randomised differential tests of tests/test_gpu_fuzz.py on seeds the suite does not run (a one-off hunt):
   python tests/hunt_fuzz_more.py [first_seed [count]]"""
import sys, os, traceback, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tests = [F.test_pca_fullframe_random_parameters, F.test_pca_annular_random_parameters, F.test_derotate_and_collapse_random_shapes,
         F.test_pca_feature_combinations_random, F.test_annular_feature_combinations_random, F.test_pca_float64_counts_random_parameters]
mp = pytest.MonkeyPatch()
bad = 0
for t in tests:
    t0 = time.time(); ran = 0
    for seed in range(first, first + count):
        try:
            t(seed); ran += 1
        except Exception as e:                               # (not only assertions: a crash in the library is a finding too)
            bad += 1
            print("FAIL %s(%d): %s" % (t.__name__, seed, "".join(traceback.format_exception_only(type(e), e)).strip()[:600]), flush=True)
    print("%-48s %d seeds from %d: %d ok, %.0f s" % (t.__name__, count, first, ran, time.time() - t0), flush=True)
mp.setenv("VIPMI_ANNULAR_FUSED", "0")
t0 = time.time(); ran = 0
for seed in range(first, first + count // 2):
    try:
        F.test_pca_annular_random_parameters(seed); ran += 1
    except Exception as e:
        bad += 1
        print("FAIL per-segment annular(%d): %s" % (seed, "".join(traceback.format_exception_only(type(e), e)).strip()[:600]), flush=True)
print("annular with VIPMI_ANNULAR_FUSED=0: %d ok, %.0f s" % (ran, time.time() - t0))
print("failures:", bad)
