"""Summarise rocprofv3 --pmc SQ / GRBM / TCC passes (one directory per pass, each with --kernel-trace) into one JSON:
per kernel the average counter value per launch, the average duration from the kernel trace of the same pass, and
derived figures (MI355X: 256 CUs x 4 SIMDs; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves, MI355X_MICROARCH.md "rocprofv3 PMC slots").
usage: python tools/pmc_sq_summary.py OUT.json DIR [DIR ...]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

N_SIMD = 256 * 4


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"vipmi::\(anonymous namespace\)::|vipmi::", "", name)
    return re.sub(r"\(.*\)\s*(\[clone.*)?$", "", name).strip()


kern = defaultdict(dict)
for d in sys.argv[2:]:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                if extra in row and row[extra] not in ("", None):
                    kern[short(row["Kernel_Name"])][extra] = float(row[extra])
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = dur[short(row["Kernel_Name"])]
            a[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            a[1] += 1
    for k, cs in acc.items():
        for c, (tot, cnt) in cs.items():
            kern[k][c] = tot / cnt
            kern[k]["launches"] = cnt
        if k in dur and dur[k][1]:
            kern[k].setdefault("dur_us_profiled", []).append(round(dur[k][0] / dur[k][1] / 1e3, 2))

for k, e in kern.items():
    g = e.get
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                     "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if g(name) is not None:
                e["frac_of_wave_cycles:" + name] = round(g(name) / wc, 4)
    if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
        e["valu_insts_per_wave"] = round(g("SQ_INSTS_VALU") / g("SQ_WAVES"), 1)
    if g("SQ_WAVES") and g("SQ_INSTS_LDS") is not None:
        e["lds_insts_per_wave"] = round(g("SQ_INSTS_LDS") / g("SQ_WAVES"), 1)
    if g("GRBM_GUI_ACTIVE") and e.get("dur_us_profiled"):
        e["clock_GHz"] = round(g("GRBM_GUI_ACTIVE") / (e["dur_us_profiled"][-1] * 1e3), 3)
    if g("SQ_ACTIVE_INST_VALU") is not None and g("GRBM_GUI_ACTIVE"):
        # VALU pipe utilisation: quad-cycles of VALU issue summed over all waves x 4 / (SIMDs x elapsed cycles)
        e["valu_busy_frac_of_simd_time"] = round(4.0 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * g("GRBM_GUI_ACTIVE")), 4)
    if wc and g("GRBM_GUI_ACTIVE"):
        e["mean_resident_waves_per_simd"] = round(4.0 * wc / (N_SIMD * g("GRBM_GUI_ACTIVE")), 3)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac_of_simd_time"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (N_SIMD * g("GRBM_GUI_ACTIVE")), 4)
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        e["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)

doc = {"command": "rocprofv3 --kernel-trace --pmc <group> (one pass per group) -- python tools/prof_stage.py pca 400 512 2",
       "kernels": kern}
json.dump(doc, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, e in sorted(kern.items(), key=lambda kv: -max(kv[1].get("dur_us_profiled", [0]))):
    print(k[:90])
    for a in sorted(e):
        print("    %-44s %s" % (a, e[a]))
