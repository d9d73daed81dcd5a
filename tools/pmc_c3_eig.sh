cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc3_$tag -o p -- python $R/tools/time_configs.py c3 > /tmp/pmc3_$tag.log 2>&1
done
python - <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pmc3_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tri_eig_kernel" not in k: continue
        k = "L1" if "512, 25" in k else "L2"
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())})
P
