"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs.  usage: python tools/pmc_table.py DIR [name-filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"^void\s+|vipmi::\(anonymous namespace\)::|vipmi::|fftw::", "", row["Kernel_Name"])
        k = re.sub(r"\(.*\)\s*(\[clone.*)?$", "", k).strip()
        if flt and flt not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k[:90])
    for c, (t, n) in sorted(cs.items()):
        print("   %-28s %16.0f  (x%d)" % (c, t / n, n))
