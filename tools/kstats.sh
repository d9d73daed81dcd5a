#!/bin/bash
# usage: tools/kstats.sh <python script and args ...>   -> rocprofv3 kernel stats (top 10) of that command, on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R:$PYTHONPATH
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python "$@" > /tmp/kstats.log 2>&1
python - <<P
import csv, glob
f = glob.glob("/tmp/kstats/**/*kernel_stats.csv", recursive=True)
if not f: print(open("/tmp/kstats.log").read()[-2000:]); raise SystemExit(1)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-72s calls %5s  avg %9.1f us  min %9.1f us" % (r["Name"].replace("void vipmi::(anonymous namespace)::", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
P
