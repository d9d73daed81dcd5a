#!/bin/bash
# kernel timeline of one un-pipelined C2 call (gaps between its kernels): tools/gaps_run.sh [serial_calls.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R:$PYTHONPATH
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaps
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -o k -- python $R/tools/serial_calls.py "$@" > /tmp/gaps.log 2>&1
f=$(find /tmp/gaps -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -20 /tmp/gaps.log; exit 1; }
python $R/tools/gaps.py $f
