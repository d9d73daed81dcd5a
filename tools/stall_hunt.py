"""Look for stalls in a rocprofv3 kernel trace: kernels much slower than their median, and spans with no kernel running.
usage: python tools/stall_hunt.py kernel_trace.csv [min_excess_ms=0.5] [min_gap_ms=0.3]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
min_exc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
ev = []
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('vipmi::(anonymous namespace)::', '').replace('vipmi::', '')
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', n)[:60], r.get('Stream_Id', '?')))
ev.sort()
t0 = ev[0][0]
by = defaultdict(list)
for s, e, n, q in ev:
    by[n].append((e - s) / 1e6)
med = {n: sorted(v)[len(v) // 2] for n, v in by.items()}
print("span %.1f ms, %d kernels" % ((ev[-1][1] - t0) / 1e6, len(ev)))
print("-- kernels slower than their median by more than %.2f ms" % min_exc)
for s, e, n, q in ev:
    d = (e - s) / 1e6
    if d - med[n] > min_exc:
        print("  t=%9.3f ms  %8.3f ms (median %7.3f)  stream %s  %s" % ((s - t0) / 1e6, d, med[n], q, n))
print("-- spans with nothing running longer than %.2f ms" % min_gap)
end = ev[0][1]
for s, e, n, q in ev[1:]:
    if (s - end) / 1e6 > min_gap:
        print("  t=%9.3f ms  gap %8.3f ms  before %s" % ((end - t0) / 1e6, (s - end) / 1e6, n))
    end = max(end, e)
