"""Timeline of the LAST call in a kernel trace of tools/serial_calls.py: every kernel with the idle gap before it.
usage: python tools/gaps.py kernel_trace.csv [first kernel name substring = gram_split]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "gram_split"
ev = []
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('vipmi::(anonymous namespace)::', '').replace('vipmi::', '')
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', n)[:50]))
ev.sort()
starts = [i for i, e in enumerate(ev) if first in e[2]]
i0, i1 = starts[-2], starts[-1]
t0 = ev[i0][0]
end = ev[i0 - 1][1] if i0 > 0 else t0
busy = 0
for s, e, n in ev[i0:i1]:
    print("%8.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - end) / 1e3, (e - s) / 1e3, n))
    busy += (e - s)
    end = max(end, e)
print("call period %.1f us, kernels %.1f us" % ((ev[i1][0] - t0) / 1e3, busy / 1e3))
