"""Print a kernel timeline from a rocprofv3 kernel_trace csv.  usage: python tools/timeline.py trace.csv [start_frac] [count]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 80
ev = []
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('vipmi::(anonymous namespace)::', '').replace('vipmi::', '')
    n = re.sub(r'[<(].*', '', n)
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n[:22], r['Stream_Id']))
ev.sort()
t0 = ev[0][0]
streams = sorted(set(e[3] for e in ev))
i0 = int(len(ev) * frac)
for s, e, n, q in ev[i0:i0 + cnt]:
    if (e - s) < 20000: continue
    col = streams.index(q)
    print("%9.3f %9.3f %7.3f  %s%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, " " * (24 * col), n))
