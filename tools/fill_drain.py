"""Fill / drain of the pipelined bench in a rocprofv3 kernel trace: for the LAST run of >= K gram_split kernels separated from the
rest by idle gaps, the start time of every call's first kernel (gram_split) and end of its last (median), per stream.
usage: python tools/fill_drain.py kernel_trace.csv [timed steps]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('vipmi::(anonymous namespace)::', '').replace('vipmi::', '')
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'[<(].*', '', n), r.get('Stream_Id', '?')))
ev.sort()
# split into busy regions at idle gaps > 150 us
regions, cur, end = [], [ev[0]], ev[0][1]
for e in ev[1:]:
    if e[0] - end > 150e3:
        regions.append(cur); cur = []
    cur.append(e); end = max(end, e[1])
regions.append(cur)
big = [r for r in regions if sum(1 for e in r if e[2] == 'gram_split_kernel') >= 8]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # number of timed steps: pick the region with exactly that many calls
ncalls = [sum(1 for e in r if e[2] == 'gram_split_kernel') for r in big]
print("busy regions with >= 8 calls:", ncalls)
reg = big[ncalls.index(want)] if want in ncalls else big[-1]
t0 = reg[0][0]
print("region: %d kernels, %.2f ms, %d calls" % (len(reg), (max(e[1] for e in reg) - t0) / 1e6, sum(1 for e in reg if e[2] == 'gram_split_kernel')))
calls = []
for e in reg:
    if e[2] == 'gram_split_kernel': calls.append([e[3], e[0], None, {}])
    for c in reversed(calls):
        if c[0] == e[3]:
            c[3].setdefault(e[2], [e[0], e[1]]); c[3][e[2]][1] = e[1]
            if e[2] == 'median_kernel': c[2] = e[1]
            break
for i, c in enumerate(calls):
    g = c[3]
    def span(n): return "%6.2f-%6.2f" % ((g[n][0] - t0) / 1e6, (g[n][1] - t0) / 1e6) if n in g else "      -      "
    print("call %2d stream %s: start %7.2f  eigh %s  shear1 %s  shear2 %s  end %7.2f ms" % (i, c[0], (c[1] - t0) / 1e6, span('tri_multi_kernel'), span('rs_shear1'), span('rs_shear2_direct'), ((c[2] or 0) - t0) / 1e6))
