"""Kernel-trace summary of one profiled run (rocprofv3 --kernel-trace CSV): per kernel total time and the union of the busy
intervals against the wall time of the last `reps` calls.  usage: trace_union.py <kernel_trace.csv> [tail_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t1 - (t1 - t0) * frac                      # steady-state tail of the run
ev = [e for e in ev if e[0] >= cut]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = max(e[1] for e in ev) - ev[0][0]
tot = collections.Counter(); cnt = collections.Counter()
for s, e, k in ev:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("vipmi::", "")
    k = k.split("(")[0].split("<")[0][:48]
    tot[k] += e - s; cnt[k] += 1
print("window %.2f ms, some kernel running %.1f %% of it, sum of kernel times %.2f ms" % (wall / 1e6, 100 * busy / wall, sum(tot.values()) / 1e6))
for k, v in tot.most_common(14):
    print("  %-60s %8.3f ms  %5d launches" % (k, v / 1e6, cnt[k]))
