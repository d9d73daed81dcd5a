"""Idle time between the kernels of ONE serial pca() call, from a rocprofv3 --kernel-trace csv (last call of the trace).
usage: python tools/gaps_serial.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'gram_split' in r['Kernel_Name'] or 'gram_partial' in r['Kernel_Name']]
i0 = idx[-1]
prev_end = None; tot_gap = 0.0; tot_k = 0.0; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    name = r['Kernel_Name'].replace('void vipmi::(anonymous namespace)::', '').replace('vipmi::', '').split('(')[0][:60]
    print("%7.1f us gap | %8.1f us | %s" % (gap, (en - st) / 1e3, name))
    if prev_end: tot_gap += max(gap, 0.0)
    tot_k += (en - st) / 1e3
    prev_end = en
print("kernels %.1f us, gaps %.1f us, span %.1f us" % (tot_k, tot_gap, (prev_end - t0) / 1e3))
