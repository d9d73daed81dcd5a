"""Instruction histogram of every loop (label .. backward branch) of one kernel in a hipcc -S listing.
usage: python tools/isa_loops.py file.s kernel_substring [min_size]"""
import re, sys, collections
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index(':', i); j = s.index('.Lfunc_end', i)
minsize = int(sys.argv[3]) if len(sys.argv) > 3 else 50
body = s[i:j].splitlines()
labels = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        loops.append((labels[m.group(1)], n))
# keep outermost distinct loops larger than minsize
loops.sort()
for a, b in loops:
    cnt = collections.Counter()
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith((';', '.')) or l.endswith(':'): continue
        cnt[l.split()[0]] += 1
    tot = sum(cnt.values())
    if tot < minsize: continue
    print("loop lines %d..%d: %d instructions" % (a, b, tot))
    print("    ", ", ".join("%s %d" % kv for kv in cnt.most_common(18)))
