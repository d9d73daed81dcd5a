"""Static instruction histogram of the kernels in a hipcc -S listing whose mangled name contains every given substring.
usage: python tools/isa_count.py file.s substr [substr ...]"""
import re, collections, sys
s = open(sys.argv[1]).read()
subs = sys.argv[2:]
for m in re.finditer(r"^(_Z\S+):", s, re.M):
    name = m.group(1)
    if not all(x in name for x in subs):
        continue
    i = m.start()
    j = s.find(".Lfunc_end", i)
    if j < 0:
        continue
    cnt = collections.Counter()
    for l in s[i:j].splitlines():
        l = l.strip()
        if not l or l.startswith((';', '.', '_Z')) or l.endswith(':'):
            continue
        cnt[l.split()[0]] += 1
    g = collections.Counter()
    for op, c in cnt.items():
        g['v_pk' if op.startswith('v_pk_') else 'v_other' if op.startswith('v_') else 'ds' if op.startswith('ds_')
          else 's' if op.startswith('s_') else 'vmem'] += c
    vg = re.search(r"\.name:\s+" + re.escape(name) + r"\n.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, re.S)
    print(name[:70], dict(g), "total", sum(cnt.values()), "vgpr", vg.group(1) if vg else "?", "spill", vg.group(2) if vg else "?")
    print("    ", ", ".join("%s %d" % kv for kv in cnt.most_common(14)))
