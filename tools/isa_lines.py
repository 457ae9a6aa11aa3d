#!/usr/bin/env python3
"""Static instruction count per source line from `hipcc -S -gline-tables-only` output.
Usage: isa_lines.py file.s source.hip first_line last_line"""
import re, sys, collections
asm, srcf, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
cur = None
for line in open(asm):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+)', line)
    if m and cur:
        op = m.group(1)
        k = 'B' if op.startswith(('s_cbranch', 's_branch')) else ('S' if op.startswith('s_') else ('V' if op.startswith('v_') else 'M'))
        cnt[cur] += 1; kinds[cur][k] += 1
src = open(srcf).read().split('\n')
tot = collections.Counter()
for (f, l), c in sorted(cnt.items()):
    if f == 0 and lo <= l <= hi:
        k = kinds[(f, l)]
        for a in k: tot[a] += k[a]
        print("%4d %4d S%-3d V%-3d B%-2d M%-2d  %s" % (l, c, k['S'], k['V'], k['B'], k['M'], src[l - 1].strip()[:100]))
print("total", dict(tot))
