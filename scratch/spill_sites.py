"""Where a kernel's scratch (spill) traffic sits relative to its loops (scratch tool).
usage: spill_sites.py <file.s> <substring of the mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_ZN6coflux\w+:)', s)
for f in funcs:
    name = f.split(':')[0]
    if sys.argv[2] not in name: continue
    lines = f.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    print(name, len(lines), 'lines')
    for i, l in enumerate(lines):
        if 'scratch_' in l:
            inl = [(a, b) for a, b in loops if a <= i <= b]
            print(i, l.strip()[:70], '| innermost loop', min((b - a for a, b in inl), default=None))
    big = sorted(loops, key=lambda ab: ab[0] - ab[1])[:12]
    for a, b in sorted(big):
        print('loop', a, b, 'valu', sum(1 for l in lines[a:b] if re.match(r'\s+v_', l)), 'lds', sum(1 for l in lines[a:b] if re.match(r'\s+ds_', l)))
