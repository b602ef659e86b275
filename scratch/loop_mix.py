"""Instruction mix of the biggest-VALU innermost loops of a kernel (scratch tool): loop_mix.py <file.s> <name substring>"""
import re, sys, collections
s = open(sys.argv[1]).read()
for f in re.split(r'\n(?=_ZN6coflux\w+:)', s):
    name = f.split(':')[0]
    if sys.argv[2] not in name: continue
    lines = f.split('\n')
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    def valu(a, b): return sum(1 for l in lines[a:b] if re.match(r'\s+v_', l))
    print(name)
    for a, b in sorted(loops):
        v = valu(a, b)
        if v < 100: continue
        c = collections.Counter()
        for l in lines[a:b]:
            m = re.match(r'\s+([vs]_\w+|ds_\w+|global_\w+|scratch_\w+|flat_\w+)', l)
            if m: c[m.group(1)] += 1
        f64 = sum(n for k, n in c.items() if re.search(r'_f64|v_mov_b64', k))
        print(f' loop {a}-{b}: valu {v} (f64-rate {f64}), lds {sum(n for k, n in c.items() if k.startswith("ds_"))}, flat/global {sum(n for k, n in c.items() if k.startswith(("flat_", "global_")))}')
        if len(sys.argv) > 3: print('   ', ', '.join(f'{n} {k}' for k, n in sorted(c.items(), key=lambda kv: -kv[1])[:30]))
