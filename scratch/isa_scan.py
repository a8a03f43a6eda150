"""Scan one kernel of an ISA listing: barriers, MFMAs, scratch traffic, branches -- where are the spills relative to the loops?
   python scratch/isa_scan.py /tmp/conv_slab.s <mangled-name-substring> [--dump a:b]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for k, (i, n) in enumerate(starts):
    if key in n:
        j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = lines[i:j]
        break
else:
    raise SystemExit("kernel not found")
print(n, len(body), "lines")
if len(sys.argv) > 4 and sys.argv[3] == "--dump":
    a, b = (int(x) for x in sys.argv[4].split(':'))
    print('\n'.join(f"{q:6d} {body[q]}" for q in range(a, b)))
    raise SystemExit
labels = {m.group(1): q for q, l in enumerate(body) if (m := re.match(r'^(\.LBB\w+):', l))}
ev = []
for q, l in enumerate(body):
    t = l.strip()
    if t.startswith('s_barrier'): ev.append((q, 'BAR'))
    elif t.startswith('v_mfma'): ev.append((q, 'M'))
    elif t.startswith('scratch_'): ev.append((q, 'SCR ' + t.split()[0]))
    elif t.startswith('s_cbranch') or t.startswith('s_branch'):
        tgt = t.split()[-1]
        ev.append((q, f"BR {t.split()[0]} -> {labels.get(tgt, '?')}"))
    elif 'global_load_lds' in t: ev.append((q, 'DMA'))
    elif t.startswith('ds_read'): ev.append((q, 'R'))
    elif t.startswith('s_waitcnt'): ev.append((q, 'W ' + ' '.join(t.split()[1:3])))
# compress runs
out, last, cnt, q0 = [], None, 0, 0
for q, e in ev:
    if e == last and e in ('M', 'R', 'DMA'):
        cnt += 1
    else:
        if last is not None: out.append((q0, last, cnt))
        last, cnt, q0 = e, 1, q
out.append((q0, last, cnt))
for q, e, c in out:
    print(f"{q:6d} {e}{' x' + str(c) if c > 1 else ''}")
