"""Finds the pattern that cost the case-batch decode 16 store round trips per chunk: an `s_waitcnt vmcnt(0)` INSIDE a predicated store block
(between s_cbranch_execz and the global_store of that block) -- the wait-count pass cannot count stores issued under run-time predicates, so the
first use of any register with a load still pending waits for every store in flight.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S FILE.hip -o FILE.s;  python tools/attic/isa_store_waits.py FILE.s"""
import re
import sys

name, lines = None, []
out = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        name, lines = m.group(1), []
        out[name] = lines
        continue
    if name:
        lines.append(line.strip())
for name, ls in out.items():
    hits = 0
    for i, l in enumerate(ls):
        if l.startswith('s_cbranch_execz'):
            blk = []
            for k in range(i + 1, min(i + 40, len(ls))):
                if ls[k].endswith(':') and ls[k].startswith('.LBB'):
                    break
                blk.append(ls[k])
            if any(b.startswith(('global_store', 'buffer_store')) for b in blk) and any(b.startswith('s_waitcnt vmcnt(0)') for b in blk) and len(blk) < 30:
                hits += 1
    if hits >= 3:
        print(f"{hits:4d}  {name[:110]}")
