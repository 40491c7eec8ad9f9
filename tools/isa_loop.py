"""Control-flow skeleton and per-region instruction mix of ONE kernel in a hipcc -S listing:
    python tools/isa_loop.py FILE.s KERNEL_SUBSTRING [first last]
prints every label, branch, barrier, wait on vmcnt and LDS-DMA / global instruction with its instruction index; with `first last` the mix of
that index range (e.g. the tile loop of a pair kernel: from its loop header to the back edge)."""
import sys


def body(path, key):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if key in l and l[:1] not in '.; \t' and ':' in l.split(';')[0]][0]
    out = []
    for l in lines[start + 1:]:
        s = l.strip()
        if not s or s.startswith(';') or (s.startswith('.') and not s.startswith('.LBB')):
            continue
        out.append(s)
        if s.startswith('s_endpgm'):
            break
    return out


def main():
    ins = body(sys.argv[1], sys.argv[2])
    rng = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
    idx = 0
    cnt = dict(mfma=0, valu=0, salu=0, smem=0, lds=0, vmem=0, wait=0, branch=0)
    for s in ins:
        if s.endswith(':'):
            if not rng:
                print(f"{idx:5d} {s[:90]}")
            continue
        op = s.split()[0]
        if not rng and (op.startswith(('s_cbranch', 's_branch', 's_barrier', 'global_', 'buffer_', 's_load')) or 'vmcnt' in s):
            print(f"{idx:5d}   {s[:90]}")
        if rng and rng[0] <= idx <= rng[1]:
            k = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
                 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else
                 'wait' if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep', 's_setprio')) else
                 'smem' if op.startswith(('s_load', 's_buffer_load')) else
                 'branch' if op.startswith(('s_cbranch', 's_branch', 's_endpgm')) else 'salu')
            cnt[k] += 1
        idx += 1
    if rng:
        other = sum(v for k, v in cnt.items() if k != 'mfma')
        print(f"instructions {rng[0]}..{rng[1]}: " + " ".join(f"{k} {v}" for k, v in cnt.items()) +
              f"   non-MFMA : MFMA = {other / max(cnt['mfma'], 1):.2f}")


if __name__ == '__main__':
    main()
