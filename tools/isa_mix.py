"""Instruction mix per kernel of a hipcc -S listing (gfx950): MFMA, other vector, scalar, LDS, global, waits / barriers.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S FILE.hip -o FILE.s;  python tools/isa_mix.py FILE.s [name-filter]
Counts are static (whole kernel body; the pair kernels' tile loops are straight-line code, so body ~ one tile + a short prologue).
The figure the round-4 verdict asks for is (everything that is not an MFMA) : MFMA."""
import re
import sys


def mix(path, flt=None):
    rows = []
    name, cnt = None, None
    for line in open(path):
        m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', line)
        if m and not line.startswith('.'):
            if m.group(1).startswith('_Z') or m.group(1).startswith('psm_'):
                if name and cnt and cnt['mfma'] + cnt['valu'] > 0:
                    rows.append((name, cnt))
                name, cnt = m.group(1), dict(mfma=0, valu=0, salu=0, smem=0, lds=0, vmem=0, wait=0, branch=0)
                continue
        if name is None:
            continue
        s = line.strip()
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
            continue
        op = s.split()[0]
        if op.startswith('v_mfma'):
            cnt['mfma'] += 1
        elif op.startswith('v_'):
            cnt['valu'] += 1
        elif op.startswith('ds_'):
            cnt['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            cnt['vmem'] += 1
        elif op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep', 's_setprio')):
            cnt['wait'] += 1
        elif op.startswith(('s_load', 's_buffer_load')):
            cnt['smem'] += 1
        elif op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc')):
            cnt['branch'] += 1
        elif op.startswith('s_'):
            cnt['salu'] += 1
        if op == 's_endpgm' and cnt['mfma'] + cnt['valu'] > 0:
            rows.append((name, cnt))
            name, cnt = None, None
    out = []
    for name, c in rows:
        if flt and flt not in name:
            continue
        other = c['valu'] + c['salu'] + c['smem'] + c['lds'] + c['vmem'] + c['wait'] + c['branch']
        ratio = other / c['mfma'] if c['mfma'] else float('nan')
        out.append(f"{name[:70]:70s} mfma {c['mfma']:5d} valu {c['valu']:5d} salu {c['salu']:5d} smem {c['smem']:3d} lds {c['lds']:4d} vmem {c['vmem']:4d} "
                   f"wait {c['wait']:4d} br {c['branch']:3d}  non-MFMA:MFMA {ratio:5.2f}  (vector+lds+vmem):MFMA {(c['valu'] + c['lds'] + c['vmem']) / max(c['mfma'], 1):5.2f}")
    return out


if __name__ == '__main__':
    for l in mix(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None):
        print(l)
