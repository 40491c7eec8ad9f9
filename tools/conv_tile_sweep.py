"""Conv path, bf16: every unfused 3x3 layer with each of the three tile shapes (PSM_UNET_FORCE=layer:arrangement:channel tiles:split), one at a
time, against the planner's choice -- the dispatch-stamped time of the layer and of its consumer, per shape.
    python tools/conv_tile_sweep.py [size=256] [cases=8]
(The plan-time autotuner tries the same candidates but keeps one only when the WHOLE pass gets more than 1 % faster.)"""
import os
import subprocess
import sys
size = sys.argv[1] if len(sys.argv) > 1 else "256"
cases = sys.argv[2] if len(sys.argv) > 2 else "8"
names = ['enc0a', 'enc0b', 'enc1a', 'enc1b', 'enc2a', 'enc2b', 'enc3a', 'enc3b', 'enc4a', 'enc4b', 'dec3a', 'dec3b', 'dec2a', 'dec2b', 'dec1a', 'dec1b', 'dec0a', 'dec0b', 'head']


def run(force):
    env = dict(os.environ)
    if force:
        env["PSM_UNET_FORCE"] = force
    out = subprocess.run([sys.executable, "tools/unet_layers.py", size, cases, "bf16"], env=env, capture_output=True, text=True, timeout=300).stdout
    t, plan = {}, {}
    for l in out.splitlines():
        p = l.split()
        if len(p) > 4 and p[0].split("+")[0] in names and "plan=" in l:
            t[p[0]] = float(l.split("]")[1].split()[0]); plan[p[0]] = l.split("plan=")[1].split("]")[0] + "]"
    tot = [float(l.split("sum of launches")[1].split()[0]) for l in out.splitlines() if "sum of launches" in l]
    return t, plan, (tot[0] if tot else float("nan"))


base, bplan, btot = run(None)
print(f"planner: sum of launches {btot:.1f} us")
for li in range(4, 14):
    nm = names[li]
    if nm not in base:
        continue
    nxt = names[li + 1] if names[li + 1] in base else None
    row = f"{nm:6s} planner {bplan[nm]:14s} {base[nm]:6.2f}" + (f" (+ {nxt} {base[nxt]:5.2f})" if nxt else "")
    for arr, nct in ((0, 2), (0, 1), (1, 4)):
        t, plan, tot = run(f"{li}:{arr}:{nct}:1")
        if nm in t:
            row += f" | {plan[nm]:14s} {t[nm]:6.2f}" + (f" (+{t[nxt]:5.2f})" if nxt and nxt in t else "") + f" sum {tot:6.1f}"
    print(row, flush=True)
