import os, subprocess, sys
names = ["encode","reduce","mlp","decode","strips","chain","paste"]
def run(keep, extra=None):
    mask = 0
    for k, n in enumerate(names):
        if n not in keep: mask |= 1 << k
    env = dict(os.environ, PSM_DEBUG_SKIP=str(mask)); env.update(extra or {})
    out = subprocess.run([sys.executable, "tools/attic/hostbound.py"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return float(out.split("total")[1].split()[0])
seq = ["encode","mlp","decode","strips","paste"]
prev = 0.0
for i in range(1, len(seq) + 1):
    t = run(seq[:i]); print(f"prefix {'+'.join(seq[:i]):40s} {t:6.1f}  (+{t-prev:5.1f})"); prev = t
for combo in (["mlp"], ["decode"], ["strips"], ["paste"], ["decode","strips"], ["strips","paste"], ["decode","paste"], ["encode","decode"]):
    print(f"only {'+'.join(combo):30s} {run(combo):6.1f}")
print("unfused prefix encode+reduce+mlp:", run(["encode","reduce","mlp"], {"PSM_NO_FUSED_REDUCE":"1"}))
