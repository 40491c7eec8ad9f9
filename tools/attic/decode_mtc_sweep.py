"""Batch decode + paste: row tiles per chunk (PSM_DECODE_MTC, diagnostic) against the launcher's choice, for several batch
sizes and both field counts.  One process per setting (the knob is read once)."""
import os, subprocess, sys
for variant, cases in (("deltas", 8), ("deltas", 16), ("deltas", 32), ("deltas", 64), ("gradp", 4), ("gradp", 8), ("gradp", 16)):
    for mtc in ("0", "1", "2"):
        env = dict(os.environ, PSM_DECODE_MTC=mtc, PSM_KT_CASES=str(cases))
        wl = "config3" if variant == "deltas" else "config1"
        out = subprocess.run([sys.executable, "tools/kernel_times.py", wl, "200"], env=env, capture_output=True, text=True).stdout
        dec = [l for l in out.splitlines() if "decode_paste" in l]
        b2b = [l for l in out.splitlines() if "back-to-back" in l]
        print(variant, cases, "MTC", mtc, dec[0].split()[0][-22:] if dec else "?", dec[0].split()[-4] if dec else "?", "us |", b2b[0] if b2b else out[-200:], flush=True)
