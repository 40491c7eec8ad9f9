"""Batch decode + paste: workgroup target of the row-group spread (PSM_DECODE_WGS, diagnostic), one 32-row tile per chunk."""
import os, subprocess, sys
for variant, cases in (("deltas", 8), ("deltas", 16), ("deltas", 64), ("gradp", 8)):
    for wgs in ("256", "512", "1024", "2048"):
        env = dict(os.environ, PSM_DECODE_WGS=wgs, PSM_KT_CASES=str(cases))
        wl = "config3" if variant == "deltas" else "config1"
        out = subprocess.run([sys.executable, "tools/kernel_times.py", wl, "200"], env=env, capture_output=True, text=True).stdout
        dec = [l for l in out.splitlines() if "decode_paste" in l]
        b2b = [l for l in out.splitlines() if "back-to-back" in l]
        print(variant, cases, "WGS", wgs, dec[0].split()[-4] if dec else "?", "us |", b2b[0] if b2b else out[-200:], flush=True)
