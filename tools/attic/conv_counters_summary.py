"""Condense tools/attic/conv_counters.sh TAG: per kernel, the SQ counters summed over the chip and averaged over launches."""
import csv, glob, os, sys
tag = sys.argv[1]
rows = {}
names = []
for p in ("p1", "p2"):
    fs = sorted(glob.glob(f"gpurun_out/cc_{tag}/{p}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        if "psm_" not in k: continue
        c = r["Counter_Name"]
        if c not in names: names.append(c)
        d = rows.setdefault(k, {}).setdefault(c, [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
with open(f"gpurun_out/cc_{tag}/summary.csv", "w") as out:
    out.write("kernel,launches," + ",".join(names) + "\n")
    for k, d in sorted(rows.items()):
        n = max(v[1] for v in d.values())
        out.write('"%s",%d,' % (k[:100], n) + ",".join("%.0f" % (d[c][0] / d[c][1]) if c in d else "" for c in names) + "\n")
        wc = d.get("SQ_WAVE_CYCLES", [0, 1]); wc = wc[0] / wc[1]
        if wc:
            f = lambda c: 100.0 * d[c][0] / d[c][1] / wc if c in d else float("nan")
            print(f"{k[40:100]:60s} n={n:4d} wait_any={f('SQ_WAIT_ANY'):5.1f}% wait_inst={f('SQ_WAIT_INST_ANY'):5.1f}% (lds {f('SQ_WAIT_INST_LDS'):4.1f}%) active={f('SQ_ACTIVE_INST_ANY'):5.1f}%  mfma_busy/gui/1024={100.0*d['SQ_VALU_MFMA_BUSY_CYCLES'][0]/d['SQ_VALU_MFMA_BUSY_CYCLES'][1]/(d['GRBM_GUI_ACTIVE'][0]/d['GRBM_GUI_ACTIVE'][1])/1024:5.1f}%")
