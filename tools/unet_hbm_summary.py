"""Condense tools/unet_hbm.sh TAG: per conv kernel and grid (= layer shape), memory-side bytes per launch
(2*FETCH_SIZE + WRITE_SIZE, KB counters; FETCH doubled per the gfx950 note of MI355X_MICROARCH.md for 16-byte-per-lane
streams) and the achieved GB/s against the un-profiled average duration of the same launches."""
import csv, glob, os, sys, collections
tag = sys.argv[1]; wl = sys.argv[2] if len(sys.argv) > 2 else "unet8_bf16"; O = f"gpurun_out/uh_{tag}"
def last(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]
def counter(d, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(last(f"{O}/{d}/*/*counter_collection.csv"))):
        if r["Counter_Name"] != name or "psm_" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"], r["Grid_Size"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: s / n for k, (s, n) in acc.items()}
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(last(f"{O}/trace/*/*kernel_trace.csv"))):
    if "psm_" not in r["Kernel_Name"]: continue
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    k = (r["Kernel_Name"], str(g))
    dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[k][1] += 1
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
with open(f"profiles/{tag}_{wl}_hbm.csv", "w") as f:
    f.write("kernel,grid_threads,launches,avg_us,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes(2*FETCH+WRITE),GB_per_s,frac_of_8TBps\n")
    for k in sorted(fetch, key=lambda k: (k[0], int(k[1]))):
        if k not in dur: continue
        us = dur[k][0] / dur[k][1] / 1e3
        hb = (2 * fetch[k] + write.get(k, 0.0)) * 1024.0
        gbs = hb / (us * 1e-6) / 1e9
        f.write('"%s",%s,%d,%.2f,%.1f,%.1f,%.0f,%.0f,%.3f\n' % (k[0][:96], k[1], dur[k][1], us, fetch[k], write.get(k, 0.0), hb, gbs, gbs / 8000))
        print(f"{k[0][40:92]:52s} grid={k[1]:>9s} {us:7.1f} us  fetch {fetch[k]/1024:7.1f} MB(x2) write {write.get(k,0)/1024:7.1f} MB  -> {gbs:6.0f} GB/s")
