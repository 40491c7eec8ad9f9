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

# the form bench.py reads for the conv legs' roofline.traffic (committed_unet_traffic): launches keyed like bench.unet_launch_key,
# guarded by a hash of the conv sources that were profiled; one file for all workloads
import json, re
sys.path.insert(0, ".")
import bench as _bench
def _key(name, grid):
    m = re.search(r"psm_conv3x3_kernel<(\d+), (\d+), (\d+), (\d+), (-?\d+), ", name)
    if m:
        return f"conv3x3|{m.group(1)}|{m.group(3)}|{m.group(5)}|{grid}"
    m = re.search(r"(psm_pair\w*<[^(]*>)\(", name)
    return "pair|" + m.group(1).replace(" ", "") if m else None
pj = os.path.join(".", _bench.UNET_PMC_FILE)
try:
    d = json.load(open(pj))
except Exception:
    d = {}
if d.get("unet_source_hash") != _bench.unet_source_hash():
    d = {"unet_source_hash": _bench.unet_source_hash(), "profile_tag": tag,
         "unit": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes with --kernel-trace only", "workloads": {}}
w, wn = {}, {}
for k in fetch:
    kk = _key(k[0], k[1])
    # several instantiations can share a key (the plan-time autotuner tries other splits / buffer counts on the same grid): the one
    # with the most launches is the planned layer
    if kk and k in dur and dur[k][1] > wn.get(kk, 0):
        w[kk] = (2 * fetch[k] + write.get(k, 0.0)) * 1024.0
        wn[kk] = dur[k][1]
d["workloads"][wl] = w
json.dump(d, open(pj, "w"), indent=1)
print("wrote", pj, len(w), "launch keys for", wl)
