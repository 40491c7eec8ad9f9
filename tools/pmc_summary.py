"""Condense the rocprofv3 outputs of tools/prof_full.sh TAG into the files kept under profiles/:
profiles/<TAG>_kernel_stats.csv, <TAG>_kernel_quantiles.csv, <TAG>_bench.json.log, <TAG>_pmc_summary.csv, <TAG>_pmc.json (what bench.py reads for roofline.traffic).
HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB counters; FETCH doubled per the gfx950 note in
MI355X_MICROARCH.md's HBM section for 16-B/lane coalesced streams)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
O = f"gpurun_out/full_{tag}"
def counter(dirname, name):
    f = sorted(glob.glob(f"{O}/{dirname}/*/*counter_collection.csv"))[-1]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        s, n = acc.get(k, (0.0, 0))
        acc[k] = (s + float(r["Counter_Value"]), n + 1)
    return {k: s / n for k, (s, n) in acc.items()}
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
rows, allk = [], {}
for k in sorted(fetch):
    if "psm_" not in k: continue
    hb = (2 * fetch[k] + write.get(k, 0.0)) * 1024.0
    rows.append((k, fetch[k], write.get(k, 0.0), hb))
    allk[k[:40]] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write.get(k, 0.0), "hbm_bytes": hb}
os.makedirs("profiles", exist_ok=True)
with open(f"profiles/{tag}_pmc_summary.csv", "w") as f:
    f.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
    for r in rows: f.write('"%s",%.3f,%.3f,%.1f\n' % r)
# the form bench.py reads for roofline.traffic: kernel names as psm_time_kernels reports them, guarded by a hash of the
# kernel sources that were profiled
sys.path.insert(0, ".")
import bench as _bench
def _short(k):
    k = k.replace("void ", "")
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: return k[:i]
    return k
json.dump({"workload": "config1", "kernel_source_hash": _bench.kernel_source_hash(), "profile_tag": tag,
           "unit": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes with --kernel-trace only",
           "kernels": {_short(r[0]): r[3] for r in rows}}, open(_bench.PMC_FILE, "w"), indent=1)
st = sorted(glob.glob(f"{O}/stats/*/*kernel_stats.csv"))[-1]
shutil.copy(st, f"profiles/{tag}_kernel_stats.csv")
# per-kernel median / p10 / p90 from the same trace (bench.py's roofline uses medians; --stats reports means)
import subprocess
subprocess.run([sys.executable, "tools/trace_quantiles.py", f"{O}/stats", f"profiles/{tag}_kernel_quantiles.csv"], check=False, stdout=subprocess.DEVNULL)
if glob.glob(f"{O}/unet_stats/*/*kernel_trace.csv"):
    subprocess.run([sys.executable, "tools/trace_quantiles.py", f"{O}/unet_stats", f"profiles/{tag}_unet8_bf16_kernel_quantiles.csv"], check=False, stdout=subprocess.DEVNULL)
open(f"profiles/{tag}_bench.json.log", "w").write([l for l in open(f"{O}/bench.log").read().strip().splitlines() if l.startswith("{")][-1] + "\n")
if os.path.exists(f"{O}/detail_bench/bench_detail.json"):
    shutil.copy(f"{O}/detail_bench/bench_detail.json", f"profiles/{tag}_bench_detail.json")
for r in list(csv.DictReader(open(st)))[:10]:
    print(f"{r['Name'][:60]:60s} calls={int(r['Calls']):6d} avg_us={float(r['AverageNs'])/1e3:8.2f}")

# convolutional path
import glob as _g
us = sorted(_g.glob(f"{O}/unet_stats/*/*kernel_stats.csv"))
if us:
    shutil.copy(us[-1], f"profiles/{tag}_unet8_bf16_kernel_stats.csv")
for name in ("bench_unet", "bench_unet8", "bench_unet8_bf16"):
    p = f"{O}/{name}.log"
    if os.path.exists(p):
        lines = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
        if lines:
            open(f"profiles/{tag}_{name}.json.log", "w").write(lines[-1] + "\n")
        dd = f"{O}/detail_{name[6:]}/bench_detail.json"
        if os.path.exists(dd):
            shutil.copy(dd, f"profiles/{tag}_{name}_detail.json")
