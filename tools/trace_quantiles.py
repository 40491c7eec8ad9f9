"""Per-kernel median / p10 / p90 of the dispatch durations in a rocprofv3 --kernel-trace output directory:
    python tools/trace_quantiles.py ROCPROF_DIR OUT.csv [warmup_fraction]
One row per (kernel, grid size) -- the same template instantiation launched on two grids is two layers.  The first
`warmup_fraction` (default 0.1) of every row's dispatches are dropped (plan-time autotune, first-touch launches).  This is the
file the bench line's `roofline.frac` can be recomputed from: bench.py takes the MEDIAN of its own dispatch stamps, rocprofv3
--stats reports means (round-5 verdict, weak 9)."""
import csv, glob, os, sys, collections
src, out = sys.argv[1], sys.argv[2]
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit(f"no kernel_trace.csv under {src}")
rows = collections.defaultdict(list)
for r in csv.DictReader(open(files[-1])):
    if "psm_" not in r["Kernel_Name"]:
        continue
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    rows[(r["Kernel_Name"], g, wg)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
def q(v, f):
    return v[min(len(v) - 1, int(f * len(v)))]
with open(out, "w") as f:
    f.write("kernel,grid_threads,workgroup_threads,dispatches_used,median_us,p10_us,p90_us,mean_us,max_us\n")
    table = []
    for (k, g, wg), v in rows.items():
        v.sort()
        d = sorted(x[1] for x in v[int(skip * len(v)):])
        table.append((k, g, wg, len(d), q(d, 0.5) / 1e3, q(d, 0.1) / 1e3, q(d, 0.9) / 1e3, sum(d) / len(d) / 1e3, d[-1] / 1e3))
    for t in sorted(table, key=lambda t: -t[3] * t[4]):
        f.write('"%s",%d,%d,%d,%.3f,%.3f,%.3f,%.3f,%.3f\n' % t)
        print(f"{t[0][:70]:70s} grid={t[1]:>9d} n={t[3]:6d} median {t[4]:8.2f} p10 {t[5]:8.2f} p90 {t[6]:8.2f} mean {t[7]:8.2f} us")
