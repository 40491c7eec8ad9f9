"""Condense tools/mfma_util.sh TAG: per kernel, the derived counter MfmaUtil of rocprofv3
(= sum SQ_VALU_MFMA_BUSY_CYCLES / (max GRBM_GUI_ACTIVE * SIMD_NUM) * 100), averaged over the launches."""
import csv, glob, os, sys
tag = sys.argv[1]
out = open(f"profiles/{tag}_mfma_util.csv", "w")
out.write("run,kernel,grid_threads,launches,MfmaUtil_percent\n")
for run in ("unet8", "unet8b", "unet64b", "unet512b", "pca", "pca64"):
    fs = sorted(glob.glob(f"gpurun_out/mfma_{tag}/{run}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs: continue
    acc = {}
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        if "psm_" not in k or r["Counter_Name"] != "MfmaUtil": continue
        k = (k, r.get("Grid_Size", ""))                    # the same instantiation on two grids is two layers
        d = acc.setdefault(k, [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
    for (k, g), (tot, n) in sorted(acc.items()):
        out.write('%s,"%s",%s,%d,%.1f\n' % (run, k[:110], g, n, tot / n))
        print(f"{run:8s} {k[:70]:70s} grid={g:>9s} n={n:5d} MfmaUtil={tot/n:5.1f}%")
out.close()
