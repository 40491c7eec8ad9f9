#!/bin/bash
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/enc_pmc; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/configs_one.py 64 > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/configs_one.py 64 > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/enc_pmc/{p}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "encode_x6_mt" not in r["Kernel_Name"]: continue
            d = acc[r["Counter_Name"]]; d[0] += float(r["Counter_Value"]); d[1] += 1
    for k, (t, n) in sorted(acc.items()): print(p, k, n, t / max(n, 1))
    ks = glob.glob(f"gpurun_out/enc_pmc/{p}/*/*kernel_trace.csv")
    for f in ks:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "encode_x6_mt" in r["Kernel_Name"]]
        if d: print(p, "duration ns median", sorted(d)[len(d)//2], "n", len(d))
PY
rm -rf $O/p1 $O/p2
