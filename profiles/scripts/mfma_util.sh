#!/bin/bash
# MFMA utilisation of the encoder kernels at batch 64 (rocprofv3 PMC pass; see profiles/README.md)
OUT=$PWD/gpurun_out/mfma
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/profiles/enc_driver.py 64 > $OUT/raw.log 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/derived -- python3 $GRAFT_REPO_ROOT/profiles/enc_driver.py 64 > $OUT/derived.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "mfma")
for sub in ("raw", "derived"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "MfmaUtil"): cnt[k] += 1
    with open(os.path.join(out, sub + "_summary.txt"), "w") as fo:
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", acc[k].get("MfmaUtil", 0))):
            line = k + " | n=%d | " % cnt[k] + " ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items()))
            print(line); fo.write(line + "\n")
PY
rm -rf $OUT/raw $OUT/derived
