#!/bin/bash
# Developer tool: VALU/SALU instruction counts of the parse kernel under the ablation knobs.
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for ab in 96 98 100 104 97 112; do
  rm -rf $R/gpurun_out/pmc_ab_$ab
  VG_LZ_ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ab_$ab -- python $R/tools/task_times.py 100 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_ab_$ab/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(float); n=0
for r in csv.DictReader(open(f)):
    if "k_lz_parse" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n+= r["Counter_Name"]=="SQ_INSTS_VALU"
print("ablate", $ab, {c: round(x/max(n,1)/1e6,1) for c,x in acc.items()})
PY
done
