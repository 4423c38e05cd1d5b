#!/bin/bash
# usage: pmc.sh <lib.so> : FETCH_SIZE of the SYRK kernel with a lab library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_$1 -- python3 $R/tools/lab/run_with_lib.py $1 syrk_only.py 0 > /dev/null 2> $R/gpurun_out/pmc_$1.err
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$1/*/*counter_collection.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE": d[r["Kernel_Name"][:24]].append(float(r["Counter_Value"]))
for k,v in d.items():
    if "wsyrk_u8" in k or "wsyrk_glds" in k: print("$1", k, len(v), "FETCH KiB %.3e" % (sum(v)/len(v)))
PY
