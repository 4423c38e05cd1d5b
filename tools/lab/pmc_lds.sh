#!/bin/bash
# usage (through gpurun): tools/lab/pmc_lds.sh TAG KERNEL_SUBSTRING script.py [args...]  -- LDS counters only
set -uo pipefail
TAG=$1; SUB=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmcl_${TAG} -- python3 $R/"$@" > /dev/null 2> $O/pmcl_${TAG}.err
cd $R
python3 - "$O" "$TAG" "$SUB" <<'PY'
import csv, glob, collections, sys
O, TAG, SUB = sys.argv[1:4]
d = collections.defaultdict(list)
for f in glob.glob('%s/pmcl_%s/*/*counter_collection.csv' % (O, TAG)):
    for r in csv.DictReader(open(f)):
        if SUB in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()):
    print('%s %-24s %.4e (%d dispatches)' % (TAG, k, sum(v) / len(v), len(v)))
PY
