#!/bin/bash
# usage (through gpurun): tools/lab/pmc.sh TAG KERNEL_SUBSTRING script.py [args...]
# PMC_SETS='A B;C D' overrides the counter sets (one pass per ';'-separated set)
# separate --pmc passes (no trace domains beside them); prints the mean of every counter over the dispatches of the kernel
set -uo pipefail
TAG=$1; SUB=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
IFS=';' read -ra SETS <<< "${PMC_SETS:-FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY;GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}"
for set in "${SETS[@]}"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 ${PMC_PASS_TIMEOUT:-150} rocprofv3 --pmc $set --output-format csv -d $O/pmc_${TAG}_$i -- python3 $R/"$@" > /dev/null 2> $O/pmc_${TAG}_$i.err
done
if false; then
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_${TAG}_$i -- python3 $R/"$@" > /dev/null 2> $O/pmc_${TAG}_$i.err
done
fi
cd $R
python3 - "$O" "$TAG" "$SUB" <<'PY'
import csv, glob, collections, sys, json
O, TAG, SUB = sys.argv[1:4]
d = collections.defaultdict(list)
for f in glob.glob('%s/pmc_%s_*/*/*counter_collection.csv' % (O, TAG)):
    for r in csv.DictReader(open(f)):
        if SUB in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: sum(v) / len(v) for k, v in sorted(d.items())}
out['dispatches'] = {k: len(v) for k, v in d.items()}
json.dump(out, open('%s/pmc_%s.json' % (O, TAG), 'w'), indent=1)
for k, v in out.items():
    if k != 'dispatches': print('%-32s %.4e' % (k, v))
PY
