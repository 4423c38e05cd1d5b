#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + separate PMC passes of the SAME
# bench command; raw output under gpurun_out/, condensed summaries for profiles/.
# usage: tools/collect_profiles.sh r01
set -euo pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_misc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_misc.err
cd $R
python3 tools/summarise_profile.py stats $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
python3 tools/summarise_profile.py pmc $O/${TAG}_pmc.json $(ls $O/pmc_*/*/*counter_collection.csv)
cp $O/bench_under_rocprof.json $O/${TAG}_bench_under_rocprof.json
head -8 $O/${TAG}_kernel_stats.csv
