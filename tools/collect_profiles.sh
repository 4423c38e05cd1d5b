#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + separate PMC passes of the SAME
# bench command; raw output under gpurun_out/, condensed summaries for profiles/.
# usage: tools/collect_profiles.sh r04 [headline|configs|all]
set -euo pipefail
TAG=${1:-r04}
WHAT=${2:-all}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs"
if [ "$WHAT" = headline ] || [ "$WHAT" = all ]; then
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/bench_under_rocprof.json 2> $O/stats.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2> $O/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2> $O/pmc_write.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2> $O/pmc_sq.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_misc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2> $O/pmc_misc.err
  cd $R
  python3 tools/summarise_profile.py stats $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
  python3 tools/summarise_profile.py pmc $O/${TAG}_pmc.json $(ls $O/pmc_*/*/*counter_collection.csv)
  cp $O/bench_under_rocprof.json $O/${TAG}_bench_under_rocprof.json
  head -8 $O/${TAG}_kernel_stats.csv
fi
if [ "$WHAT" = configs ] || [ "$WHAT" = all ]; then
  # the other BASELINE.json configurations: the bench line of each, and kernel stats of the two whose steps changed this round
  cd $R
  for c in c2 c3 c4; do python3 bench.py --config $c --steps 20 --warmup 5 > $O/${TAG}_bench_$c.json 2> $O/bench_$c.err; done
  python3 bench.py --config c5 --steps 3 --warmup 1 > $O/${TAG}_bench_c5.json 2> $O/bench_c5.err
  cd /tmp
  for c in c2 c3 c4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 $R/bench.py --config $c --steps 10 --warmup 3 > /dev/null 2> $O/stats_$c.err
    python3 $R/tools/summarise_profile.py stats $(ls $O/stats_$c/*/*kernel_stats.csv | head -1) $O/${TAG}_${c}_kernel_stats.csv
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --config c5 --steps 2 --warmup 1 > /dev/null 2> $O/stats_c5.err
  python3 $R/tools/summarise_profile.py stats $(ls $O/stats_c5/*/*kernel_stats.csv | head -1) $O/${TAG}_c5_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lmm -- python3 $R/tools/time_lmm_stats.py > $O/lmm_stats.log 2> $O/stats_lmm.err
  python3 $R/tools/summarise_profile.py stats $(ls $O/stats_lmm/*/*kernel_stats.csv | head -1) $O/${TAG}_lmm_stats_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cg -- python3 $R/tools/time_hvp_multi.py > $O/cg.log 2> $O/stats_cg.err
  python3 $R/tools/summarise_profile.py stats $(ls $O/stats_cg/*/*kernel_stats.csv | head -1) $O/${TAG}_cg_kernel_stats.csv
  cd $R
  head -6 $O/${TAG}_c4_kernel_stats.csv
fi
