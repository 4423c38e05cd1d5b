#!/bin/bash
# usage (through gpurun): tools/lab/prof.sh TAG script.py [args...]  -> gpurun_out/r3/TAG.log, gpurun_out/r3/TAG_stats.csv (kernel summary)
set -uo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/"$@" > $O/$TAG.log 2> $O/$TAG.err
cd $R
tail -5 $O/$TAG.log
python3 tools/summarise_profile.py stats $(ls $O/prof_$TAG/*/*kernel_stats.csv | head -1) $O/${TAG}_stats.csv && head -${HEAD:-14} $O/${TAG}_stats.csv
