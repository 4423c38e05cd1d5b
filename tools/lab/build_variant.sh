#!/bin/bash
# usage: tools/lab/build_variant.sh NAME "file1.hip file2.hip" "-DFLAG ..."  -> tools/lab/libs/NAME.so
# One or more translation units rebuilt with extra flags and linked against the product's other objects (csrc/.obj, run
# csrc/build.sh first).  Load with tools/lab/run_with_lib.py libs/NAME.so <script>.
set -euo pipefail
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/linearresponsevariationalbayes.py_amd/csrc
NAME=$1; FILES=$2; EXTRA=$3
O=$R/tools/lab/libs; mkdir -p $O/obj_$NAME
OBJS=""
for o in $C/.obj/*.o; do
    b=$(basename $o .o); skip=0
    for f in $FILES; do [ "$b" = "${f%.hip}" ] && skip=1; done
    [ $skip -eq 0 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -I$C $EXTRA -c $C/$f -o $O/obj_$NAME/${f%.hip}.o &
done
wait
for f in $FILES; do OBJS="$OBJS $O/obj_$NAME/${f%.hip}.o"; done
hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -Wl,--exclude-libs,ALL -Wl,--version-script=$C/exports.map $OBJS -o $O/$NAME.so
echo built $O/$NAME.so
