#!/bin/bash
# usage: tools/lab/build_variant.sh NAME "file1.hip file2.hip" "-DFLAG ..." ['sed expression']  -> tools/lab/libs/NAME.so
# One or more translation units rebuilt with extra flags and linked against the product's other objects (csrc/.obj, run
# csrc/build.sh first).  Load with tools/lab/run_with_lib.py libs/NAME.so <script>.
# The product sources carry NO compile-time ablation switches (round-4 hygiene): a variant PATCHES A COPY -- the optional sed
# expression is applied to copies of the named files (and of the headers they include) in tools/lab/libs/src_NAME before
# compiling, e.g.  's/if (nxt < cend) issue(nxt, buf ^ 1);/if (nxt < cend \&\& nxt < cbeg + 4 * step) issue(nxt, buf ^ 1);/'
# for the compute-only timing of the fused multi-vector pass.
set -euo pipefail
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/linearresponsevariationalbayes.py_amd/csrc
NAME=$1; FILES=$2; EXTRA=$3; PATCH=${4:-}
O=$R/tools/lab/libs; mkdir -p $O/obj_$NAME
SRC=$C
if [ -n "$PATCH" ]; then
    SRC=$O/src_$NAME; rm -rf $SRC; mkdir -p $SRC/../../include_dummy
    cp $C/*.hip $C/*.h $SRC/
    sed -i 's#"../../include/lrvb_hip.h"#"'$R'/include/lrvb_hip.h"#' $SRC/lrvb_internal.h
    for f in $FILES; do sed -i -e "$PATCH" $SRC/$f; done
    sed -i -e "$PATCH" $SRC/*.h
fi
OBJS=""
for o in $C/.obj/*.o; do
    b=$(basename $o .o); skip=0
    for f in $FILES; do [ "$b" = "${f%.hip}" ] && skip=1; done
    [ $skip -eq 0 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -I$SRC $EXTRA -c $SRC/$f -o $O/obj_$NAME/${f%.hip}.o &
done
wait
for f in $FILES; do OBJS="$OBJS $O/obj_$NAME/${f%.hip}.o"; done
hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -Wl,--exclude-libs,ALL -Wl,--version-script=$C/exports.map $OBJS -o $O/$NAME.so
echo built $O/$NAME.so
