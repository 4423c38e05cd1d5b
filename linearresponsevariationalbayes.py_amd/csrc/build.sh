#!/bin/bash
# Builds liblrvb_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Translation units are compiled in parallel (the per-row mixture kernels are instantiated for
# K = 2 .. 32, split over four units), then linked into one shared library.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../liblrvb_hip.so
SRCS="lrvb_api.hip k_elementwise.hip k_gauss.hip k_hvec.hip k_models.hip k_cg.hip k_wsyrk.hip k_glm.hip k_pack.hip k_linalg.hip k_finish.hip k_hyper.hip k_mixture.hip k_hvp_multi.hip k_lmm.hip \
      k_mixture_inst0.hip k_mixture_inst1.hip k_mixture_inst2.hip k_mixture_inst3.hip"
OBJDIR=.obj
mkdir -p "$OBJDIR"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function -Wno-unused-variable ${LRVB_HIPCC_EXTRA:-}"
pids=()
for f in $SRCS; do
    extra=""
    # k_linalg.hip: the fully unrolled 64 x 64 diagonal-block kernel sends LLVM's CodeGenPrepare pass quadratic
    # (96 of 100 s); without that pass the object code has the same register count and speed
    [ "$f" = "k_linalg.hip" ] && extra="-mllvm -disable-cgp"
    hipcc $FLAGS $extra -c "$f" -o "$OBJDIR/${f%.hip}.o" &
    pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" -eq 0 ] || { echo "compilation failed" >&2; exit 1; }
OBJS=""
for f in $SRCS; do OBJS="$OBJS $OBJDIR/${f%.hip}.o"; done
exec hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -Wl,--exclude-libs,ALL -Wl,--version-script=exports.map $OBJS -o "$OUT"
