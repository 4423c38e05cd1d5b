#!/bin/bash
# Builds liblrvb_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../liblrvb_hip.so
SRCS="lrvb_api.hip k_wsyrk.hip k_glm.hip k_pack.hip k_linalg.hip k_finish.hip k_mixture.hip k_hvp_multi.hip"
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
     -Wall -Wno-unused-function -Wno-unused-variable \
     ${LRVB_HIPCC_EXTRA:-} $SRCS -o "$OUT"
