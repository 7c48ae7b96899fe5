#!/usr/bin/env bash
# Host half of libehyb.so (partitioner, reorder, layout builder, plan cache, Matrix Market reader)
# under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU test suite.
# GPU sanitizers are not available on this pool; the HIP objects are linked as they are.
#
#   bash tools/asan_host.sh            # builds _ab/libehyb_asan.so and runs pytest -m "not gpu" against it
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/ehyb_spmv_gpu_amd/csrc"
OUT="$ROOT/_ab"
TMP="$(mktemp -d)"
mkdir -p "$OUT"
make -C "$SRC" -j8 >/dev/null                      # the HIP objects (build/ehyb_hip.o, ehyb_cg.o, ehyb_fill.o, er_panel_dev.o, ehyb_comm.o)
for f in common partition reorder layout er_panel plan plan_io matrix_io; do
    g++ -O1 -g -fPIC -fopenmp -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
        -I"$ROOT/include" -I"$SRC" -c "$SRC/$f.cpp" -o "$TMP/$f.o"
done
g++ -shared -o "$OUT/libehyb_asan.so" "$TMP"/*.o "$SRC/build/ehyb_hip.o" "$SRC/build/ehyb_cg.o" "$SRC/build/ehyb_fill.o" "$SRC/build/er_panel_dev.o" "$SRC/build/ehyb_comm.o" \
    -fsanitize=address,undefined -fopenmp -L/opt/rocm/lib -lamdhip64 -ldl -lz -Wl,-rpath,/opt/rocm/lib
rm -rf "$TMP"
cd "$ROOT"
LD_PRELOAD="$(g++ -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    EHYB_LIB="$OUT/libehyb_asan.so" python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
