#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never the shipped library): eigenexa_amd/lib/libeigenexa_amd_dbg.so
# use with EIGX_LIB=eigenexa_amd/lib/libeigenexa_amd_dbg.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/eigenexa_amd/csrc"
OUT="$ROOT/eigenexa_amd/lib"
OBJ="$OUT/obj_dbg"
mkdir -p "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -DEIGX_STAMPS -I$ROOT/include -I$SRC"
pids=()
for f in "$SRC"/*.hip; do
  b=$(basename "$f" .hip)
  $HIPCC $FLAGS -c "$f" -o "$OBJ/$b.o" 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libeigenexa_amd_dbg.so" "$OBJ"/*.o -ldl
echo "built $OUT/libeigenexa_amd_dbg.so"
