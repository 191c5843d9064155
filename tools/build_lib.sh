#!/bin/bash
# Build libeigenexa_amd.so for gfx950 (cross-compiles without a GPU).  Usage: tools/build_lib.sh [-j N]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/eigenexa_amd/csrc"
OUT="$ROOT/eigenexa_amd/lib"
OBJ="$OUT/obj"
mkdir -p "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I$ROOT/include -I$SRC"
pids=()
for f in "$SRC"/*.hip; do
  b=$(basename "$f" .hip)
  o="$OBJ/$b.o"
  # rebuild if source or any header is newer than the object
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find "$SRC" "$ROOT/include" -name '*.h' -newer "$o" 2>/dev/null)" ]; then
    echo "hipcc $b.hip"
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libeigenexa_amd.so" "$OBJ"/*.o -ldl
echo "built $OUT/libeigenexa_amd.so"
