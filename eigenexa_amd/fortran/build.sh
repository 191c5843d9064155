#!/bin/bash
# Builds the Fortran module and the example against libeigenexa_amd.so with the image's flang.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
OUT="$HERE/_build"
mkdir -p "$OUT"
cd "$OUT"
$FC -cpp -O2 -c "$HERE/eigen_libs_mod.F90" -o eigen_libs_mod.o
$FC -cpp -O2 -c "$HERE/example_frank.F90" -o example_frank.o
$FC -o example_frank example_frank.o eigen_libs_mod.o -L"$HERE/../lib" -leigenexa_amd -Wl,-rpath,"$HERE/../lib"
echo "built $OUT/example_frank"
