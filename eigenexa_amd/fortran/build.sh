#!/bin/bash
# Builds the Fortran module and the example against libeigenexa_amd.so with the image's flang; with an MPI whose
# mpif.h and Fortran-77 binding library are found (MPI_INC / MPI_LIB, default /opt/conda), also the MPI build of the
# module (+ eigen_blacs_mod) and tests/fortran/ref_caller.F90, a caller that uses the symbol set of the reference's
# own benchmark sources.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
OUT="$HERE/_build"
mkdir -p "$OUT" "$OUT/mpi"
cd "$OUT"
$FC -cpp -O2 -c "$HERE/eigen_libs_mod.F90" -o eigen_libs_mod.o
$FC -cpp -O2 -c "$HERE/example_frank.F90" -o example_frank.o
$FC -o example_frank example_frank.o eigen_libs_mod.o -L"$HERE/../lib" -leigenexa_amd -Wl,-rpath,"$HERE/../lib"
echo "built $OUT/example_frank"
MPI_INC=${MPI_INC:-/opt/conda/include}
MPI_LIB=${MPI_LIB:-/opt/conda/lib}
if [ -f "$MPI_INC/mpif.h" ] && ls "$MPI_LIB"/libmpifort.so* >/dev/null 2>&1; then
  cd "$OUT/mpi"
  $FC -cpp -O2 -DEIGX_WITH_MPI -DEIGX_WITH_BLACS -I"$MPI_INC" -c "$HERE/eigen_libs_mod.F90" -o eigen_libs_mod.o
  $FC -cpp -O2 -I"$MPI_INC" -c "$ROOT/tests/fortran/ref_caller.F90" -o ref_caller.o
  # eigen_blacs_mod's BLACS calls stay unresolved unless the caller links a BLACS: keep that module out of this link
  $FC -cpp -O2 -DEIGX_WITH_MPI -I"$MPI_INC" -c "$HERE/eigen_libs_mod.F90" -o eigen_libs_mod_noblacs.o
  $FC -o ref_caller ref_caller.o eigen_libs_mod_noblacs.o -L"$HERE/../lib" -leigenexa_amd -Wl,-rpath,"$HERE/../lib" \
      -L"$MPI_LIB" -lmpifort -lmpi -Wl,-rpath,"$MPI_LIB"
  echo "built $OUT/mpi/ref_caller"
fi
