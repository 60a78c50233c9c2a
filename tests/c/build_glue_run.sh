#!/bin/sh
# Builds tests/c/glue_run: the LAMMPS-side binding (lammps_glue/) + tests/c/glue_run_harness.cpp against the reference's own headers
# and unmodified base classes (compiled where they lie, as oracle/build_ref.sh does - nothing generated, nothing stood in for),
# linked with the real meso_amd/libmeso_hip.so.  Only where the reference tree is mounted (this container); the binary travels to
# the GPU box with the snapshot.  TEST INFRASTRUCTURE (tests/test_gpu_glue_run.py).
set -e
REF=${MESO_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
SRC=$REF/src
[ -d "$SRC" ] || { echo "build_glue_run: $SRC not present: keeping prebuilt tests/c/glue_run" >&2; exit 0; }
OBJ=$ROOT/oracle/_ref/obj
mkdir -p "$OBJ"
CXXFLAGS="-O1 -fPIC -w -std=c++11 -DLAMMPS_GZIP -I$SRC -I$SRC/STUBS -I$SRC/MOLECULE -I$ROOT/include -I$ROOT/lammps_glue"
OBJS=""
for f in pair fix compute integrate bond angle group domain comm procmap neighbor neigh_half_bin neigh_half_nsq neigh_half_multi neigh_full \
         neigh_derive neigh_gran neigh_respa neigh_bond neigh_stencil neigh_list neigh_request atom_vec atom_vec_atomic memory error universe math_extra; do
    o=$OBJ/$f.o
    if [ ! -f "$o" ] || [ "$SRC/$f.cpp" -nt "$o" ]; then g++ -O2 -fPIC -ffp-contract=off -w -I$SRC -I$SRC/STUBS -c "$SRC/$f.cpp" -o "$o"; fi
    OBJS="$OBJS $o"
done
[ -f "$OBJ/mpi_stubs.o" ] || gcc -O2 -fPIC -w -I"$SRC/STUBS" -c "$SRC/STUBS/mpi.c" -o "$OBJ/mpi_stubs.o"
g++ $CXXFLAGS -c "$ROOT/lammps_glue/meso_hip_glue.cpp" -o "$OBJ/meso_hip_glue.o"
g++ $CXXFLAGS -c "$HERE/glue_run_harness.cpp" -o "$OBJ/glue_run_harness.o"
g++ -o "$HERE/glue_run" "$OBJ/glue_run_harness.o" "$OBJ/meso_hip_glue.o" $OBJS "$OBJ/mpi_stubs.o" -L"$ROOT/meso_amd" -lmeso_hip \
    -Wl,-rpath,'$ORIGIN/../../meso_amd' -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-all -lm
echo "$HERE/glue_run"
