#!/bin/sh
# Builds oracle/_ref/ref_lmp: the reference's own stock-CPU sources for the DPD step, compiled UNMODIFIED where they
# lie under $REF/src (g++ directly on the files; the reference's build system is not run, nothing is generated,
# no header is stood in for), plus this repository's driver oracle/ref_harness.cpp.  Outputs go to oracle/_ref only.
# TEST INFRASTRUCTURE: used by tests/ to pin the CPU restatement oracle/lmp_dpd_cpu.c; never linked into the product.
#
# The remaining undefined symbols (members of Atom, Force, Update, Modify, Output, Input, Variable, Lattice, Region ...:
# classes whose .cpp includes a committed style_*.h naming absent package headers, see ref_harness.cpp) are left
# unresolved on purpose (--unresolved-symbols=ignore-all): the driver never reaches a call of them.
set -e
REF=${MESO_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
SRC=$REF/src
[ -d "$SRC" ] || { echo "build_ref: $SRC not present (GPU box?): keeping prebuilt oracle/_ref" >&2; exit 0; }
mkdir -p "$OUT/obj"
FILES="random_mars random_park pair_dpd pair fix_nve fix neighbor neigh_half_bin neigh_half_nsq neigh_half_multi
neigh_full neigh_derive neigh_gran neigh_respa neigh_bond neigh_stencil neigh_list neigh_request comm procmap domain
atom_vec atom_vec_atomic group memory error universe math_extra"
CXXFLAGS="-O2 -fPIC -ffp-contract=off -w -I$SRC -I$SRC/STUBS"
OBJS=""
for f in $FILES; do
    o=$OUT/obj/$f.o
    if [ ! -f "$o" ] || [ "$SRC/$f.cpp" -nt "$o" ]; then g++ $CXXFLAGS -c "$SRC/$f.cpp" -o "$o"; fi
    OBJS="$OBJS $o"
done
gcc -O2 -fPIC -w -I"$SRC/STUBS" -c "$SRC/STUBS/mpi.c" -o "$OUT/obj/mpi_stubs.o"
g++ $CXXFLAGS -c "$HERE/ref_harness.cpp" -o "$OUT/obj/ref_harness.o"
g++ -o "$OUT/ref_lmp" "$OUT/obj/ref_harness.o" $OBJS "$OUT/obj/mpi_stubs.o" -Wl,--unresolved-symbols=ignore-all -lm
echo "$OUT/ref_lmp"
# the reference's own bonded styles (src/MOLECULE, compiled where they lie) behind oracle/ref_bonded.cpp -> oracle/_ref/ref_bonded
BOBJS=""
for f in bond angle memory error universe; do
    o=$OUT/obj/$f.o
    if [ ! -f "$o" ] || [ "$SRC/$f.cpp" -nt "$o" ]; then g++ $CXXFLAGS -c "$SRC/$f.cpp" -o "$o"; fi
    BOBJS="$BOBJS $o"
done
for f in bond_harmonic bond_fene angle_harmonic; do
    o=$OUT/obj/$f.o
    if [ ! -f "$o" ] || [ "$SRC/MOLECULE/$f.cpp" -nt "$o" ]; then g++ $CXXFLAGS -I"$SRC/MOLECULE" -c "$SRC/MOLECULE/$f.cpp" -o "$o"; fi
    BOBJS="$BOBJS $o"
done
g++ $CXXFLAGS -I"$SRC/MOLECULE" -c "$HERE/ref_bonded.cpp" -o "$OUT/obj/ref_bonded.o"
g++ -o "$OUT/ref_bonded" "$OUT/obj/ref_bonded.o" $BOBJS "$OUT/obj/mpi_stubs.o" -Wl,--unresolved-symbols=ignore-all -lm
echo "$OUT/ref_bonded"
