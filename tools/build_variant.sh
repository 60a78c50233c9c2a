#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [file.hip ...]: meso_amd/libmeso_hip_NAME.so = the in-tree objects with the named
# sources recompiled under extra flags (A/B timing of kernel variants in one GPU session through MESO_LIB)
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../meso_amd"
mkdir -p csrc/_obj/var_$name
objs=""
for o in csrc/_obj/*.hip.o; do
  b=$(basename $o .o)
  use=$o
  for f in "$@"; do
    if [ "$f" = "$b" ]; then
      hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w $flags -c csrc/$f -o csrc/_obj/var_$name/$b.o
      use=csrc/_obj/var_$name/$b.o
    fi
  done
  objs="$objs $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o libmeso_hip_$name.so $objs -L/opt/rocm/lib -lrccl
echo libmeso_hip_$name.so
