#!/bin/bash
# tools/neigh_ablate_lib.sh "LIBS" [bench args]: tools/neigh_ablate.sh under several in-tree builds ('-' = default library)
libs=$1; shift
for v in $libs; do
  if [ $v = - ]; then unset MESO_LIB; else export MESO_LIB=$PWD/meso_amd/libmeso_hip_$v.so; fi
  echo "== $v"
  bash tools/neigh_ablate.sh "$@"
done
