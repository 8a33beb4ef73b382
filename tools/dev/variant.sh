#!/bin/bash
# tools/dev/variant.sh NAME "flags" file1 [file2 ...]: a variant library whose listed sources (ls_api ls_mq ...) are compiled with the flags
N=$1; F=$2; shift 2
cd $(cd "$(dirname "$0")/../.." && pwd)/lean-explore_amd/csrc
mkdir -p _build_$N ../variants
OBJS=$(ls _build/*.o)
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $F -c $f.hip -o _build_$N/$f.o || exit 1
  OBJS=$(echo "$OBJS" | grep -v "/$f.o"); OBJS="$OBJS _build_$N/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libleansearch_$N.so $OBJS -ldl
