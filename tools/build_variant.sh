#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip "-DFLAG ..." — a variant BUILD of libmdhip.so for A/B runs (tools/ab_libs*.py):
# SOURCE is recompiled with the extra flags, every other object is taken from the regular build; the result is
# tools/_bin/libmdhip_NAME.so (git-ignored; travels to the GPU box with the snapshot).
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$R/mdproptools_amd/csrc/_obj
mkdir -p $R/tools/_bin/obj_$NAME
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function "$@" \
  -c $R/mdproptools_amd/csrc/$SRC -o $R/tools/_bin/obj_$NAME/${SRC%.hip}.o
OBJS=""
for o in $OBJ/*.o; do
  b=$(basename $o)
  if [ "$b" = "${SRC%.hip}.o" ]; then OBJS="$OBJS $R/tools/_bin/obj_$NAME/$b"; else OBJS="$OBJS $o"; fi
done
hipcc $OBJS -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lpthread -Wl,-rpath,/opt/rocm/lib -o $R/tools/_bin/libmdhip_$NAME.so
echo built $R/tools/_bin/libmdhip_$NAME.so
