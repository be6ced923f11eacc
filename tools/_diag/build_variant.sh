#!/bin/bash
# Alternate build for same-box A/B runs: bash tools/_diag/build_variant.sh <tag> <src1,src2,...> <-D flags...>
# (sources without the .hip suffix, e.g. gdn_forward_dense,gdn_forward_dense_d128) -> gdn_amd/libgdn_var_<tag>.so,
# loaded with GDN_HIP_LIB=...; the objects of every other source come from the regular build.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; SRCS=${2//,/ }; shift 2
OBJ=$ROOT/gdn_amd/csrc/_obj
mkdir -p $OBJ/var_$TAG
for f in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -c $ROOT/gdn_amd/csrc/$f.hip -o $OBJ/var_$TAG/$f.o "$@" &
done
wait
LINK=""
for o in $OBJ/*.o; do
  b=$(basename $o .o)
  if [ -f $OBJ/var_$TAG/$b.o ]; then LINK="$LINK $OBJ/var_$TAG/$b.o"; else LINK="$LINK $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gdn_amd/libgdn_var_$TAG.so $LINK
echo built $ROOT/gdn_amd/libgdn_var_$TAG.so
