#!/bin/bash
# Alternate build of the dense kernels for same-box A/B runs: bash tools/_diag/build_variant.sh <tag> <-D flags...>
# -> gdn_amd/libgdn_var_<tag>.so (load it with GDN_HIP_LIB=...); the other objects come from the regular build.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; shift
OBJ=$ROOT/gdn_amd/csrc/_obj
mkdir -p $OBJ/var_$TAG
for f in gdn_forward_dense gdn_forward_dense_d128; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -c $ROOT/gdn_amd/csrc/$f.hip -o $OBJ/var_$TAG/$f.o "$@" &
done
wait
OTHERS=$(ls $OBJ/*.o | grep -v gdn_forward_dense)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gdn_amd/libgdn_var_$TAG.so $OTHERS $OBJ/var_$TAG/gdn_forward_dense.o $OBJ/var_$TAG/gdn_forward_dense_d128.o
echo built $ROOT/gdn_amd/libgdn_var_$TAG.so
