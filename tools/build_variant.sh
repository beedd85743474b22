#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-DFLAG ...]   ->  gpurun_variants/libb4r_<name>.so
# A/B of two builds of one translation unit inside one GPU session (B4R_LIB_PATH=gpurun_variants/libb4r_<name>.so): the other
# objects are the ones bert4rec_amd/build.py left in csrc/.
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/gpurun_variants
obj=$root/gpurun_variants/${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $root/bert4rec_amd/csrc/$src -o $obj
others=$(ls $root/bert4rec_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/gpurun_variants/libb4r_${name}.so $obj $others
rm -f $obj
echo built gpurun_variants/libb4r_${name}.so
