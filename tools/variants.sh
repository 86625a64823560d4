#!/bin/bash
# Build variant libraries for A/B timing: tools/variants.sh <file.hip> NAME1="-DFLAG1 -DFLAG2" NAME2="..." ...
# -> iclr2025_3d-mom_amd/lib/var/NAME.so (all other objects are those of the current build)
src=$1; shift
cd "$(dirname "$0")/../iclr2025_3d-mom_amd/csrc"
make -j8 >/dev/null || exit 1
mkdir -p ../lib/var
base=$(basename $src .hip)
extra=""
extra="-fno-slp-vectorize"
case $base in raster_preprocess|knn) extra="$extra -ffp-contract=off";; raster_render) extra="$extra -mllvm -amdgpu-sched-strategy=max-ilp";; deform_bwd_b3) extra="$extra -mllvm -amdgpu-mfma-vgpr-form";; esac
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $extra $flags -c $base.hip -o ../lib/var/$name.o || exit 1
  objs=$(ls ../lib/obj/*.o | grep -v "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/$name.so $objs ../lib/var/$name.o || exit 1
  rm ../lib/var/$name.o
  echo built $name "($flags)"
done
