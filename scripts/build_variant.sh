#!/bin/bash
# build_variant.sh <name> <source.hip> <extra hipcc flags...>: a second libflvis_hip with ONE source compiled with extra flags, written to
# build_variants/libflvis_hip_<name>.so (git-ignored; FLVIS_LIB_PATH selects it).  The in-tree library and its objects stay untouched.
set -eu
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -c "import flvis_amd.build as b; b.build()"
mkdir -p build_variants
obj=build_variants/$(basename "$src").$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-result -Wno-unused-value "$@" -c "flvis_amd/csrc/$src" -o "$obj"
objs=$(ls flvis_amd/csrc/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build_variants/libflvis_hip_$name.so" $objs "$obj" -lz
echo "build_variants/libflvis_hip_$name.so"
