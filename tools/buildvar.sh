#!/bin/bash
# build a variant library on the build host: tools/buildvar.sh <name> <extra hipcc flags for fused.hip ...>
#   -> pyiga_amd/libigx_<name>.so  (the other objects are the ones of the regular build; select with IGX_LIB on the GPU box)
name=$1; shift
src=${VARSRC:-fused}          # VARSRC=geoa tools/buildvar.sh <name> <flags>: the variant file (default fused.hip)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/pyiga_amd/csrc/build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wno-unused-function -cuid=igx_$src -Xarch_host -ffp-contract=off "$@" -c $R/pyiga_amd/csrc/$src.hip -o $R/pyiga_amd/csrc/build_var/${src}_$name.o || exit 1
objs=""
for f in igx_api kern_basis kern_entries kern_vector sumfact sumfact_hi geoa aca fused fused3 rtc kron; do [ $f = $src ] || objs="$objs $R/pyiga_amd/csrc/build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/pyiga_amd/libigx_$name.so $objs $R/pyiga_amd/csrc/build_var/${src}_$name.o -ldl
