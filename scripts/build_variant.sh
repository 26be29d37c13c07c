#!/bin/bash
# builds scs_amd/lib_var/<name>/libscsamd.so = the fp64 library with cones.hip compiled with extra flags (A/B measurements on one box)
# usage: scripts/build_variant.sh <name> [flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/scs_amd/lib_var/$name
cd $R/scs_amd/csrc
hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 "$@" -c cones.hip -o $R/scs_amd/lib_var/$name/cones.o
objs=$(ls ../lib/obj64/*.o | grep -v "/cones.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scs_amd/lib_var/$name/libscsamd.so $objs $R/scs_amd/lib_var/$name/cones.o -Wl,--version-script=exports_full.map -Wl,-Bsymbolic
rm $R/scs_amd/lib_var/$name/cones.o
