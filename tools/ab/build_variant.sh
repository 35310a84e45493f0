#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG=.. -DFLAG2=.."   -> scratch_dbg/lib_NAME.so (a full build of the library
# with extra compiler flags, for A/B runs on one GPU box through FSGM_LIB_PATH; scratch_dbg/ is git-ignored)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../.."
mkdir -p scratch_dbg/obj_$NAME
for f in fsgm_amd/csrc/*.hip; do
  o=scratch_dbg/obj_$NAME/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function -Iinclude $FLAGS -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch_dbg/lib_$NAME.so scratch_dbg/obj_$NAME/*.o
rm -rf scratch_dbg/obj_$NAME
ls -la scratch_dbg/lib_$NAME.so
