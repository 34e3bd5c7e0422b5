#!/bin/bash
# Build libsmhip with extra compiler flags into tools/bin/<name>.so (experiments only; SMHIP_LIBRARY=<that file> makes the Python
# binding load it).  usage: tools/build_variant.sh name -DFOO ...   |   tools/build_variant.sh preload -mllvm -amdgpu-kernarg-preload-count=16
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/obj_$name; mkdir -p $out
for f in runtime contiguous broadcast reduce fill fused chain tiny jit sharded inline; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I$root/include -I$root/simplemath_amd/csrc "$@" -c $root/simplemath_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/bin/$name.so $out/*.o -lhiprtc -ldl
echo built $root/tools/bin/$name.so
