#!/bin/bash
# Builds libsmhip variants that differ in the tile kernel's patch walk and q extent (tools/bin/tile_o<order>_q<bytes>.so);
# tools/tile_sizes.py <lib> measures each.  Development tool: run here (hipcc cross-compiles), measure on the GPU box.
set -e
cd "$(dirname "$0")/.."
python -c "from simplemath_amd.build import build_lib; build_lib()"
mkdir -p tools/bin
for o in 0 1; do for q in 512 1024; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Iinclude -Isimplemath_amd/csrc \
      -DSMHIP_TILE_ORDER=$o -DSMHIP_TILE_Q_BYTES=$q -c simplemath_amd/csrc/broadcast.hip -o tools/bin/bcast_o${o}_q${q}.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/tile_o${o}_q${q}.so tools/bin/bcast_o${o}_q${q}.o \
      $(ls simplemath_amd/lib/obj/*.o | grep -v broadcast.o) -lhiprtc -ldl ) &
done; done
wait
ls -la tools/bin/tile_o*.so
