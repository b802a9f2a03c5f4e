#!/bin/bash
# Dev tool: build an experimental variant of libfv_hotpath.so with extra compiler flags, e.g.
#   tools/build_variant.sh myflag -DFV_MY_EXPERIMENT     (the FV_ABLATE_* macros of rounds 1-2 were removed from csrc/ in round 3)
#   FV_LIB_PATH=tools/_variants/libfv_myflag.so python tools/layer_bench.py
# (tools/_variants/ is git-ignored; the .so still travels to the GPU box.)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/_variants; obj=$out/obj_$name
mkdir -p "$obj"
pids=()
for s in "$root"/face_vijnana_yolov3_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" -c "$s" -o "$obj/$(basename "${s%.hip}").o" 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libfv_$name.so" "$obj"/*.o
echo "$out/libfv_$name.so"
