#!/bin/bash
# Experimental library next to the shipped one, rebuilding only the named translation units with extra flags:
#   bash tools/build_variant.sh NAME "-DFV_X_FOO=1 ..." "pair_wino_k3 pair_wino_k7 pair_wino_k11"
# -> vocoder_amd/csrc/libfishvoc_x_NAME.so (the other objects come from the shipped build/; run `make` there first).
# Use with FV_LIB_PATH=vocoder_amd/csrc/libfishvoc_x_NAME.so; `make clean` removes libfishvoc_x*.so and build_*.
set -e
NAME=$1; FLAGS=$2; TUS=$3
cd "$(dirname "$0")/../vocoder_amd/csrc"
mkdir -p build_$NAME
objs=""
pids=""
for tu in $TUS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -fvisibility=hidden -DFV_BUILD $FLAGS -c $tu.hip -o build_$NAME/$tu.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
for o in $(make -s print-objs); do
  b=$(basename $o .o)
  if echo " $TUS " | grep -q " $b "; then objs="$objs build_$NAME/$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfishvoc_x_$NAME.so $objs
echo built libfishvoc_x_$NAME.so
