#!/bin/bash
# Developer aid: build a variant of the library with extra compiler flags for ONE source file.
#   bash profiles/tools/mkvariant.sh <name> <file.hip> "<flags>"   ->  variants/<name>.so
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../../cameracalibration_amd/csrc"
make -s -j8
mkdir -p ../../variants /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $flags -c $src -o /tmp/var_$name/${src%.hip}.o
objs=""
for o in _build/*.o; do
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs /tmp/var_$name/${src%.hip}.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/$name.so $objs
echo built variants/$name.so
