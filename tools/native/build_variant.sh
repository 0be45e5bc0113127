#!/bin/bash
# tools/native/build_variant.sh <name> <csrc file, e.g. elementwise.hip>: the tree's library with ONE translation unit replaced by its patched copy
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2
tmp=$(mktemp -d)
cp internnav_amd/csrc/$src $tmp/$src
(cd $tmp && patch -p3 $src < $OLDPWD/tools/native/variants/$name.patch)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=fast -Iinternnav_amd/csrc -x hip -c $tmp/$src -o $tmp/variant.o
objs=$(ls internnav_amd/csrc/build/*.o | grep -v "/$src.o")
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs $tmp/variant.o -o tools/native/libina_$name.so
echo tools/native/libina_$name.so
