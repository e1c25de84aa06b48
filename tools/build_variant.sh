#!/bin/bash
# build an ablation variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>
set -e
name=$1; shift
out=/root/repo/presight_amd/_variants
mkdir -p $out/$name
cd /root/repo/presight_amd/csrc
for f in *.hip lib.cpp; do
  x=""; [[ $f == *.cpp ]] && x="-x hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -I. "$@" $x -c $f -o $out/$name/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $out/$name/*.o
echo built $out/lib_$name.so
