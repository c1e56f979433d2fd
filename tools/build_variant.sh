#!/bin/bash
# dev aid: build a variant of the library for A/B timing: tools/build_variant.sh <tag> [extra hipcc flags]
# -> guided-vae-nmf_amd/vaenmf/libvaenmf_<tag>.so   (run with VAENMF_LIB=<path>; see tools/ab.sh)
# Only the sources named in VN_VARIANT_SRCS (default: chain stream) are recompiled with the extra flags; the rest
# comes from the objects of the main build.
set -e
tag=$1; shift
cd "$(dirname "$0")/../guided-vae-nmf_amd/csrc"
srcs=${VN_VARIANT_SRCS:-"chain stream"}
objs=""
for s in engine chain aux plan labels stream; do
  if [[ " $srcs " == *" $s "* ]]; then
    ff=""; [ $s = chain ] && ff="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $ff "$@" -c $s.hip -o /tmp/${s}_$tag.o &
    objs="$objs /tmp/${s}_$tag.o"
  else
    objs="$objs $s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vaenmf/libvaenmf_$tag.so $objs
