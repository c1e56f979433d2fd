#!/bin/bash
# dev aid: build a variant of the library for A/B timing: tools/build_variant.sh <tag> [extra hipcc flags]
# -> guided-vae-nmf_amd/vaenmf/libvaenmf_<tag>.so   (run with VAENMF_LIB=<path>; see tools/ab.sh)
set -e
tag=$1; shift
cd "$(dirname "$0")/../guided-vae-nmf_amd/csrc"
for s in engine aux plan labels stream; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c $s.hip -o /tmp/${s}_$tag.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vaenmf/libvaenmf_$tag.so /tmp/engine_$tag.o /tmp/aux_$tag.o /tmp/plan_$tag.o /tmp/labels_$tag.o /tmp/stream_$tag.o
