#!/bin/bash
# dev aid: -DVN_STAMP build of the library (libvaenmf_dbg.so, not shipped) for tools/stamps.py
set -e
cd "$(dirname "$0")/../guided-vae-nmf_amd/csrc"
for s in engine aux plan labels stream; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DVN_STAMP -c $s.hip -o /tmp/${s}_dbg.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vaenmf/libvaenmf_dbg.so /tmp/engine_dbg.o /tmp/aux_dbg.o /tmp/plan_dbg.o /tmp/labels_dbg.o /tmp/stream_dbg.o
