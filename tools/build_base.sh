#!/bin/bash
# build tmp_ab/base.so from the csrc of a git revision (default HEAD) -- NB host-side python is NOT switched
REV=${1:-HEAD}
mkdir -p tmp_ab/src/ist-net_amd/csrc tmp_ab/src/include
for f in pw_mlp.hip pn2_index_ops.hip; do git show $REV:ist-net_amd/csrc/$f > tmp_ab/src/ist-net_amd/csrc/$f; done
for f in istnet_pn2.h istnet_pw.h; do git show $REV:include/$f > tmp_ab/src/include/$f; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -I tmp_ab/src/include tmp_ab/src/ist-net_amd/csrc/*.hip -o tmp_ab/base.so && rm -rf tmp_ab/src && ls -la tmp_ab
