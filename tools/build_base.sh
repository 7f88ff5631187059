#!/bin/bash
# build ab_base/base.so from the csrc + include of a git revision (default HEAD), with build.py's flags.
# NB the host-side python is NOT switched: only use for kernel-only A/Bs where the C ABI did not change.
REV=${1:-HEAD}
rm -rf ab_base/src; mkdir -p ab_base/src/csrc ab_base/src/include ab_base/src/obj
for f in $(git ls-tree --name-only $REV ist-net_amd/csrc/); do git show $REV:$f > ab_base/src/csrc/$(basename $f); done
for f in $(git ls-tree --name-only $REV include/); do git show $REV:$f > ab_base/src/include/$(basename $f); done
# build.py resolves "../../include" relative to csrc: mirror that layout
mkdir -p ab_base/src/pkg; mv ab_base/src/csrc ab_base/src/pkg/csrc
for s in ab_base/src/pkg/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -c $s -o ab_base/src/obj/$(basename $s .hip).o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden ab_base/src/obj/*.o -o ab_base/base.so && rm -rf ab_base/src && ls -la ab_base
