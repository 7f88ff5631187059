#!/bin/bash
# build ab_base/phase.so from the WORKING TREE with -DISTNET_PHASE_TIMING (build.py's flags otherwise): the library
# tools/bwd_mid_phases.py and tools/fwd_sk_phases.py copy over the product's on the GPU box.
mkdir -p ab_base/obj_phase
for s in ist-net_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -DISTNET_PHASE_TIMING -c $s -o ab_base/obj_phase/$(basename $s .hip).o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden ab_base/obj_phase/*.o -o ab_base/phase.so && rm -rf ab_base/obj_phase && ls -la ab_base/phase.so
