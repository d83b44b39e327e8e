#!/bin/bash
# A/B library WITH the LDS ring of the long static chains (mtg_lane.h: MTG_LDS_RING, MtgCfg::kRing): the translation units
# that hold such configurations rebuilt with -DMTG_LDS_RING=1, every other object shared with the product build.
# -> mav_trajectory_generation_amd/csrc/libmtg_hip_ring.so   (then: tools/gpu_ring.sh on the GPU box)
set -e
cd "$(dirname "$0")/../mav_trajectory_generation_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result --offload-compress -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -DMTG_LDS_RING=1"
mkdir -p /tmp/ring_ab
for tu in mtg_dimlane mtg_dimlane_h6b mtg_dimlane_h5b; do hipcc $F -c $tu.hip -o /tmp/ring_ab/$tu.o & done
wait
objs=""
for o in $(python3 -c "import sys; sys.path.insert(0, '../..'); import __graft_entry__ as g; print(' '.join(g.HIP_TUS))") mtg_host mtg_basic; do
  if [ -f /tmp/ring_ab/$o.o ]; then objs="$objs /tmp/ring_ab/$o.o"; else objs="$objs $o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o libmtg_hip_ring.so $objs
ls -la libmtg_hip_ring.so
