#!/bin/bash
# GPU visit: the LDS ring of the long static chains (MtgCfg::kRing) -- layout probe, product build (direct form) against the
# ring build (tools/build_ring_ab.sh), bit identity, then the long-chain parity tests ON THE RING BUILD.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ring; mkdir -p $O
timeout 60 tools/micro/lds_dma_probe > $O/probe.txt 2>&1; cat $O/probe.txt
RING=$PWD/mav_trajectory_generation_amd/csrc/libmtg_hip_ring.so
timeout 600 python tools/ab_lds_ring.py write $O > $O/noring.jsonl 2>$O/noring.err
MTG_HIP_LIB=$RING timeout 600 python tools/ab_lds_ring.py check $O > $O/ring.jsonl 2>$O/ring.err
rm -f $O/*.pt
paste -d'\n' $O/noring.jsonl $O/ring.jsonl
tail -3 $O/ring.err
MTG_HIP_LIB=$RING timeout 1500 python -m pytest tests -m gpu -q -x -k "long or dimlane or default_dispatch or vs_reference or multi or forms" 2>&1 | tail -8 | tee $O/pytest_tail.txt
