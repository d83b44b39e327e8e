#!/bin/bash
# round 5, visit C: the whole GPU suite (no -x: every failure in one visit)
mkdir -p gpurun_out/r05c
timeout 1700 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r05c/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05c/tests.log
tail -30 gpurun_out/r05c/tests.log
