#!/bin/bash
# Sensitivity of the head start (share of the event chain, number of workgroups): tools/ab_probe.py per setting.
for hs in 0 0.72,1024 0.5,1024 0.6,1024 0.8,1024 0.9,1024 0.72,768 0.72,1280 0.72,1536 0.6,768 0.8,1280 0.72,1024; do
  echo "DPOMP_HEAD_START=$hs"
  DPOMP_HEAD_START=$hs timeout -k 10 200 python -u tools/ab_probe.py - 2>&1 | grep -v amdgpu.ids || exit 1
done
