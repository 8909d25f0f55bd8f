#!/bin/bash
for c in 1 0; do echo "== GSDR_PFB_COL=$c"; GSDR_PFB_COL=$c STAMP_PY=scratch/stamp_pfb3.py bash scratch/stamp_pfb.sh 1230 1000 1016; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_pfb_stamps_col.log
for c in 1 0; do echo "== GSDR_PFB_COL=$c"; GSDR_PFB_COL=$c python scratch/pfb_sweep.py 1000 1230 1016 2>&1 | grep TONES; done | tee -a gpurun_out/r03_pfb_stamps_col.log
