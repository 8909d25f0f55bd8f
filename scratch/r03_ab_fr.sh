#!/bin/bash
# same-box A/B: frames per workgroup and threads per workgroup of the frame-per-workgroup kernel
for rep in 1 2; do
echo "== shipped, rep $rep"; python scratch/pfb_sweep.py 256 512 1024 2048 2>&1 | grep TONES
echo "== GSDR_PFB_FR=2 (1024: two frames per 256 threads), rep $rep"; GSDR_PFB_FR=2 python scratch/pfb_sweep.py 1024 2048 2>&1 | grep TONES
echo "== GSDR_PFB_FR=4, rep $rep"; GSDR_PFB_FR=4 python scratch/pfb_sweep.py 512 1024 2>&1 | grep TONES
echo "== GSDR_PFB_WIDE=0 (2048 on 256 threads), rep $rep"; GSDR_PFB_WIDE=0 python scratch/pfb_sweep.py 2048 2>&1 | grep TONES
echo "== GSDR_PFB_WIDE=1 (512 threads), rep $rep"; GSDR_PFB_WIDE=1 python scratch/pfb_sweep.py 256 512 1024 2>&1 | grep TONES
echo "== GSDR_PFB_WIDE=1 GSDR_PFB_FR=2, rep $rep"; GSDR_PFB_WIDE=1 GSDR_PFB_FR=2 python scratch/pfb_sweep.py 1024 2>&1 | grep TONES
done | tee gpurun_out/r03_pfb_ab_fr.log
