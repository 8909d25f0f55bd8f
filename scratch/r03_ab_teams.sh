#!/bin/bash
# same-box A/B: the frames of a run through their stages as independent teams of waves (shipped) against all waves in step
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_teams.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_teams.log
for rep in 1 2; do
echo "== teams (shipped), rep $rep"; timeout -k 10 120 python scratch/pfb_sweep.py 128 256 512 1000 1024 1536 2048 2560 2>&1 | grep "TONES"
echo "== GSDR_PFB_TEAMS=0, rep $rep"; GSDR_PFB_TEAMS=0 timeout -k 10 120 python scratch/pfb_sweep.py 128 256 512 1000 1024 1536 2048 2560 2>&1 | grep "TONES"
done | tee gpurun_out/r03_pfb_ab_teams.log
