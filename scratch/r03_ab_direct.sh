#!/bin/bash
# same-box A/B: the run kernel's filter straight out of global memory (shipped) against the staged one (GSDR_PFB_DIRECT=0)
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_direct.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03_pytest_direct.log
for rep in 1 2; do
echo "== shipped (direct), rep $rep"; python scratch/pfb_sweep.py 600 1000 1230 1016 1536 1018 2>&1 | grep TONES
echo "== GSDR_PFB_DIRECT=0 (staged), rep $rep"; GSDR_PFB_DIRECT=0 python scratch/pfb_sweep.py 600 1000 1230 1016 1536 1018 2>&1 | grep TONES
echo "== GSDR_PFB_CU=1 direct: powers of two through the run kernel, rep $rep"; GSDR_PFB_CU=1 python scratch/pfb_sweep.py 512 1024 2048 4096 2>&1 | grep TONES
echo "== the library's choice for powers of two, rep $rep"; python scratch/pfb_sweep.py 512 1024 2048 4096 2>&1 | grep TONES
done | tee gpurun_out/r03_pfb_ab_direct.log
