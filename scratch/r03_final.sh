#!/bin/bash
# round-3 closing run: smoke(), the whole GPU suite (margins), the default bench line, the two-rank rehearsal
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -q > gpurun_out/r03_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --steps 100 --warmup 10 --no-extras --no-cpu > gpurun_out/r03_n2_one_device.json 2> gpurun_out/r03_n2_one_device.err; echo "n2 rc=$?"
cut -c1-400 gpurun_out/r03_n2_one_device.json
python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 4096 8192 1018 1004 2>&1 | grep TONES | tee gpurun_out/r03_pfb_sweep_final.log
{ echo "== radix 8 / 6 / 10 (shipped)"; python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 2>&1 | grep TONES; echo "== GSDR_PFB_RADIX8=0: radix 4 / 2 as in round 2"; GSDR_PFB_RADIX8=0 python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 2>&1 | grep TONES; echo "== GSDR_PFB_CU=1: the run kernel forced"; GSDR_PFB_CU=1 python scratch/pfb_sweep.py 64 256 1024 2048 2>&1 | grep TONES; } > gpurun_out/r03_pfb_sweep_r8.log
python scratch/pfb_api_ab.py 2>&1 | grep TONES > gpurun_out/r03_pfb_api_ab_final.log
