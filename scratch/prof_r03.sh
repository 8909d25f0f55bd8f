#!/bin/bash
# round-3 profiles: rocprofv3 kernel-trace stats of the in-order entry (the roofline's kernel duration) and of the
# default (overlapped) command, plus separate --pmc passes (kernel alone: dispatches serialised).  "c3flat" is C3
# through the packed-FP32 VALU engine (GSDR_DDC_MFMA=0).  Progress goes to gpurun_out/prof_r03.progress.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for WL in ${WLS:-c3 c3flat c2 pfb c4}; do
  W=$WL; unset GSDR_DDC_MFMA
  if [ $WL = c3flat ]; then W=c3; export GSDR_DDC_MFMA=0; fi
  OUT=$R/gpurun_out/prof_$WL; IO=$R/gpurun_out/prof_${WL}_io
  mkdir -p $OUT $IO
  B="$R/bench.py --workload $W --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 --min-seconds 0.2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $IO/trace -- python3 $B --api inorder > $IO/trace.log 2>&1 || true
  echo "$WL inorder trace done" >> $R/gpurun_out/prof_r03.progress
  if [ $WL != c3flat ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B > $OUT/trace.log 2>&1 || true
    echo "$WL default trace done" >> $R/gpurun_out/prof_r03.progress
  fi
  P="$R/bench.py --workload $W --api inorder --no-extras --no-cpu --no-host-api --steps 30 --warmup 5 --min-seconds 0.01"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $P > $OUT/pmc1.log 2>&1 || true
  echo "$WL pmc1 done" >> $R/gpurun_out/prof_r03.progress
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc2 -- python3 $P > $OUT/pmc2.log 2>&1 || true
  echo "$WL pmc2 done" >> $R/gpurun_out/prof_r03.progress
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $P > $OUT/pmc3.log 2>&1 || true
  echo "$WL pmc3 done" >> $R/gpurun_out/prof_r03.progress
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $P > $OUT/pmc4.log 2>&1 || true
  # gpurun copies at most 64 MiB back: the per-launch trace of a 4 us kernel is 15 MB, its first 30 000 rows do
  find $OUT $IO -name '*kernel_trace.csv' -size +4M -exec sh -c 'head -n 30000 "$1" > "$1.t" && mv "$1.t" "$1"' _ {} \;
  echo "prof $WL done" | tee -a $R/gpurun_out/prof_r03.progress
done
