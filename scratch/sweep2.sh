#!/bin/bash
run() { # name env...
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 5 --workload $WL --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$WL $name', d['value'], 'Msps  kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])
"
}
WL=c3
run nch128 GSDR_DDC_NCH=128
run nch128_lds40k GSDR_DDC_NCH=128 GSDR_DDC_LDS=40960
run nch125_lds40k GSDR_DDC_NCH=125 GSDR_DDC_LDS=40960
run nch160_lds32k GSDR_DDC_NCH=160 GSDR_DDC_LDS=32768
run nch192 GSDR_DDC_NCH=192
run nch96_lds53k GSDR_DDC_NCH=96 GSDR_DDC_LDS=54000
run nch64_lds80k GSDR_DDC_NCH=64 GSDR_DDC_LDS=81920
run nch250 GSDR_DDC_NCH=250
WL=c2
run nch1280 GSDR_DDC_NCH=1280
run nch1280_lds32k GSDR_DDC_NCH=1280 GSDR_DDC_LDS=32768
run nch1024_lds40k GSDR_DDC_NCH=1024 GSDR_DDC_LDS=40960
run nch1536 GSDR_DDC_NCH=1536
run nch1000_lds40k GSDR_DDC_NCH=1000 GSDR_DDC_LDS=40960
run nch2000 GSDR_DDC_NCH=2000
