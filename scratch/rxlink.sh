#!/bin/bash
# PCIe-inclusive table of DESIGN.md section 6
hipcc -O2 -std=c++17 -w -Iinclude tools/rx_link.cpp -Lgpu_sdr_amd -lgsdr -Wl,-rpath,$PWD/gpu_sdr_amd -lpthread -o /tmp/rx_link || exit 1
for cfg in "256 100 300" "256 100 300 pipe" "2048 1000 300" "2048 1000 300 pipe" "8192 1000 150 pipe" "32768 1000 60 pipe"; do
  echo "== $cfg"
  timeout -k 10 120 /tmp/rx_link $cfg 2>&1 | tail -2
done
