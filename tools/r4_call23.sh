#!/bin/bash
# round 4, call 23: what do the per-frame loads / stores of the cooperative BiLSTM kernels cost beside the exchange?
set -e
mkdir -p gpurun_out/r4w
cd /root/repo
timeout -k 10 120 python tools/lstm_probe.py > gpurun_out/r4w/lstm_product.txt 2>gpurun_out/r4w/err.txt
cat gpurun_out/r4w/lstm_product.txt
for v in 3 12; do
  APTAI_HIP_LIB=$PWD/tools/ab/lstm/lib_lstm_$v.so timeout -k 10 120 python tools/lstm_probe.py > gpurun_out/r4w/lstm_$v.txt 2>>gpurun_out/r4w/err.txt
  cat gpurun_out/r4w/lstm_$v.txt
done
