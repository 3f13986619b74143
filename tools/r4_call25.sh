#!/bin/bash
set -e
mkdir -p gpurun_out/r4C
cd /root/repo
for r in 1 2; do
timeout -k 10 120 python tools/lstm_probe.py 2>> gpurun_out/r4C/err.txt | tee -a gpurun_out/r4C/lstm.txt
APTAI_HIP_LIB=$PWD/tools/ab/lstm/lib_lstm_bwdorig.so timeout -k 10 120 python tools/lstm_probe.py 2>> gpurun_out/r4C/err.txt | tee -a gpurun_out/r4C/lstm.txt
done
