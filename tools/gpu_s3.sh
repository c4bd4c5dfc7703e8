#!/bin/bash
mkdir -p gpurun_out/s3
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/s3/tests.txt
tail -8 gpurun_out/s3/tests.txt
# forward pass: chunk loads / stores through scalar bases (FOA_TRIM 1, the product) against the compiler's addressing (0)
tools/ab_libs.sh trim 3 fun_ofdm_amd/csrc/libfun_ofdm_amd.so build/var_notrim.so > gpurun_out/s3/ab_trim.txt 2>&1
cat gpurun_out/s3/ab_trim.txt
