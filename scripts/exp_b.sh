#!/bin/bash
mkdir -p /tmp/ab && cp ab/*.so /tmp/ab/
CELLMAP_HIP_LIB=/tmp/ab/B.so timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -v > gpurun_out/b_tests.log 2>&1; grep -n "PASSED\|FAILED\|Fatal\|fault\|test_" gpurun_out/b_tests.log | tail -12
