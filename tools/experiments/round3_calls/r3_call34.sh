#!/bin/bash
mkdir -p gpurun_out; timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s -k "attention_maps" 2>&1 | tail -12 > gpurun_out/c34.log; cat gpurun_out/c34.log
