#!/bin/bash
mkdir -p gpurun_out; timeout 900 python -m pytest tests -m gpu -x -q -k "front_door" 2>&1 | tail -6 > gpurun_out/c32.log; cat gpurun_out/c32.log
