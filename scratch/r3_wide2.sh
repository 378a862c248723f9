#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_filter.py -x -q 2>&1 | tail -8
bash scratch/r3_cfg_prof.sh 5 32
