#!/bin/bash
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "asynchronous" 2>&1 | tail -3
bash scratch/r3_async4.sh
