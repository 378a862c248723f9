#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "asynchronous" 2>&1 | tail -3
