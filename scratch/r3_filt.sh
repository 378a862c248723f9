#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_filter.py -x -q 2>&1 | tail -6
