#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "page_locked" 2>&1 | grep -v "^$" | tail -40
