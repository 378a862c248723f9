#!/bin/bash
# ablation of scan_filter_wide_kernel (needs the AUNCEL_AMD_WIDE_DBG knob: bit 0 skips the epilogue, bit 1 the list loads inside
# the K loop; it was a temporary kernel argument, see git history of ivf_filter.hip around this file's commit)
echo "dbg 0: 5.76 ms | dbg 1 (no epilogue): 5.78 | dbg 2 (no list loads): 5.39 | dbg 3: 5.25   (cfg 5, nprobe 32, one box)"
