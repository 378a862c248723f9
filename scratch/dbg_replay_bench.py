"""per-wave phase cycles of the selection kernels on the bench workload: AUNCEL_AMD_DEBUG_REPLAY=1 (sync rounds)"""
import os, sys
os.environ["AUNCEL_AMD_DEBUG_REPLAY"] = "1"
sys.argv = [sys.argv[0], "--no-cpu", "--steps", "1", "--warmup", "0", "--in-flight", "1"]
sys.path.insert(0, "/root/repo")
import bench
bench.main()
