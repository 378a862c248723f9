"""one adaptive search of the bench workload with the replay debug counters (AUNCEL_AMD_DEBUG_REPLAY=1)"""
import sys, os, subprocess
os.environ["AUNCEL_AMD_DEBUG_REPLAY"] = "1"
sys.argv = ["bench.py", "--no-cpu", "--steps", "1", "--warmup", "0"]
sys.path.insert(0, "/root/repo")
import runpy
runpy.run_path("/root/repo/bench.py", run_name="__main__")
