"""AKAZE61 (config #5: 1280 x 720) throughput — thin wrapper over `bench.py --workload akaze61` (kept for the command lines quoted in
DESIGN.md / profiles).  usage: python tools/bench_akaze.py [batch] [steps] [cpu_frames]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = sys.argv[1] if len(sys.argv) > 1 else "16"
steps = sys.argv[2] if len(sys.argv) > 2 else "10"
cpu = sys.argv[3] if len(sys.argv) > 3 else "0"
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "akaze61", "--batch", batch, "--steps", steps, "--warmup", "2",
                          "--cpu-frames", cpu]))
