import sys, os, time
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import numpy as np
import bench
if os.environ.get("PRE"):
    a = bench.parse()
    bench.run_hot_loop(a, 1, 0, 0, "w1", 64, 1, False, 20, 5)
r = bench.run_per_frame_classes(0)
print({k: r[k] for k in ("ms_per_frame_median", "yolo_mode_detect_ms_median", "stage_ms_median")})
