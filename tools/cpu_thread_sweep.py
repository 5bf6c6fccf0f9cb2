"""Thread-count sweep of the CPU baseline (bench.cpu_baseline: the oracle's train step on the host) -> JSON for profiles/.
usage: cpu_thread_sweep.py [threads ...]   (default 16 32 64 128)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

counts = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]
rows = []
for t in counts:
    r = bench.cpu_baseline(544, 960, 25, "ocrnet_hrnet48", t)
    rows.append({"threads": r["cores"], "frames_per_s": r["value"], "frames_per_s_anomaly_mode": r["value_anomaly_mode_on"], "sample": r["sample"]})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"cpu_model": bench.host_cpu_info()[0], "physical_cores": bench.host_cpu_info()[1], "sweep": rows}))
