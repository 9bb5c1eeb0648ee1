"""All-cores CPU baseline (oracle/fem_oracle_mt.c) under the OMP_* settings of the environment: python tools/cpu_mt_probe.py [nx]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
f, b, _ = bench.cpu_baseline(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
print(os.environ.get("OMP_NUM_THREADS"), os.environ.get("OMP_PROC_BIND"), "| faithful %.0f DOF/s | all-cores %s DOF/s | %s" % (f["value"], b.get("value"), b.get("sample", b.get("error"))[12:110]))
