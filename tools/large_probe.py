"""extra.large by itself: the regime above the single launch (workloads.run_large).  usage: large_probe.py [nx]"""
import json
import sys

sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen, workloads

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print(json.dumps(workloads.run_large(capi, meshgen, nx=nx), indent=1))
