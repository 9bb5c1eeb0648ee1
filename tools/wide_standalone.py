import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
r = workloads.run_wide(capi, meshgen)
print("run_wide standalone:", {k: r[k] for k in ("us_per_iteration_in_launch", "frac", "iterations", "operator_phase_us", "gather_avg_us")})
