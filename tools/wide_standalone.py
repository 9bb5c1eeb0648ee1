"""workloads.run_wide (bench.py's extra.wide_2p35M: 3-D P1, 2.35 M DOFs through the wide single launch) on its own: is the figure inside the bench line an ordering effect?"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
r = workloads.run_wide(capi, meshgen)
print("run_wide standalone:", {k: r[k] for k in ("us_per_iteration_in_launch", "frac", "iterations", "operator_phase_us", "gather_avg_us")})
