"""C5 on one GPU: us per BiCGStab iteration for 1 and 3 timed steps, standalone (is the figure inside bench.py's extras an ordering effect?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
for steps, ts in ((1, 16), (3, 16), (3, 0)):
    r = workloads.run_c5(capi, meshgen, steps=steps, time_spmv=ts)
    print(f"steps {steps} time_spmv {ts}: us/iteration {r['us_per_iteration']:.1f} iterations {r['iterations']} t_solve {r['t_solve_ms']:.1f} ms spmv {r.get('spmv_avg_us')}", flush=True)
