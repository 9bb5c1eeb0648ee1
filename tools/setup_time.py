import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["FDAPDE_DEBUG_SETUP"]="1"
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import meshgen, capi
nx=int(sys.argv[1]); order=int(sys.argv[2])
nodes,cells,bnd=meshgen.unit_cube(nx)
c=capi.Context(device=None)
t=time.time(); c.mesh_upload(nodes,cells,bnd); t1=time.time(); c.dofs_build(order); t2=time.time()
print("upload %.3f s  dofs_build %.3f s"%(t1-t,t2-t1), "cpus", os.cpu_count())
