import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import conftest, torch
from orion_amd import api
for name in ("nanosuit","soup","hindu","soup20"):
    m = conftest.load_model(name)
    torch.cuda.synchronize(); t0=time.time(); sc = api.Scene(m, device=0); torch.cuda.synchronize(); dt=time.time()-t0
    print(name, "scene create %.3f s"%dt, "free GiB %.1f"%(torch.cuda.mem_get_info()[0]/2**30)); sc.close()
