import os, sys, numpy as np, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import conftest
from orion_amd import api
g = torch.Generator(device="cuda"); g.manual_seed(7)
for name in sys.argv[1:]:
    model = conftest.load_model(name); sc = api.Scene(model, device=0)
    tris = torch.from_numpy(np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles])).cuda()
    lo = tris.reshape(-1,3).min(0).values; hi = tris.reshape(-1,3).max(0).values
    n = 1<<22
    hist = np.zeros(8, np.int64); tot = np.zeros(8, np.int64)
    for r in range(int(os.environ.get("ROUNDS","16"))):
        pick = torch.randint(0, len(tris), (n,), device="cuda", generator=g); tp = tris[pick]
        e1 = tp[:,1]-tp[:,0]; e2 = tp[:,2]-tp[:,0]
        nr = torch.linalg.cross(e1,e2); nr = nr/nr.norm(dim=1,keepdim=True).clamp_min(1e-30)
        a = torch.rand(n, device="cuda", generator=g)*1.4-0.2; b = torch.rand(n, device="cuda", generator=g)*1.4-0.2
        p = tp[:,0]+a[:,None]*e1+b[:,None]*e2
        ang = torch.rand(n, device="cuda", generator=g)*6.2831853
        u1 = e1/e1.norm(dim=1,keepdim=True).clamp_min(1e-30); u2 = torch.linalg.cross(nr,u1)
        inpl = torch.cos(ang)[:,None]*u1+torch.sin(ang)[:,None]*u2
        ls = -8.0+8.0*torch.rand(n, device="cuda", generator=g)
        slope = 10.0**ls
        d = inpl + slope[:,None]*nr
        back = (hi-lo).max()*10.0**(-2.0+2.5*torch.rand(n, device="cuda", generator=g))
        o = (p - back[:,None]*d).float().contiguous(); d = d.float().contiguous()
        sc.set_option("traversal",1); want = sc.closest_hit(o,d).clone()
        sc.set_option("traversal",4); got = sc.closest_hit(o,d)
        bad = (want!=got).any(1)
        bins = torch.clamp((ls+8).long(),0,7)
        tot += torch.bincount(bins, minlength=8).cpu().numpy()
        hist += torch.bincount(bins[bad], minlength=8).cpu().numpy()
    print(name, "slope decade 1e-8..1e0: differing / total")
    for k in range(8): print("   1e%d..1e%d: %d / %d  (%.2e)" % (k-8,k-7,hist[k],tot[k],hist[k]/max(1,tot[k])))
    sc.close()
