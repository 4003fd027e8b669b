import os, sys, numpy as np, torch
ROOT='/root/repo' if os.path.exists('/root/repo/orion_amd') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from orion_amd import api
fr = api.load_frame(os.path.join(ROOT,'configs','c3_nanosuit_fhd.rtc'), load_textures=False)
W,H = 1920,1080
sc = api.Scene(fr.model, device=0)
c = np.asarray(fr.camera12, np.float32); eye, front, right, up = c[0:3], c[3:6], c[6:9], c[9:12]
jj, ii = np.meshgrid(np.arange(W), np.arange(H))
x = (2*(jj.astype(np.float32)/W)-1 + 0.3/W).astype(np.float32).ravel(); y = (2*(ii.astype(np.float32)/H)-1+0.3/H).astype(np.float32).ravel()
d = (front[None,:] + x[:,None]*right[None,:] + (-y)[:,None]*up[None,:]).astype(np.float32)
o = np.repeat(eye[None,:], len(d), 0).astype(np.float32)
to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
out = torch.empty((len(o),5), dtype=torch.int32, device='cuda')
res = {}
for mode in (2, 4, 0):
    sc.set_option('traversal', mode)
    sc.closest_hit(to, td, out); sc.profile(True)
    for _ in range(5): sc.closest_hit(to, td, out)
    st = sc.profile_read()['closest_hit']; sc.profile(False)
    res[mode] = out.clone()
    print('mode', mode, 'pixel-order primary', len(o), 'rays', round(st['ms']/5,3), 'ms per batch', round(5*len(o)/st['ms']/1e3), 'Mrays/s', 'hit', float((out[:,0]!=-1).float().mean()), 'same', torch.equal(res[2], out))
# only the rays whose wave (64 consecutive) has any hit
hit = (res[2][:,0] != -1).view(-1, 64).any(1).repeat_interleave(64)
to2, td2 = to[hit].contiguous(), td[hit].contiguous(); out2 = torch.empty((len(to2),5), dtype=torch.int32, device='cuda')
for mode in (2, 4):
    sc.set_option('traversal', mode)
    sc.closest_hit(to2, td2, out2); sc.profile(True)
    for _ in range(5): sc.closest_hit(to2, td2, out2)
    st = sc.profile_read()['closest_hit']; sc.profile(False)
    print('mode', mode, 'hitting waves only', len(to2), 'rays', round(st['ms']/5,3), 'ms', round(5*len(to2)/st['ms']/1e3), 'Mrays/s')
# ... and many samples of them in one batch (a frame's batch holds 32 samples per pixel): is the launch of 0.3 M rays above just short?
rep = 16
jit = (torch.rand((rep, 1, 3), device='cuda') - 0.5) * 2e-4
to3 = to2.unsqueeze(0).expand(rep, -1, -1).reshape(-1, 3).contiguous()
td3 = (td2.unsqueeze(0) + jit).reshape(-1, 3).contiguous(); out3 = torch.empty((len(to3),5), dtype=torch.int32, device='cuda')
for mode in (2, 4, 0):
    sc.set_option('traversal', mode)
    sc.closest_hit(to3, td3, out3); sc.profile(True)
    for _ in range(5): sc.closest_hit(to3, td3, out3)
    st = sc.profile_read()['closest_hit']; sc.profile(False)
    print('mode', mode, 'hitting waves x', rep, len(to3), 'rays', round(st['ms']/5,3), 'ms', round(5*len(to3)/st['ms']/1e3), 'Mrays/s')
# ... and the whole frame's rays, sixteen samples in one batch, as the first bounce of a render batch holds them
jit = (torch.rand((rep, 1, 3), device='cuda') - 0.5) * 2e-4
to4 = to.unsqueeze(0).expand(rep, -1, -1).reshape(-1, 3).contiguous()
td4 = (td.unsqueeze(0) + jit).reshape(-1, 3).contiguous(); out4 = torch.empty((len(to4),5), dtype=torch.int32, device='cuda')
for mode in (2,):
    sc.set_option('traversal', mode)
    sc.closest_hit(to4, td4, out4); sc.profile(True)
    for _ in range(5): sc.closest_hit(to4, td4, out4)
    st = sc.profile_read()['closest_hit']; sc.profile(False)
    print('mode', mode, 'whole frame x', rep, len(to4), 'rays', round(st['ms']/5,3), 'ms', round(5*len(to4)/st['ms']/1e3), 'Mrays/s')
