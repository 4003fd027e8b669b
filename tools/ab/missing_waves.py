import os, sys, time, torch, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
from orion_amd import api
fr = api.load_frame(os.path.join(ROOT,"configs","c3_nanosuit_fhd.rtc"), load_textures=False)
W,H = 1920,1080
sc = api.Scene(fr.model, device=0)
acc = torch.zeros((H,W,4), dtype=torch.float32, device="cuda")
cams = {"frame": fr.camera12, "away": api.camera_vectors([0,7.5,17],[0,7.5,34],[0,1,0],1.0,16/9)}
for name, cam in cams.items():
    sc.render(cam, W,H,0,32,1,0,seed=1,lights=fr.rtc.lights,accum=acc)
    sc.profile(True)
    sc.render(cam, W,H,0,64,1,0,seed=1,lights=fr.rtc.lights,accum=acc)
    st = sc.profile_read(); sc.profile(False)
    print(name, {k:(round(v['ms'],3), v['launches'], v['items']) for k,v in st.items() if v['launches']}, "extend G rays/s %.1f" % (st['closest_hit']['items']/st['closest_hit']['ms']/1e6))
