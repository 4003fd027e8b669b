import os, sys, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
from orion_amd import api
fr = api.load_frame(os.path.join(ROOT,"configs","c3_nanosuit_fhd.rtc"), load_textures=False)
W,H = 1920,1080
sc = api.Scene(fr.model, device=0)
acc = torch.zeros((H,W,4), dtype=torch.float32, device="cuda")
for rep in range(2):
  for wa in (1,0):
    sc.set_option("wave_ahead", wa)
    sc.render(fr.camera12, W,H,0,64,1,0,seed=1,lights=fr.rtc.lights,accum=acc)
    sc.profile(True)
    sc.render(fr.camera12, W,H,0,64,1,0,seed=1,lights=fr.rtc.lights,accum=acc)
    st = sc.profile_read(); sc.profile(False)
    print("wave_ahead", wa, {k:round(v['ms'],3) for k,v in st.items() if v['launches']})
