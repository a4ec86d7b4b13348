import sys, time, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch, idgrec_amd.host as H, idgrec_amd.synth as S
U,I,E=S.SHAPES["yelp2018"]
users,items=S.generate(U,I,E,seed=0)
pos_ptr=np.zeros(U+1,dtype=np.int64); pos_ptr[1:]=np.cumsum(np.bincount(users,minlength=U))
rng=H.Rng(2024); it32=items.astype(np.int32)
for k in range(4):
    t0=time.time(); tri=rng.sample_epoch(users,items,pos_ptr,it32,I); t1=time.time(); p=rng.shuffle_perm(len(tri)); t2=time.time()
    d=torch.from_numpy(tri).cuda(); pd=torch.from_numpy(p).cuda(); u=d[:,0][pd].contiguous(); torch.cuda.synchronize(); t3=time.time()
    print("sample %.1f ms, shuffle perm %.1f ms, H2D+gather %.1f ms"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
