import sys; sys.path.insert(0,'/root/repo')
import numpy as np, faiss_amd
from faiss_amd.datasets import synthetic_dataset
res=faiss_amd.StandardGpuResources(0)
for (nb,nq,k) in ((20000,64,100),(30000,300,10),(30000,2500,100)):
    _,xb,xq=synthetic_dataset(128,0,nb,nq,seed=5)
    idx=faiss_amd.GpuIndexFlatL2(res,128); idx.add(xb)
    D,I=idx.search(xq,k); st=idx.filter_stats()
    S,eb=idx.filter_scores(xq[:64])
    idx.set_use_filter_kernel(False); D0,I0=idx.search(xq,k)
    ex=(xq[:64].astype(np.float64)@xb.T.astype(np.float64))-0.5*(xb.astype(np.float64)**2).sum(1)[None,:]
    err=np.abs(S-ex); bad=np.argwhere(err>1.0)
    print(nb,nq,k,"stats",st,"equal",np.array_equal(I,I0),np.array_equal(D,D0),"dump maxerr",err.max(),"nbad",len(bad),bad[:6].tolist(),flush=True)
    if len(bad):
        rows=np.unique(bad[:,1]); print(" bad rows mod 64:",np.unique(rows%64)[:40], "bad q", np.unique(bad[:,0])[:20])
