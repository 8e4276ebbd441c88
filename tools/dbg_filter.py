import sys; sys.path.insert(0,'/root/repo')
import numpy as np, faiss_amd
from faiss_amd.datasets import synthetic_dataset
res=faiss_amd.StandardGpuResources(0)
for (nb,nq,k) in ((30000,2500,10),(30000,2500,100),(30000,300,100),(200000,4096,100),(1000000,10000,100)):
    _,xb,xq=synthetic_dataset(128,0,nb,nq,seed=5)
    idx=faiss_amd.GpuIndexFlatL2(res,128); idx.add(xb)
    D,I=idx.search(xq,k); st=idx.filter_stats()
    idx.set_use_filter_kernel(False); D0,I0=idx.search(xq,k)
    print(nb,nq,k,"stats",st,"equal",np.array_equal(I,I0),np.array_equal(D,D0),flush=True)
