import sys, os, json, time, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensorrec_amd as T
n=1_000_000; d=128; S=100
rng=np.random.default_rng(0)
cols=rng.integers(0,n,size=(n,20),dtype=np.int32)
inter=sp.csr_matrix((np.ones(n*20,np.float32),cols.reshape(-1),np.arange(0,(n+1)*20,20,dtype=np.int64)),shape=(n,n)); inter.sum_duplicates(); inter.data[:]=1
uf=sp.identity(n,dtype=np.float32,format="csr"); itf=sp.identity(n,dtype=np.float32,format="csr")
configs=[dict(kv.split("=") for kv in a.split(",")) if a else {} for a in sys.argv[1:]] or [{}]
for knobs in configs:
    for k,v in knobs.items(): T._native.set_tuning(k,int(v))
    m=T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
    m.fit_partial(inter,uf,itf,epochs=1,n_sampled_items=S); torch.cuda.synchronize()
    t0=time.perf_counter(); m.fit_partial(inter,uf,itf,epochs=1,n_sampled_items=S); torch.cuda.synchronize(); one=time.perf_counter()-t0
    t0=time.perf_counter(); m.fit_partial(inter,uf,itf,epochs=5,n_sampled_items=S); torch.cuda.synchronize(); five=time.perf_counter()-t0
    print(knobs, "ms/epoch %.2f"%((five-one)/4*1e3), flush=True)
    del m; torch.cuda.empty_cache()
