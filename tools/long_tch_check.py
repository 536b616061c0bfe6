import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import torch; torch.cuda.init()
from __graft_entry__ import load_package
pkg=load_package()
import numpy as np, workloads, oracle_lib
api=pkg.api; api.load(); api.init(0)
kc=np.arange(8,dtype=np.uint8)
t=time.time()
b,tc,c,kc,_,_,s9=workloads.bcch_tch_csd_triple(pkg, oracle_lib, 5, seconds=40.0, kc=kc, mix9=(0.2,0.6))
print("gen",time.time()-t)
t=time.time(); rec,big,status,chains=api.rx_run_full(b,tc,c,[0],[b.size],sps=4,kc=kc,max_records=1<<18,max_big=1<<16); print("gpu (first call: kernels and tables load)",time.time()-t,len(rec),len(big))
t=time.time(); rec,big,status,chains=api.rx_run_full(b,tc,c,[0],[b.size],sps=4,kc=kc,max_records=1<<18,max_big=1<<16); print("gpu (second call, host buffers in)",time.time()-t,len(rec),len(big))
t=time.time(); orv,orec,obig,och=oracle_lib.rx_run_full(b,tc,c,kc=kc,max_records=1<<18,max_big=1<<16); print("cpu",time.time()-t,len(orec),len(obig))
k=lambda r:[(int(x["type"]),int(x["fn"]),int(x["tn"]),int(x["len"]),bytes(x["l2"][:int(x["len"])])) for x in r]
print("records equal", k(rec)==k(orec), "big equal", k(big)==k(obig))
