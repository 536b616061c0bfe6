import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
from __graft_entry__ import load_package
import workloads
pkg = load_package(); api = pkg.api; api.load(); api.init(0)
for sps in (8, 5):
    wl = workloads.bcch_ccch_mix(pkg, n=20000, seed=3, sps=sps)
    dev = torch.device('cuda', 0)
    iq = torch.from_numpy(wl['iq'].view(np.float32)).to(dev); off = torch.from_numpy(wl['offset'].astype(np.int64)).to(dev); kind = torch.from_numpy(wl['kind']).to(dev)
    n = len(wl['kind'])
    l2 = torch.zeros((n,24),dtype=torch.uint8,device=dev); crc=torch.zeros(n,dtype=torch.int32,device=dev); conv=torch.zeros(n,dtype=torch.int32,device=dev)
    toa=torch.zeros(n,dtype=torch.float32,device=dev); fe=torch.zeros(n,dtype=torch.float32,device=dev); rv=torch.zeros(n,dtype=torch.int32,device=dev)
    st = torch.cuda.current_stream(dev)
    def step(): api.rx_bcch_ccch_batch_dev(st.cuda_stream, n, sps, iq.data_ptr(), off.data_ptr(), kind.data_ptr(), None, l2.data_ptr(), crc.data_ptr(), conv.data_ptr(), toa.data_ptr(), fe.data_ptr(), None, None, rv.data_ptr())
    for _ in range(20): step()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
    good = (crc.cpu().numpy()==0)
    print('sps', sps, 'n', n, 'ms', round(dt*1e3,4), 'Mbursts/s', round(n/dt/1e6,1), 'crc pass', good.mean(), 'payload ok', bool(np.array_equal(l2.cpu().numpy()[good], wl['l2'][good])))
