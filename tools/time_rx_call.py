import sys, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from __graft_entry__ import load_package
import workloads
pkg = load_package(); api = pkg.api; api.load(); api.init(0)
A, sps, secs = 64, 4, 60.0
ns = int(secs * 23400 * sps)
host = [workloads.bcch_carrier(pkg, 700 + a, seconds=secs, sps=sps, stn=(5 * a) % 24, delay=a % 8, cfo_hz=40.0 * (a - 3), esn0_db=10.0 + a)[0] for a in range(8)]
base = torch.from_numpy(np.concatenate(host).view(np.float32)).cuda()
iq = torch.cat([base] * 8)[:A * ns * 2].contiguous()
offset = np.arange(A, dtype=np.uint64) * np.uint64(ns); length = np.full(A, ns, np.uint64)
st = torch.cuda.current_stream().cuda_stream
out = np.empty(1 << 18, api.RX_RECORD); n_rec = C.c_int(0); status = np.zeros(A, np.int32); chains = np.zeros(A, np.int32)
f = api.load().gmr1_hip_rx_run_dev; f.restype = C.c_int
def call():
    return f(C.c_void_p(st), C.c_int(A), C.c_int(sps), C.c_void_p(iq.data_ptr()), offset.ctypes.data_as(C.c_void_p), length.ctypes.data_as(C.c_void_p), None,
             out.ctypes.data_as(C.c_void_p), C.c_int(1 << 18), C.byref(n_rec), status.ctypes.data_as(C.c_void_p), chains.ctypes.data_as(C.c_void_p))
for _ in range(3): call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): call()
torch.cuda.synchronize()
print("C call only: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), n_rec.value)
t0 = time.perf_counter()
for _ in range(10): api.rx_run_dev(st, iq.data_ptr(), offset, length, sps=sps, max_records=1 << 18)
torch.cuda.synchronize()
print("api.rx_run_dev: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
