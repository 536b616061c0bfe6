import sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from __graft_entry__ import load_package
import workloads
pkg = load_package(); api = pkg.api; api.load(); api.init(0)
dev = torch.device("cuda", 0)
n = 1024
wl = workloads.fcch_streams(pkg, n, seed=2); ns = wl["n_samples"]
iq = torch.from_numpy(wl["iq"].view(np.float32)).to(dev)
offset = torch.from_numpy(wl["offset"].astype(np.int64)).to(dev)
toa = torch.zeros(n, dtype=torch.int32, device=dev); rv = torch.zeros(n, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
f_fine = api.load().gmr1_hip_fcch_fine_batch_dev; f_fine.restype = C.c_int
ftoa = torch.zeros(n, dtype=torch.int32, device=dev); ferr = torch.zeros(n, dtype=torch.float32, device=dev)
def rough():
    api.fcch_rough_batch_dev(stream.cuda_stream, "fcch", n, 4, ns, iq.data_ptr(), offset.data_ptr(), None, toa.data_ptr(), rv.data_ptr())
def fine(off_f):
    f_fine(C.c_void_p(stream.cuda_stream), C.c_int(0), C.c_int(n), C.c_int(4), C.c_void_p(iq.data_ptr()), C.c_void_p(off_f.data_ptr()), None, C.c_void_p(ftoa.data_ptr()), C.c_void_p(ferr.data_ptr()))
def t(fn, k=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3
print("rough: host ms/call %.3f, total %.3f" % t(rough))
off_f = offset + torch.clamp(toa.to(torch.int64), 0, ns - 468)
print("fine: host %.3f total %.3f" % t(lambda: fine(off_f)))
print("torch ops: host %.3f total %.3f" % t(lambda: offset + torch.clamp(toa.to(torch.int64), 0, ns - 468)))
def step():
    rough()
    off = offset + torch.clamp(toa.to(torch.int64), 0, ns - 468)
    fine(off)
print("rough + ops + fine: host %.3f total %.3f" % t(step))
def step2():
    rough()
    fine(off_f)
print("rough + fine (fixed offsets): host %.3f total %.3f" % t(step2))
# where a one-off stall inside a timed loop comes from: per-step times with a synchronise after each
import bench
bench.preroll(step, 0.3)
for _ in range(10): step()
torch.cuda.synchronize()
ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
ts = []
t0 = time.perf_counter(); ev0.record(stream)
for i in range(100):
    step()
    if i < 12:
        ts.append(time.perf_counter() - t0)
ev1.record(stream); torch.cuda.synchronize()
print("host time after steps 0..11 (ms):", [round(x * 1e3, 3) for x in ts])
print("100 steps: wall %.3f ms/step, events %.3f ms/step" % ((time.perf_counter() - t0) * 10, ev0.elapsed_time(ev1) / 100))
