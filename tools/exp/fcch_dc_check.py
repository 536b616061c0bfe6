"""FCCH rough sweep on streams with a large DC offset: toa of the library (whatever GMR1_HIP_LIBRARY / switches say) and of the oracle."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
import torch; torch.cuda.init()
pkg.api.load(); pkg.api.init(0)
import oracle_lib
SPS = 4
rng = np.random.default_rng(23)
n, ns = 6, 93600
x = np.zeros((n, ns), np.complex64)
for i in range(n):
    x[i], _ = pkg.synth.synth_fcch_stream(ns, SPS, rng, snr_db=float(rng.choice((0.0, 6.0))), cfo_hz=float(rng.uniform(-2000, 2000)))
x[1] += np.complex64(7.0 - 3.0j)
x[2] += np.complex64(-40.0 + 25.0j)
x[4] *= np.float32(1e-3)
toa, rv = pkg.api.fcch_rough_batch(x, (np.arange(n) * ns).astype(np.uint64), ns, sps=SPS)
want = [oracle_lib.fcch_rough(x[i], SPS) for i in range(n)]
print(json.dumps({"lib": os.environ.get("GMR1_HIP_LIBRARY", "product"), "switch": {k: v for k, v in os.environ.items() if k.startswith("GMR1_HIP_FCCH")},
                  "toa": [int(t) for t in toa], "rv": [int(r) for r in rv], "oracle": [int(w[1]) for w in want]}))

try:
    import ctypes as C
    f = pkg.api.load().gmr1_hip_prof_fold_dbg
    buf = np.zeros((2, 64, 4), np.float32)
    f(buf.ctypes.data_as(C.c_void_p))
    print("fold   tiles:", buf[0][:12].tolist())
    print("energy group:", buf[1][:12].tolist())
except AttributeError:
    pass
