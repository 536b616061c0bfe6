import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from __graft_entry__ import load_package
import ambe_streams as S, oracle_lib
pkg = load_package(); api = pkg.api; api.load(); api.init(0)
g = np.load('tests/golden/ambe_vectors.npz')
fr = g['mixed_frames']; pcm,_,_ = api.codec_decode_batch(fr[None]); d = pcm[0].astype(int) - g['mixed_pcm']
for f, i in zip(*np.nonzero(d)): print('mixed: frame', f, 'sample', i, 'type', hex(fr[f,0] & 0xfc), 'got', pcm[0][f,i], 'want', g['mixed_pcm'][f,i])
tot = {}
for c in range(96):
    n_fr = 150
    x = S.mixed_stream(n_fr, 500 + c, invalid_tones=(c % 5 == 0)) if c % 3 else S.random_stream(n_fr, 500 + c)
    p,_,_ = api.codec_decode_batch(x[None]); w,_ = oracle_lib.ambe_decode(x)
    for f, i in zip(*np.nonzero(p[0].astype(int) - w)): 
        t = hex(x[f,0] & 0xfc) if (x[f,0]&0xfc) in (0xfc,0xf8) else 'speech'
        tot[t] = tot.get(t,0)+1
print(tot)
