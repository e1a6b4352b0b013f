import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import matchingpursuit as mp, _native as nat, synth, encode_packed
A, L, N, B, K = 512, 512, 32768, 64, 64
d = torch.from_numpy(synth.make_dictionary(A, L, seed=1000)).cuda()
x = torch.from_numpy(synth.make_segments(B, N, synth.make_dictionary(A, L, seed=1000), n_events=192, seed=1002)).cuda()[:, None, :]
def T(f, n=20):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        t0=time.perf_counter(); o=f(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    return float(np.median(ts)), o
du = nat.unit_norm(d)
t, _ = T(lambda: nat.unit_norm(d)); print(f"unit_norm {t:.3f}")
t, _ = T(lambda: mp._dict_unit(d, d.device)); print(f"_dict_unit {t:.3f}")
t, o = T(lambda: nat.encode(x[:,0,:], du, K, path=nat.MP_PATH_FFT)); print(f"encode (packed, with residual) {t:.3f}")
t, o = T(lambda: nat.encode(x[:,0,:], du, K, path=nat.MP_PATH_FFT, want_residual=False)); print(f"encode (no residual) {t:.3f}")
t, o = T(lambda: nat.encode_checked(x[:,0,:], du, K)); print(f"encode_checked {t:.3f}")
atom, lag, gain, res = o
t, st = T(lambda: mp._EventStore(atom, lag, gain, du, x.device, A)); print(f"_EventStore {t:.3f}")
t, fl = T(lambda: st.flat(eager=False)); print(f"flat {t:.3f}")
sc = mp.build_scatter_segments(N, L)
t, _ = T(lambda: sc(x.shape, fl)); print(f"scatter(shape, events) {t:.3f}")
t, _ = T(lambda: mp.sparse_code(x, d, n_steps=K, flatten=True)); print(f"sparse_code flatten {t:.3f}")
def full():
    ev, s = mp.sparse_code(x, d, n_steps=K, flatten=True); return s(x.shape, ev)
t, _ = T(full); print(f"sparse_code + scatter {t:.3f}")
t, _ = T(lambda: encode_packed(x, d, K)); print(f"encode_packed {t:.3f}")
