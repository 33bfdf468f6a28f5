"""Generates tests/golden/idwt_pywt.npz with the REAL PyWavelets (pywt.idwt2, mode='zero').

Run in this container only:  /opt/conda/bin/python3.9 tests/golden/make_golden_pywt.py
(PyWavelets is the third-party library underneath the reference's pytorch_wavelets.DWTInverse,
triplane_encoder.py:167,185,394; requirements2.txt:113,115.)

For every wavelet the reference supports (triplane_encoder.py:174-180) it stores: the synthesis
taps, a random LL (2 slices, 8x8) + two levels of detail bands, and the planes obtained by
applying  x <- idwt2((pad(2x), (pad(cH), pad(cV), pad(cD))), mode='zero')  twice, exactly the
loop body of TriPlaneVolume.build_planes (:379,392-394) with yh[:,:,0/1/2] = cH/cV/cD.
"""
import os
import numpy as np
import pywt

PAD = {"bior6.8": 4, "bior2.6": 3, "bior4.4": 2, "bior2.2": 1, "haar": 0}  # triplane_encoder.py:174-180
rng = np.random.default_rng(1234)
out = {"pywt_version": np.array(pywt.__version__)}
for wave, pad in PAD.items():
    w = pywt.Wavelet(wave)
    out[f"{wave}/rec_lo"] = np.array(w.rec_lo)
    out[f"{wave}/rec_hi"] = np.array(w.rec_hi)
    S, n = 2, 8
    ll = rng.standard_normal((S, n, n))
    out[f"{wave}/ll"] = ll
    x = ll
    for lvl in range(2):
        n = x.shape[-1]
        yh = rng.standard_normal((S, 3, n, n)) * 0.5
        out[f"{wave}/yh{lvl}"] = yh
        nxt = []
        for s in range(S):
            p = lambda a: np.pad(a, pad)
            nxt.append(pywt.idwt2((p(2 * x[s]), (p(yh[s, 0]), p(yh[s, 1]), p(yh[s, 2]))), wave, mode="zero"))
        x = np.stack(nxt)
        assert x.shape[-1] == 2 * n, x.shape
    out[f"{wave}/planes"] = x
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "idwt_pywt.npz"), **out)
print("wrote idwt_pywt.npz", {k: v.shape for k, v in out.items() if k.endswith("planes")})
