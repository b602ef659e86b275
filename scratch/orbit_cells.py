"""How well do the cells the sea-ice interface solve abandons at maxiter agree with the oracle in VALUE?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import util
from test_gpu_parity import run_ice
for (nx, ny, h, cfg, scheme) in ((90, 40, 3, "sea_ice_corrected", 0), (90, 40, 3, "sea_ice_ncar", 0), (90, 40, 3, "sea_ice_default", 0),
                                 (131, 67, 3, "sea_ice_corrected", 0), (1440, 560, 7, "sea_ice_corrected", 0), (1440, 560, 7, "sea_ice_corrected", 1)):
    got, ref = run_ice(util.build_case(nx, ny, h, h), cfg, scheme=scheme)
    unconv = np.asarray(ref["iterations"]) >= 100
    n = int(unconv.sum())
    if not n:
        print(nx, ny, cfg, scheme, "no abandoned cells"); continue
    worst = np.zeros(unconv.shape)
    for k in util.ICE_FLUX_FIELDS:
        g, r = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        worst = np.maximum(worst, np.abs(g - r) / np.maximum(np.abs(r), util.FIELD_SCALE[k]))
    e = worst[unconv]
    print(nx, ny, cfg, scheme, "abandoned", n, "of", int((np.asarray(ref["iterations"]) > 0).sum()),
          " <=1e-9: %.4f  <=1e-6: %.4f  <=1e-3: %.4f  max %.2e" % ((e <= 1e-9).mean(), (e <= 1e-6).mean(), (e <= 1e-3).mean(), e.max()))
