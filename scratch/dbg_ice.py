import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "climaocean.jl_amd")]
import numpy as np
import util, test_gpu_parity as t
cfg = sys.argv[1] if len(sys.argv) > 1 else "sea_ice_ncar"
case = util.build_case(90, 40)
got, ref = t.run_ice(case, cfg)
k = "friction_velocity"
err = np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-3)
conv = ref["iterations"] < 100
err_c = np.where(conv, err, 0)
idx = np.dstack(np.unravel_index(np.argsort(-err_c.ravel())[:8], err.shape))[0]
W = lambda a: util.window(a, 3, 3, 90, 40, 1)
for (j, i) in idx:
    print(j, i, "it", ref["iterations"][j, i], got["iterations"][j, i], "err", err[j, i],
          "us", ref[k][j, i], got[k][j, i], "ts", ref["temperature_scale"][j, i], got["temperature_scale"][j, i],
          "Ts", ref["temperature"][j, i], got["temperature"][j, i], "h", W(case["ice_state"]["thickness"])[j, i])
