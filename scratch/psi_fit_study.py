"""(segments per binade, degree) -> max |error| / max(|psi|, 1) of the piecewise polynomial psi tables (scratch, CPU)."""
import sys, os
import numpy as np
from numpy.polynomial import chebyshev as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy_oracle as no

def err(name, fn, unstable, S, d, b0, b1):
    worst = 0.0
    for b in range(b0, b1):
        for s in range(S):
            w = 2.0 ** b / S
            x0 = 2.0 ** b + s * w
            f = lambda t: (no.psi_h if fn else no.psi_m)(name, (-1 if unstable else 1) * np.maximum((x0 - 1 + 0.5 * (t + 1) * w) / 16.0, 0 if not unstable else 1e-300))
            c = C.chebinterpolate(f, d)
            tt = np.linspace(-1, 1, 400)
            e = np.abs(C.chebval(tt, c) - f(tt)) / np.maximum(np.abs(f(tt)), 1.0)
            worst = max(worst, e.max())
    return worst

for name in ("edson2013", "sheba", "large_yeager"):
    print(name)
    for (S, d) in ((4, 9), (4, 8), (4, 7), (4, 6), (8, 7), (8, 6), (8, 5), (16, 6), (16, 5), (16, 4), (32, 4)):
        e = max(err(name, fn, un, S, d, 0, 12) for fn in (0, 1) for un in (False, True))
        print("  fine tier (x < 4096) S=%2d deg=%d: %.1e   bytes/side-pair: %d" % (S, d, e, 12 * S * 2 * (d + 1) * 16))
    for (S, d) in ((1, 9), (1, 7), (1, 6), (1, 5), (2, 7), (2,5)):
        e = max(err(name, fn, un, S, d, 12, 34) for fn in (0, 1) for un in (False, True))
        print("  coarse tier (x >= 4096) S=%d deg=%d: %.1e" % (S, d, e))
