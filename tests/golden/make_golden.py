"""Writes tests/golden/known_answers.json: the known-answer data the reference's own tests hold for this path.

The reference ships no bitwise golden vectors.  Its tests are analytic known answers:
  * Frank matrix a_ij = min(i,j): eigenvalues 1/(2(1-cos((2k-1)pi/(2n+1))))  (benchmark/mat_set.f:117-132,
    :638-647; gate: max relative error < sqrt(eps), benchmark/w_test.f:141-151)
  * C binding smoke matrix [[-2,1],[1,-2]] -> eigenvalues -3, -1            (C/c_test.c:5-77)
  * accuracy gates 768 / 8                                                    (benchmark/ev_test.f:181-204)
Run:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np

out = {"frank": {}, "c_test": {"matrix": [[-2.0, 1.0], [1.0, -2.0]], "eigenvalues": [-3.0, -1.0]},
       "gates": {"residual": 768.0, "orthogonality": 8.0, "frank_rel_err": float(np.sqrt(np.finfo(float).eps))}}
for n in (3, 4, 5, 7, 64, 200, 255, 256, 257, 1000, 1024):
    k = np.arange(1, n + 1)
    lam = np.sort(1.0 / (2.0 * (1.0 - np.cos((2 * k - 1) * np.pi / (2 * n + 1)))))
    out["frank"][str(n)] = [float(lam[0]), float(lam[n // 2]), float(lam[-1]), float(lam.sum())]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json"), "w") as f:
    json.dump(out, f, indent=1)
