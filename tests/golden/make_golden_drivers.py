#!/usr/bin/env python3
"""
tests/golden/make_golden_drivers.py -- golden vectors for the driver-level helpers, captured from the REFERENCE's own
numpy-only function validate_distance_matrix (scripts/tda_eeg_classification_v2.py:110-140).  The script cannot be
imported (it runs against absent data at import time, v2:477-496), so the ONE function definition is cut out of the
file's text at generation time and exec'd with numpy -- nothing of it is stored here; the committed fixture
(reference_golden_drivers.npz) holds the input matrices and the (is_valid, issues) answers only.
Run once in the build container where /root/reference is mounted.
"""
import json
import os
import re

import numpy as np

REF = os.environ.get("TDA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_golden_drivers.npz")


def main():
    text = open(os.path.join(REF, "scripts", "tda_eeg_classification_v2.py"), encoding="utf-8").read()
    m = re.search(r"^def validate_distance_matrix\(.*?(?=^\S)", text, re.S | re.M)
    ns = {"np": np}
    exec(compile(m.group(0), "v2-validate_distance_matrix", "exec"), ns)
    fn = ns["validate_distance_matrix"]
    rng = np.random.default_rng(7)
    base = rng.random((47, 47)); base = (base + base.T) / 2; np.fill_diagonal(base, 0.0)
    cases = {"ok": base.copy()}
    a = base.copy(); a[3, 5] += 1e-3; cases["asym"] = a
    a = base.copy(); a[3, 5] += 1e-9; cases["asym_below_tol"] = a
    a = base.copy(); a[7, 9] = a[9, 7] = -0.25; cases["negative"] = a
    a = base.copy(); a[7, 9] = a[9, 7] = -1e-12; cases["negative_below_tol"] = a
    a = base.copy(); a[4, 4] = 0.5; cases["diag"] = a
    a = base.copy(); a[1, 2] = a[2, 1] = np.nan; cases["nan"] = a
    a = base.copy(); a[1, 2] = a[2, 1] = np.inf; cases["inf"] = a
    a = base.copy(); a[0, 1] = 2.0; a[5, 5] = 1.0; a[8, 9] = a[9, 8] = -1.0; cases["several"] = a
    cases["not_square"] = base[:, :40].copy()
    cases["not_2d"] = base.reshape(47, 47, 1).copy()
    out, answers = {}, {}
    with np.errstate(all="ignore"):
        for k, mat in cases.items():
            ok, issues = fn(mat, k)
            out[f"vdm_{k}"] = mat
            answers[k] = {"valid": bool(ok), "issues": list(issues)}
    out["vdm_answers"] = np.array(json.dumps(answers, ensure_ascii=False))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v["valid"] for k, v in answers.items()})


if __name__ == "__main__":
    main()
