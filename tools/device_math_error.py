"""Accuracy of the lean fp64 log2 / exp2 / pow on the real hardware (dev tool): lgar_leaf_batch ops 7-9 against numpy
long double (x87 extended: 64-bit mantissa).  Prints the worst errors; tests/test_gpu_parity.py asserts the bounds."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg

rng = np.random.default_rng(0)
n = 1 << 20
one = np.ones(n)
kw = dict(alpha=one, n=one * 2, ksat=one, theta_e=one, theta_r=one * 0)
res = {}
x = np.concatenate([np.exp(rng.uniform(np.log(1e-12), np.log(1e12), n // 2)), 1.0 + rng.uniform(-0.3, 0.4, n // 2)])
got = lg.leaf_batch("log2", x, **kw).cpu().numpy().astype(np.longdouble)
ref = np.log2(x.astype(np.longdouble))
res["log2_abs_over_max1"] = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
near = np.abs(x - 1.0) < 0.4
res["log2_rel_near_1"] = float(np.max(np.abs(got[near] - ref[near]) / np.maximum(np.abs(ref[near]), 1e-300)))
y = rng.uniform(-1000.0, 1000.0, n)
got = lg.leaf_batch("exp2", y, **kw).cpu().numpy().astype(np.longdouble)
ref = np.exp2(y.astype(np.longdouble))
res["exp2_rel"] = float(np.max(np.abs(got - ref) / ref))
xb = np.exp(rng.uniform(np.log(1e-8), np.log(1e8), n))
yb = rng.uniform(-6.0, 6.0, n)
got = lg.leaf_batch("pow", xb, yb, **kw).cpu().numpy().astype(np.longdouble)
ref = np.power(xb.astype(np.longdouble), yb.astype(np.longdouble))
res["pow_rel"] = float(np.max(np.abs(got - ref) / ref))
print(json.dumps(res))
