import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg
g = np.load("tests/golden/leaf_kats.npz")
soils = g["soils"]; S = len(soils); n = g["geff"].shape[1]
bc = dict(zip(["alpha", "n", "ksat", "theta_e", "theta_r"], [np.repeat(soils[:, j], n) for j in range(5)]))
t1, t2 = g["geff_theta1"].ravel(), g["geff_theta2"].ravel()
te32 = bc["theta_e"].astype(np.float32)
t1c, t2c = np.minimum(t1.astype(np.float32), te32), np.minimum(t2.astype(np.float32), te32)
got = lg.leaf_batch("geff", t1c, t2c, dtype=torch.float32, **bc).cpu().numpy()
bad = np.nonzero(np.isnan(got))[0]
print("nan count", len(bad))
for b in bad[:12]:
    s = b // n
    tr32, te = np.float32(bc["theta_r"][b]), te32[b]
    se1 = (t1c[b] - tr32) / (te - tr32); se2 = (t2c[b] - tr32) / (te - tr32)
    print(b, "soil", s, soils[s], "t1", t1c[b], "t2", t2c[b], "te", te, "se1", se1, "se2", se2, "ref", g["geff"].ravel()[b])
for op, x in (("h_from_se", np.float32([1.0, 0.9999999, 0.999999, 0.99999, 0.5])), ("k_from_se", np.float32([1.0, 0.9999999, 0.999999, 0.99999, 0.5]))):
    k = len(x); s0 = soils[0]
    r = lg.leaf_batch(op, x, dtype=torch.float32, alpha=[s0[0]] * k, n=[s0[1]] * k, ksat=[s0[2]] * k, theta_e=[s0[3]] * k, theta_r=[s0[4]] * k)
    print(op, r.cpu().numpy())
