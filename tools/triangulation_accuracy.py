"""How accurate is the triangulation kernel's algorithm (csrc/triangulate.hip: Householder QR of the 4x4 DLT matrix, then
inverse iteration on R^T R through the two triangular solves) -- restated here in numpy, operation for operation except
the hardware reciprocal estimates -- against LAPACK (what the reference calls, utils/utills.py:236) and against a
40-digit SVD (mpmath), on bench-shaped stereo correspondences: true matches with 0.3 px noise plus random mismatches
that pass the stereo-row test.  CPU only.

    python tools/triangulation_accuracy.py
"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from slam_amd.synth import SynthConfig, SynthSequence, K_KITTI, M1_KITTI, M2_KITTI
cfg = SynthConfig(n_frames=60, n_features=2000, seed=1234)
seq = SynthSequence(cfg)
L, R = [], []
rng = np.random.default_rng(0)
for i in range(0, 60, 3):
    f = seq.frame(i)
    lm0, lm1 = f['lm'][0], f['lm'][1]
    idx1 = {int(l): k for k, l in enumerate(lm1) if l >= 0}
    for k, l in enumerate(lm0):
        if l >= 0 and int(l) in idx1:
            L.append(f['xy'][0][k]); R.append(f['xy'][1][idx1[int(l)]])
    # mismatches that pass the row test: random left/right with |dy| <= 2
    n0, n1 = f['count']
    a = rng.integers(0, n0, 400); b = rng.integers(0, n1, 400)
    ok = np.abs(f['xy'][0][a, 1] - f['xy'][1][b, 1]) <= 2
    L += list(f['xy'][0][a[ok]]); R += list(f['xy'][1][b[ok]])
L = np.array(L, np.float64); R = np.array(R, np.float64)
print('points', len(L))
P = K_KITTI @ M1_KITTI; Q = K_KITTI @ M2_KITTI
A = np.stack([L[:, 0:1] * P[2] - P[0], L[:, 1:2] * P[2] - P[1], R[:, 0:1] * Q[2] - Q[0], R[:, 1:2] * Q[2] - Q[1]], axis=1)
U, S, Vh = np.linalg.svd(A)
v_ref = Vh[:, -1, :]
X_ref = v_ref[:, :3] / v_ref[:, 3:]
print('sigma ratios s4/s3: median %.3g  90%% %.3g  99%% %.3g  max %.3g' % tuple(np.quantile(S[:, 3] / S[:, 2], [0.5, 0.9, 0.99, 1.0])))
print('s3 quantiles', np.quantile(S[:, 2], [0, 0.01, 0.5, 1.0]), 's1', np.quantile(S[:, 0], [0, 0.5, 1.0]))

def householder_R(A):
    R = A.copy()
    n = R.shape[0]
    for k in range(3):
        x = R[:, k:, k]
        alpha = np.sqrt((x * x).sum(1))
        s = np.where(x[:, 0] >= 0, 1.0, -1.0)
        v = x.copy()
        v[:, 0] += s * alpha
        vn2 = (v * v).sum(1)
        vn2 = np.where(vn2 == 0, 1.0, vn2)
        # apply H = I - 2 v v^T / vn2 to R[:, k:, k:]
        sub = R[:, k:, k:]
        w = np.einsum('ni,nij->nj', v, sub)
        sub -= (2.0 / vn2)[:, None, None] * v[:, :, None] * w[:, None, :]
        R[:, k:, k:] = sub
    return np.triu(R)

Rm = householder_R(A)
def solve_lower_T(R, x):  # R^T w = x
    w = np.zeros_like(x)
    for i in range(4):
        acc = x[:, i].copy()
        for j in range(i):
            acc -= R[:, j, i] * w[:, j]
        w[:, i] = acc / R[:, i, i]
    return w
def solve_upper(R, w):
    z = np.zeros_like(w)
    for i in range(3, -1, -1):
        acc = w[:, i].copy()
        for j in range(i + 1, 4):
            acc -= R[:, i, j] * z[:, j]
        z[:, i] = acc / R[:, i, i]
    return z
nrm = np.abs(Rm).max(axis=(1, 2))
for i in range(4):
    d = Rm[:, i, i]
    tiny = np.abs(d) < 1e-300
    Rm[:, i, i] = np.where(tiny, 1e-300, d)
x = np.zeros((len(A), 4)); x[:, 3] = 1
iters = np.zeros(len(A), int); done = np.zeros(len(A), bool)
for it in range(40):
    z = solve_upper(Rm, solve_lower_T(Rm, x))
    z /= np.sqrt((z * z).sum(1))[:, None]
    sgn = np.where((z * x).sum(1) < 0, -1.0, 1.0)
    diff = np.abs(z - sgn[:, None] * x).max(1)
    x = np.where(done[:, None], x, z)
    newly = (~done) & (diff <= 1e-12)
    iters[~done] += 1
    done |= newly
    if done.all(): break
print('iterations: mean %.2f  hist' % iters.mean(), np.bincount(iters)[:15], 'not converged', (~done).sum())
Xi = x[:, :3] / x[:, 3:]
rel = np.linalg.norm(Xi - X_ref, axis=1) / np.linalg.norm(X_ref, axis=1)
print('inverse iteration vs LAPACK: median %.3g  99%% %.3g  max %.3g' % tuple(np.quantile(rel, [0.5, 0.99, 1.0])))
import mpmath
mpmath.mp.dps = 40
worst = np.argsort(-rel)[:60]
some = np.concatenate([worst, rng.integers(0, len(A), 60)])
err_ii, err_la = [], []
for i in some:
    M = mpmath.matrix(A[i].tolist())
    U_, S_, V_ = mpmath.svd_r(M)
    k = min(range(4), key=lambda j: S_[j])
    v = np.array([float(V_[k, j] / V_[k, 3]) for j in range(3)])
    err_ii.append(np.linalg.norm(Xi[i] - v) / np.linalg.norm(v))
    err_la.append(np.linalg.norm(X_ref[i] - v) / np.linalg.norm(v))
err_ii, err_la = np.array(err_ii), np.array(err_la)
print('worst-60: inv-iter vs exact max %.3g median %.3g | LAPACK vs exact max %.3g median %.3g' % (err_ii[:60].max(), np.median(err_ii[:60]), err_la[:60].max(), np.median(err_la[:60])))
print('random-60: inv-iter vs exact max %.3g median %.3g | LAPACK vs exact max %.3g median %.3g' % (err_ii[60:].max(), np.median(err_ii[60:]), err_la[60:].max(), np.median(err_la[60:])))
