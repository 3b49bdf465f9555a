import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd.engine import Engine
e = Engine()
n, nblk, B = 14, 101, 64
rng = np.random.default_rng(0)
Hd = np.zeros((B, nblk, n, n)); Ho = np.zeros((B, nblk - 1, n, n))
for b in range(B):
    A = rng.normal(size=(nblk, 3 * n, 2 * n))
    for i in range(nblk):
        Hd[b, i] += A[i][:, :n].T @ A[i][:, :n] + 1e-3 * np.eye(n)
        if i + 1 < nblk:
            Hd[b, i + 1] += A[i][:, n:].T @ A[i][:, n:]
            Ho[b, i] = A[i][:, n:].T @ A[i][:, :n]
rhs = rng.normal(size=(B, nblk, n))
for _ in range(5):
    x, ok = e.block_tridiag_solve(Hd, Ho, rhs)
print('ok', ok.sum())
