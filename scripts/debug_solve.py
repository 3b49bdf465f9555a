import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd.engine import Engine
e = Engine()
np.set_printoptions(precision=5, linewidth=200, suppress=True)
for n, nblk in [(1, 1), (2, 1), (4, 1), (4, 2), (14, 2)]:
    rng = np.random.default_rng(n * 10 + nblk)
    Hd = np.zeros((1, nblk, n, n)); Ho = np.zeros((1, max(nblk - 1, 0), n, n))
    for i in range(nblk):
        A = rng.normal(size=(3 * n, 2 * n))
        Hd[0, i] += A[:, :n].T @ A[:, :n] + 1e-3 * np.eye(n)
        if i + 1 < nblk:
            Hd[0, i + 1] += A[:, n:].T @ A[:, n:]
            Ho[0, i] = A[:, n:].T @ A[:, :n]
    rhs = rng.normal(size=(1, nblk, n))
    x, ok = e.block_tridiag_solve(Hd, Ho, rhs)
    H = np.zeros((nblk * n, nblk * n))
    for i in range(nblk):
        H[i*n:(i+1)*n, i*n:(i+1)*n] = Hd[0, i]
        if i + 1 < nblk:
            H[(i+1)*n:(i+2)*n, i*n:(i+1)*n] = Ho[0, i]; H[i*n:(i+1)*n, (i+1)*n:(i+2)*n] = Ho[0, i].T
    xd = np.linalg.solve(H, rhs.reshape(-1))
    print(n, nblk, 'ok', ok, 'maxerr', np.abs(x.reshape(-1) - xd).max())
    if n <= 4: print(' x ', x.reshape(-1), '\n xd', xd)
