"""Independent float64 statement of the Fast Global Smoother (test infrastructure).

Not a restatement of the reference's code order: it builds the tridiagonal
systems (I + lambda_n * L_w) explicitly and solves them with LAPACK
(scipy.linalg.solve_banded) in float64.  Used to bound the float32 oracle's
error from a second, unrelated implementation.

Maths (FGS.cpp:50-59, :439-464, :674; EF.hpp:388-391):
  w_j   = exp(-sqrt(sum_c (g_j - g_{j+1})^2) / sigma)      weight between j and j+1
  a_j   = -lambda * w_{j-1},  c_j = -lambda * w_j,  b_j = 1 - a_j - c_j
  iteration n uses lambda_n = lambda * attenuation**n, H pass then V pass.
"""
import numpy as np
from scipy.linalg import solve_banded


def edge_weights_f64(guide, sigma):
    g = np.asarray(guide).astype(np.float64)
    if g.ndim == 2:
        g = g[:, :, None]
    dh = np.sqrt(((g[:, :-1] - g[:, 1:]) ** 2).sum(axis=2))
    dv = np.sqrt(((g[:-1, :] - g[1:, :]) ** 2).sum(axis=2))
    wh = np.zeros(g.shape[:2])
    wv = np.zeros(g.shape[:2])
    wh[:, :-1] = np.exp(-dh / sigma)
    wv[:-1, :] = np.exp(-dv / sigma)
    return wh, wv  # positive weights; last column / row zero


def _solve_lines(w_lines, f_lines, lam):
    """Solve (I + lam*L_w) x = f for each line (axis 1 is the scanline)."""
    out = np.empty_like(f_lines)
    n = w_lines.shape[1]
    ab = np.zeros((3, n))
    for i in range(w_lines.shape[0]):
        w = w_lines[i]
        c = -lam * w            # super-diagonal, c[n-1] == 0
        a = np.zeros(n)
        a[1:] = -lam * w[:-1]   # sub-diagonal
        ab[0, 1:] = c[:-1]
        ab[1] = 1.0 - a - c
        ab[2, :-1] = a[1:]
        out[i] = solve_banded((1, 1), ab, f_lines[i])
    return out


def fgs_f64(guide, src, lam, sigma, atten=0.25, num_iter=3):
    """float64 FGS of a single-channel image `src` (h, w)."""
    wh, wv = edge_weights_f64(guide, sigma)
    u = np.asarray(src, np.float64).copy()
    # the reference multiplies lambda by a float32 attenuation in float32 (FGS.cpp:146-147,211)
    lam_n = np.float32(lam)
    att = np.float32(atten)
    for _ in range(num_iter):
        u = _solve_lines(wh, u, float(lam_n))
        u = _solve_lines(wv.T.copy(), u.T.copy(), float(lam_n)).T.copy()
        lam_n = np.float32(lam_n * att)
    return u
