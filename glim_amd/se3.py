"""Host-side SE(3) helpers (numpy): gtsam::Pose3 conventions, tangent [omega; v], T (+) xi = T @ exp(xi)."""
import numpy as np


def hat(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


def se3_exp(xi):
    xi = np.asarray(xi, dtype=np.float64).reshape(6)
    w, v = xi[:3], xi[3:]
    th2 = float(w @ w)
    th = np.sqrt(th2)
    W = hat(w)
    W2 = W @ W
    if th < 1e-8:
        a, b, c = 1.0 - th2 / 6.0, 0.5 - th2 / 24.0, 1.0 / 6.0 - th2 / 120.0
    else:
        a, b, c = np.sin(th) / th, (1.0 - np.cos(th)) / th2, (th - np.sin(th)) / (th2 * th)
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + a * W + b * W2
    T[:3, 3] = (np.eye(3) + b * W + c * W2) @ v
    return T


def solve_damped(H, b, lam=0.0):
    """Gauss-Newton / LM step: (H + lam I) x = -b."""
    return np.linalg.solve(np.asarray(H) + lam * np.eye(6), -np.asarray(b))
