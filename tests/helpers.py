import numpy as np


def rot_trans_err(A, B):
    A = np.asarray(A, np.float64).reshape(3, 4); B = np.asarray(B, np.float64).reshape(3, 4)
    D = A[:, :3].T @ B[:, :3]
    w = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    ang = float(np.arctan2(np.linalg.norm(w), (np.trace(D) - 1.0) / 2.0))     # robust near 0, unlike arccos of the trace
    return ang, float(np.linalg.norm(A[:, 3] - B[:, 3]))


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def se3_exp(omega, v, dt):
    """Reference-independent SE(3) exponential of dt*[omega; v] via scipy expm."""
    from scipy.linalg import expm
    X = np.zeros((4, 4)); X[:3, :3] = hat(omega); X[:3, 3] = v
    return expm(dt * X)


def make_tf(axis, ang, t):
    axis = np.asarray(axis, np.float64); axis = axis / np.linalg.norm(axis)
    K = hat(axis)
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    return np.concatenate([R, np.asarray(t, np.float64)[:, None]], axis=1).astype(np.float32)
