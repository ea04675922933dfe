"""Independent numpy statement of the matching semantics (a third opinion beside oracle/ and HIP)."""
import numpy as np


def sq_dists(A, B):
    a = np.rint(np.asarray(A)).astype(np.int64)
    b = np.rint(np.asarray(B)).astype(np.int64)
    return (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * (a @ b.T)


def directed(A, B, ratio=0.8):
    """q -> t for rows passing Lowe's test; exact 2-NN, ties -> lower train index."""
    n1, n2 = len(A), len(B)
    if n1 == 0 or n2 < 2:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    D = sq_dists(A, B)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    d0 = np.sqrt(np.take_along_axis(D, order[:, :1], 1)[:, 0].astype(np.float32))
    d1 = np.sqrt(np.take_along_axis(D, order[:, 1:2], 1)[:, 0].astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        quot = (d0 / d1).astype(np.float32)
    ok = quot.astype(np.float64) < ratio
    q = np.nonzero(ok)[0].astype(np.uint32)
    return q, order[ok, 0].astype(np.uint32)


def mutual(A, B, ratio=0.8, min_dir=30, min_mutual=30):
    q12, t12 = directed(A, B, ratio)
    q21, t21 = directed(B, A, ratio)
    back = dict(zip(q21.tolist(), t21.tolist()))
    keep = [(q, t) for q, t in zip(q12.tolist(), t12.tolist()) if back.get(t, -1) == q]
    edge = len(q12) >= min_dir and len(q21) >= min_dir and len(keep) > min_mutual
    stats = np.array([len(q12), len(q21), len(keep), int(edge)], dtype=np.int32)
    if not edge:
        keep = []
    q = np.array([k[0] for k in keep], dtype=np.uint32)
    t = np.array([k[1] for k in keep], dtype=np.uint32)
    return q, t, stats
