"""Trajectory I/O and accuracy metrics for the result files of the reference system (KITTI 3x4 row format written by
VSlamSystem::saveTrajectoryAndPosition, src/System.cpp:87-124).

  read_kitti(path)            -> (n, 4, 4) camera-to-world poses
  ate_rmse(est, gt)           -> absolute trajectory error: RMSE of the positions after the least-squares rigid
                                 alignment (rotation + translation, no scale) of est onto gt (Horn / Umeyama)
  rpe(est, gt, delta=1)       -> relative pose error over `delta` frames: (translation RMSE, rotation RMSE in rad)
"""
import numpy as np


def read_kitti(path):
    rows = np.loadtxt(path, ndmin=2)
    if rows.shape[1] != 12:
        raise ValueError("expected 12 numbers per line, got %d" % rows.shape[1])
    T = np.tile(np.eye(4), (len(rows), 1, 1))
    T[:, :3, :] = rows.reshape(-1, 3, 4)
    return T


def align_rigid(src, dst):
    """Least-squares R, t with dst ~ R src + t (points as rows)."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    H = (src - mu_s).T @ (dst - mu_d)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mu_d - R @ mu_s


def ate_rmse(est, gt):
    est, gt = np.asarray(est, float), np.asarray(gt, float)
    if est.shape != gt.shape or len(est) < 3:
        raise ValueError("need two trajectories of the same length (>= 3 poses)")
    R, t = align_rigid(est[:, :3, 3], gt[:, :3, 3])
    d = (R @ est[:, :3, 3].T).T + t - gt[:, :3, 3]
    return float(np.sqrt((d ** 2).sum(1).mean()))


def rpe(est, gt, delta=1):
    est, gt = np.asarray(est, float), np.asarray(gt, float)
    if est.shape != gt.shape or len(est) <= delta:
        raise ValueError("need two trajectories of the same length (> delta poses)")
    te, re_ = [], []
    for i in range(len(est) - delta):
        dE = np.linalg.inv(est[i]) @ est[i + delta]
        dG = np.linalg.inv(gt[i]) @ gt[i + delta]
        E = np.linalg.inv(dG) @ dE
        te.append(np.linalg.norm(E[:3, 3]))
        re_.append(np.arccos(np.clip((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)))
    return float(np.sqrt(np.mean(np.square(te)))), float(np.sqrt(np.mean(np.square(re_))))
