"""Seeded synthetic inputs for the tracking / local-BA hot path.

No dataset is available (EuRoC / KITTI are not in the container and there is no
network), so every test and the benchmark run on procedural scenes rendered to
the rectified stereo rig of the config being measured.
"""
import numpy as np

# rigs from the reference configs (config/config_MH_01.yaml:33-36,95-98,
# config/config_kitti_00.yaml:20-23,42-45); C5 from SURVEY.md §8d
RIGS = {
    "euroc": dict(w=752, h=480, fx=435.2046959714599, fy=435.2046959714599, cx=367.4517211914062,
                  cy=252.2008514404297, bl=0.110074137800478, fps=20.0),
    "kitti": dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bl=0.53716, fps=10.0),
    "synthetic": dict(w=1920, h=1200, fx=1100.0, fy=1100.0, cx=960.0, cy=600.0, bl=0.12, fps=20.0),
}


def make_texture(seed=0xC0FFEE, size=2048, nrect=4000):
    """Multi-octave value noise plus random high-contrast rectangles, u8 [size,size]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tex = np.zeros((size, size), np.float32)
    amp, tot = 1.0, 0.0
    for octave in range(3, 10):
        n = 2 ** octave
        g = rng.random((n + 1, n + 1), dtype=np.float32)
        # bilinear upsample of the coarse grid
        xs = np.linspace(0, n, size, endpoint=False, dtype=np.float32)
        i0 = np.floor(xs).astype(np.int32)
        f = xs - i0
        rows = g[i0, :] * (1 - f)[:, None] + g[i0 + 1, :] * f[:, None]
        up = rows[:, i0] * (1 - f)[None, :] + rows[:, i0 + 1] * f[None, :]
        tex += amp * up
        tot += amp
        amp *= 0.6
    tex = tex / tot
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    img = (40 + 140 * tex).astype(np.float32)
    for _ in range(nrect):
        w, h = rng.integers(6, 60, 2)
        x, y = rng.integers(0, size - 60, 2)
        v = float(rng.integers(0, 256))
        img[y:y + h, x:x + w] = 0.35 * img[y:y + h, x:x + w] + 0.65 * v
    return np.clip(img, 0, 255).astype(np.uint8)


_TEX_CACHE = {}


def texture(seed=0xC0FFEE, size=2048):
    key = (seed, size)
    if key not in _TEX_CACHE:
        _TEX_CACHE[key] = make_texture(seed, size)
    return _TEX_CACHE[key]


def make_scene(seed=7, nboxes=14):
    """Planes: list of (origin[3], ex[3], ey[3], half_w, half_h, tex_off[2], tex_scale px/m)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    planes = []
    # back wall, floor-ish and two side walls so every ray hits something
    planes.append((np.array([0, 0, 9.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), 30, 30, (100, 100), 120.0))
    planes.append((np.array([0, 1.6, 5.0]), np.array([1.0, 0, 0]), np.array([0, 0.0, 1.0]), 30, 30, (900, 300), 150.0))
    planes.append((np.array([-6.0, 0, 5.0]), np.array([0, 0, 1.0]), np.array([0, 1.0, 0]), 30, 30, (300, 1200), 140.0))
    planes.append((np.array([6.0, 0, 5.0]), np.array([0, 0, 1.0]), np.array([0, 1.0, 0]), 30, 30, (1300, 800), 140.0))
    for _ in range(nboxes):
        z = rng.uniform(1.8, 7.0)
        c = np.array([rng.uniform(-0.7, 0.7) * z, rng.uniform(-0.45, 0.35) * z, z])
        yaw = rng.uniform(-0.5, 0.5)
        ex = np.array([np.cos(yaw), 0, np.sin(yaw)])
        ey = np.array([0, 1.0, 0])
        planes.append((c, ex, ey, rng.uniform(0.25, 0.8), rng.uniform(0.25, 0.8),
                       (int(rng.integers(0, 1500)), int(rng.integers(0, 1500))), rng.uniform(250, 450)))
    return planes


def render(planes, tex, rig, T_wc, noise_seed=None, noise_sigma=2.0):
    """Render one u8 view.  T_wc: 4x4 world<-camera.  Returns (image, depth)."""
    w, h = rig["w"], rig["h"]
    fx, fy, cx, cy = rig["fx"], rig["fy"], rig["cx"], rig["cy"]
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)  # camera rays, z=1
    R, t = T_wc[:3, :3], T_wc[:3, 3]
    d_w = d_c @ R.T
    best = np.full((h, w), np.inf)
    out = np.zeros((h, w), np.float32)
    ts = tex.shape[0]
    for (o, ex, ey, hw, hh, toff, tsc) in planes:
        n = np.cross(ex, ey)
        denom = d_w @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            lam = ((o - t) @ n) / denom  # depth along camera z (since ray has z_c = 1)
        p = t[None, None, :] + lam[..., None] * d_w
        a = (p - o) @ ex
        b = (p - o) @ ey
        hit = (lam > 0.05) & (np.abs(a) <= hw) & (np.abs(b) <= hh) & (lam < best) & np.isfinite(lam)
        if not hit.any():
            continue
        tu = (toff[0] + (a[hit] + hw) * tsc) % (ts - 1)
        tv = (toff[1] + (b[hit] + hh) * tsc) % (ts - 1)
        x0 = np.floor(tu).astype(np.int64)
        y0 = np.floor(tv).astype(np.int64)
        fxr, fyr = (tu - x0).astype(np.float32), (tv - y0).astype(np.float32)
        x1, y1 = np.minimum(x0 + 1, ts - 1), np.minimum(y0 + 1, ts - 1)
        val = (tex[y0, x0] * (1 - fxr) * (1 - fyr) + tex[y0, x1] * fxr * (1 - fyr)
               + tex[y1, x0] * (1 - fxr) * fyr + tex[y1, x1] * fxr * fyr)
        out[hit] = val
        best[hit] = lam[hit]
    if noise_seed is not None and noise_sigma > 0:
        rng = np.random.Generator(np.random.PCG64(noise_seed))
        out = out + rng.normal(0, noise_sigma, out.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8), best


def pose_at(i, fps=20.0):
    """Smooth 6-DoF trajectory sample (world<-left camera), ~0.5 m/s, small rotations."""
    s = i / fps
    yaw = 0.08 * np.sin(0.7 * s)
    pitch = 0.04 * np.sin(0.45 * s + 0.3)
    roll = 0.03 * np.sin(0.6 * s + 1.0)
    cy_, sy_ = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rx @ Rz
    T[:3, 3] = [0.35 * np.sin(0.5 * s), 0.05 * np.sin(0.9 * s), 0.25 * s * 0.5 + 0.15 * np.sin(0.4 * s)]
    return T


_FRAME_CACHE = {}


def stereo_frame(i, rig_name="euroc", scene_seed=7, tex_seed=0xC0FFEE, noise=True):
    """Left/right u8 images of frame i plus the ground-truth pose (rendered once per process: a 1920x1200 pair takes
    seconds on the host)."""
    key = (i, rig_name, scene_seed, tex_seed, noise)
    if key not in _FRAME_CACHE:
        if len(_FRAME_CACHE) > 96:
            _FRAME_CACHE.clear()
        _FRAME_CACHE[key] = _render_stereo_frame(i, rig_name, scene_seed, tex_seed, noise)
    L, R, T = _FRAME_CACHE[key]
    return L.copy(), R.copy(), T.copy()


def prerender(frames, rig_name="euroc", workers=0):
    """Render several frames into the cache on `workers` threads (numpy releases the GIL in the heavy parts)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    todo = [i for i in frames if (i, rig_name, 7, 0xC0FFEE, True) not in _FRAME_CACHE]
    if not todo:
        return
    texture(0xC0FFEE)                       # (built once, before the threads race for it)
    n = workers if workers > 0 else min(len(todo), max(1, (os.cpu_count() or 2) - 1), 16)
    with ThreadPoolExecutor(n) as pool:
        res = list(pool.map(lambda i: _render_stereo_frame(i, rig_name, 7, 0xC0FFEE, True), todo))
    for i, r in zip(todo, res):
        _FRAME_CACHE[(i, rig_name, 7, 0xC0FFEE, True)] = r


def _render_stereo_frame(i, rig_name, scene_seed, tex_seed, noise):
    rig = RIGS[rig_name]
    planes = make_scene(scene_seed)
    tex = texture(tex_seed)
    T = pose_at(i, rig["fps"])
    ext = np.eye(4)
    ext[0, 3] = rig["bl"]  # right camera = pose * extrinsics (reference src/Camera.cpp:57)
    left, _ = render(planes, tex, rig, T, noise_seed=(1000 + 2 * i) if noise else None)
    right, _ = render(planes, tex, rig, T @ ext, noise_seed=(1001 + 2 * i) if noise else None)
    return left, right, T


def random_image(w, h, seed):
    """Cheap textured test image (crop of the noise texture + gaussian noise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tex = texture()
    x, y = int(rng.integers(0, tex.shape[1] - w)) if w < tex.shape[1] else 0, \
        int(rng.integers(0, tex.shape[0] - h)) if h < tex.shape[0] else 0
    if w <= tex.shape[1] and h <= tex.shape[0]:
        img = tex[y:y + h, x:x + w].astype(np.float32)
    else:
        reps = (h // tex.shape[0] + 1, w // tex.shape[1] + 1)
        img = np.tile(tex, reps)[:h, :w].astype(np.float32)
    img = img + rng.normal(0, 2.0, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _small_pose(rng, rot_sigma, trans_sigma):
    """Random small rigid perturbation exp([w, v])."""
    w = rng.normal(0, rot_sigma, 3)
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + W if th < 1e-12 else np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.normal(0, trans_sigma, 3)
    return T


def make_ba_problem(rig_name="euroc", n_local=10, n_fixed=4, n_lm=3000, seed=0xBA5E, max_views=12,
                    pix_noise=0.5, pose_noise=(0.009, 0.02), point_noise=0.03, outlier_frac=0.02, circle=False):
    """Flattened local-BA problem: keyframes on a trajectory (or on a circle looking
    inward for the 64-KF global case), landmarks seen by every keyframe whose frustum holds them
    (cap max_views), left + `close` right observations, pixel noise scaled by the octave."""
    rig = RIGS[rig_name]
    rng = np.random.Generator(np.random.PCG64(seed))
    K = n_local + n_fixed
    poses = []
    for k in range(K):
        if circle:
            a = 2 * np.pi * k / K
            c = np.array([5 * np.sin(a), 0.0, -5 * np.cos(a)])
            z = -c / np.linalg.norm(c)
            x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
            y = np.cross(z, x)
            T = np.eye(4); T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, c
        else:
            T = pose_at(4 * k, rig["fps"])
        poses.append(T)
    poses = np.array(poses)
    if circle:
        lm = rng.uniform(-2, 2, (n_lm, 3))
    else:
        lm = np.stack([rng.uniform(-4, 4, n_lm), rng.uniform(-2.5, 2, n_lm), rng.uniform(2.0, 12, n_lm)], 1)
    scale = 1.2 ** np.arange(8)
    pk, pl, pf, puv, poct = [], [], [], [], []
    for l in range(n_lm):
        views = 0
        order = rng.permutation(K)
        for k in order:
            Tcw = np.linalg.inv(poses[k])
            q = Tcw[:3, :3] @ lm[l] + Tcw[:3, 3]
            if q[2] <= 0.3:
                continue
            u = rig["fx"] * q[0] / q[2] + rig["cx"]; v = rig["fy"] * q[1] / q[2] + rig["cy"]
            uR = rig["fx"] * (q[0] - rig["bl"]) / q[2] + rig["cx"]
            if not (20 <= u < rig["w"] - 20 and 20 <= v < rig["h"] - 20 and 20 <= uR):
                continue
            oct_ = int(np.clip(np.round(np.log(max(q[2], 1e-3) / 2.0) / np.log(1.2)), 0, 7))
            s = pix_noise * scale[oct_]
            close = q[2] < 40 * rig["bl"]
            kind = rng.random()
            if kind < 0.07:         # right-only observation
                flags = 2
            elif close:
                flags = 3
            else:
                flags = 1
            nz = rng.normal(0, s, 4)
            if rng.random() < outlier_frac:
                nz += rng.normal(0, 25, 4)
            pk.append(k); pl.append(l); pf.append(flags)
            puv.append([u + nz[0], v + nz[1], uR + nz[2], v + nz[3]]); poct.append([oct_, oct_])
            views += 1
            if views >= max_views:
                break
    kf_fixed = np.zeros(K, np.uint8); kf_fixed[n_local:] = 1
    kf_local = np.ones(K, np.uint8); kf_local[n_local:] = 0
    init_poses = poses.copy()
    for k in range(n_local):
        init_poses[k] = poses[k] @ _small_pose(rng, pose_noise[0], pose_noise[1])
    init_lm = lm + rng.normal(0, point_noise, lm.shape)
    # keyframe ids: local ones are the newest
    kf_id = np.concatenate([np.arange(n_fixed, K), np.arange(0, n_fixed)]).astype(np.int64)
    return dict(rig=rig, kf_pose=init_poses, kf_pose_true=poses, kf_id=kf_id, kf_fixed=kf_fixed, kf_local=kf_local,
                lm=init_lm, lm_true=lm, pair_kf=np.array(pk, np.int32), pair_lm=np.array(pl, np.int32),
                pair_flags=np.array(pf, np.uint8), pair_uv=np.array(puv, np.float32), pair_oct=np.array(poct, np.int32))


def make_ba_problem_c5(n_lm=100000, n_local=62, n_fixed=2, seed=0xBA5E, max_views=12, rig_name="synthetic",
                       pix_noise=0.5, pose_noise=(0.009, 0.02), point_noise=0.03, outlier_frac=0.02):
    """The C5 global-BA problem of SURVEY section 8d at full size (64 keyframes on a circle of radius 5 m looking inward,
    100 000 landmarks uniform in a 4 m cube, <= 12 views each, pixel noise 0.5 px x octave scale, pose noise 2 cm /
    0.5 deg, point noise 3 cm): the same construction as make_ba_problem(circle=True), vectorised (own random
    stream) so that 10^5 landmarks take seconds, not minutes."""
    rig = RIGS[rig_name]
    rng = np.random.Generator(np.random.PCG64(seed))
    K = n_local + n_fixed
    poses = np.zeros((K, 4, 4))
    for k in range(K):
        a = 2 * np.pi * k / K
        c = np.array([5 * np.sin(a), 0.0, -5 * np.cos(a)])
        z = -c / np.linalg.norm(c)
        x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        T = np.eye(4); T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, c
        poses[k] = T
    lm = rng.uniform(-2, 2, (n_lm, 3))
    Tcw = np.linalg.inv(poses)
    pk, pl, pf, puv, poct = [], [], [], [], []
    scale = 1.2 ** np.arange(8)
    step = 20000
    for l0 in range(0, n_lm, step):
        P = lm[l0:l0 + step]
        q = np.einsum("kij,nj->nki", Tcw[:, :3, :3], P) + Tcw[None, :, :3, 3]          # (n, K, 3)
        zc = q[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = rig["fx"] * q[..., 0] / zc + rig["cx"]; v = rig["fy"] * q[..., 1] / zc + rig["cy"]
            uR = rig["fx"] * (q[..., 0] - rig["bl"]) / zc + rig["cx"]
        vis = (zc > 0.3) & (u >= 20) & (u < rig["w"] - 20) & (v >= 20) & (v < rig["h"] - 20) & (uR >= 20)
        pri = rng.random(vis.shape)
        pri[~vis] = 2.0
        order = np.argsort(pri, axis=1)[:, :max_views]                                 # random subset of the visible views
        take = np.take_along_axis(vis, order, 1)
        li, ci = np.nonzero(take)
        kk = order[li, ci]
        zz = zc[li, kk]
        oct_ = np.clip(np.round(np.log(np.maximum(zz, 1e-3) / 2.0) / np.log(1.2)), 0, 7).astype(np.int32)
        sg = pix_noise * scale[oct_]
        close = zz < 40 * rig["bl"]
        kind = rng.random(len(li))
        flags = np.where(kind < 0.07, 2, np.where(close, 3, 1)).astype(np.uint8)
        nz = rng.normal(0, 1, (len(li), 4)) * sg[:, None]
        outl = rng.random(len(li)) < outlier_frac
        nz[outl] += rng.normal(0, 25, (int(outl.sum()), 4))
        uv = np.stack([u[li, kk], v[li, kk], uR[li, kk], v[li, kk]], 1) + nz
        pk.append(kk.astype(np.int32)); pl.append((li + l0).astype(np.int32)); pf.append(flags)
        puv.append(uv.astype(np.float32)); poct.append(np.stack([oct_, oct_], 1))
    kf_fixed = np.zeros(K, np.uint8); kf_fixed[n_local:] = 1
    kf_local = np.ones(K, np.uint8); kf_local[n_local:] = 0
    init_poses = poses.copy()
    for k in range(n_local):
        init_poses[k] = poses[k] @ _small_pose(rng, pose_noise[0], pose_noise[1])
    init_lm = lm + rng.normal(0, point_noise, lm.shape)
    kf_id = np.concatenate([np.arange(n_fixed, K), np.arange(0, n_fixed)]).astype(np.int64)
    return dict(rig=rig, kf_pose=init_poses, kf_pose_true=poses, kf_id=kf_id, kf_fixed=kf_fixed, kf_local=kf_local,
                lm=init_lm, lm_true=lm, pair_kf=np.concatenate(pk), pair_lm=np.concatenate(pl),
                pair_flags=np.concatenate(pf), pair_uv=np.concatenate(puv), pair_oct=np.concatenate(poct).astype(np.int32))


# T_bc1 of the EuRoC configs (config/config_MH_01.yaml T_bc1.data): body_P_sensor of the IMU factor
T_BC1 = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                  [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                  [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                  [0.0, 0.0, 0.0, 1.0]])
IMU_NOISE = dict(gyro_density=1.6968e-04, gyro_walk=1.9393e-05, acc_density=2.0e-3, acc_walk=3.0e-3, hz=200)


def imu_samples(t0, t1, fps=20.0, hz=200, T_bs=T_BC1, gravity=(0.0, 9.81, 0.0), noise_seed=None, bias=None, pose_fn=None):
    """IMU samples (acc, gyro in the SENSOR frame) strictly between frame times t0 < t1 (in frame units), generated
    from the pose_at() spline by central differences; returns (samples [n,6], dts [n]) with the reference's dt rule
    (dt_i = t_{i+1} - t_i, the last sample reuses the previous dt; src/FeatureTracker.cpp:338-353)."""
    g = np.asarray(gravity, float)
    ts = np.arange(np.ceil(t0 * hz / fps + 1e-9), np.floor(t1 * hz / fps - 1e-9) + 1) / hz     # seconds
    h = 1e-4
    out = []
    Rbs, tbs = T_bs[:3, :3], T_bs[:3, 3]
    for s in ts:
        f = s * fps
        pf = pose_fn if pose_fn is not None else pose_at
        Tm, T0, Tp = pf(f - h * fps, fps), pf(f, fps), pf(f + h * fps, fps)
        R = T0[:3, :3]
        Rdot = (Tp[:3, :3] - Tm[:3, :3]) / (2 * h)
        Wm = R.T @ Rdot
        w_b = np.array([Wm[2, 1] - Wm[1, 2], Wm[0, 2] - Wm[2, 0], Wm[1, 0] - Wm[0, 1]]) * 0.5
        acc_n = (Tp[:3, 3] - 2 * T0[:3, 3] + Tm[:3, 3]) / (h * h)
        # sensor position = p_b + R t_bs : add the centripetal term the factor models (angular acceleration ignored)
        a_b = R.T @ (acc_n - g) + np.cross(w_b, np.cross(w_b, tbs))
        out.append(np.concatenate([Rbs.T @ a_b, Rbs.T @ w_b]))
    samples = np.array(out).reshape(-1, 6)
    if bias is not None:
        samples = samples + np.asarray(bias)[None, :]
    if noise_seed is not None:
        rng = np.random.Generator(np.random.PCG64(noise_seed))
        samples[:, :3] += rng.normal(0, IMU_NOISE["acc_density"] * np.sqrt(hz), samples[:, :3].shape)
        samples[:, 3:] += rng.normal(0, IMU_NOISE["gyro_density"] * np.sqrt(hz), samples[:, 3:].shape)
    n = len(ts)
    dts = np.full(n, 1.0 / hz)
    return samples, dts, g


def make_mono_points_problem(rig_name="euroc", n_kf=4, n_points=400, seed=0x4D4F, trans_sigma=0.01, rot_sigma=0.002,
                             pix_noise=0.3, outlier_frac=0.1):
    """Input of FeatureTracker::addMappointsMono after its matchByRadius passes: n_kf keyframes a few centimetres apart
    (mono initialisation), every keypoint of keyframe 0 with 1..n_kf views.  A share of the views are outliers, a few
    points sit behind a camera or below world z = 0.1 (the reference's accept test looks at the world z).
    Returns dict(rig, kf_pose (n_kf,4,4) camera-to-world, kf_id, n_views, view_kf, view_xy, view_oct, truth)."""
    rng = np.random.default_rng(seed)
    rig = RIGS[rig_name]
    poses = [np.eye(4)]
    for _ in range(n_kf - 1):
        poses.append(poses[-1] @ _small_pose(rng, rot_sigma, trans_sigma))
    poses = np.stack(poses)
    ids = np.arange(100, 100 + n_kf, dtype=np.int32)[::-1].copy()        # lastKF (index 0) is the newest
    truth = np.zeros((n_points, 3))
    n_views = np.zeros(n_points, np.int32)
    view_kf = np.zeros((n_points, n_kf), np.int32)
    view_xy = np.zeros((n_points, n_kf, 2), np.float32)
    view_oct = np.zeros((n_points, n_kf), np.int32)
    for i in range(n_points):
        z = rng.uniform(2.0, 12.0) if rng.random() > 0.04 else rng.uniform(-1.0, 0.09)
        u, v = rng.uniform(20, rig["w"] - 20), rng.uniform(20, rig["h"] - 20)
        pc = np.array([(u - rig["cx"]) / rig["fx"] * z, (v - rig["cy"]) / rig["fy"] * z, z, 1.0])
        pw = poses[0] @ pc
        truth[i] = pw[:3]
        nv = int(rng.integers(1, n_kf + 1))
        others = list(rng.permutation(np.arange(1, n_kf))[:nv - 1])
        ks = [0] + sorted(int(k) for k in others)
        n_views[i] = len(ks)
        for e, k in enumerate(ks):
            q = np.linalg.inv(poses[k]) @ pw
            zz = q[2] if abs(q[2]) > 1e-6 else 1e-6
            x = rig["fx"] * q[0] / zz + rig["cx"] + rng.normal(0, pix_noise)
            y = rig["fy"] * q[1] / zz + rig["cy"] + rng.normal(0, pix_noise)
            if rng.random() < outlier_frac and e > 0:
                x += rng.uniform(-40, 40); y += rng.uniform(-40, 40)
            view_kf[i, e] = k; view_xy[i, e] = (x, y); view_oct[i, e] = int(rng.integers(0, 8))
    return dict(rig=rig, kf_pose=poses, kf_id=ids, n_views=n_views, view_kf=view_kf, view_xy=view_xy, view_oct=view_oct,
                truth=truth)


def make_kf_update_problem(rig_name="euroc", n_left=1500, n_right=1400, n_lm=2500, seed=0x4B46, shift=0.02, pix_noise=0.7):
    """Input of KeyFrame::updatePose: a keyframe (numb 7) whose pose is re-derived from a corrected previous keyframe;
    its slots point at landmarks created by itself (kdx == 7: they move along), by older keyframes (re-projected and
    gated) or newer ones (untouched); some slots are empty, some landmarks outliers.  Keypoints: structured array with
    fields x, y, octave (the caller converts to its keypoint dtype)."""
    rng = np.random.default_rng(seed)
    rig = RIGS[rig_name]
    cur = _small_pose(rng, 0.2, 1.0)                                  # current pose (world <- camera)
    ref = _small_pose(rng, 0.05, 0.3)                                 # refPose: relative to the previous keyframe
    key_old = cur @ np.linalg.inv(ref)
    key_new = key_old @ _small_pose(rng, 0.004, shift)                # the corrected previous keyframe
    kdx = rng.choice([3, 5, 7, 7, 9], n_lm).astype(np.int64)
    outlier = (rng.random(n_lm) < 0.05).astype(np.uint8)
    z = rng.uniform(1.5, 25.0, n_lm)
    u = rng.uniform(10, rig["w"] - 10, n_lm); v = rng.uniform(10, rig["h"] - 10, n_lm)
    pc = np.stack([(u - rig["cx"]) / rig["fx"] * z, (v - rig["cy"]) / rig["fy"] * z, z, np.ones(n_lm)], 1)
    lm = (cur @ pc.T).T[:, :3]

    def side(n, right):
        slot = rng.permutation(n_lm)[:n].astype(np.int32) if n <= n_lm else rng.integers(0, n_lm, n).astype(np.int32)
        slot[rng.random(n) < 0.15] = -1                               # (a map point sits in at most one slot per side)
        x = np.zeros(n, np.float32); y = np.zeros(n, np.float32); octv = rng.integers(0, 8, n).astype(np.int32)
        for i in range(n):
            m = slot[i] if slot[i] >= 0 else 0
            q = pc[m].copy()
            if right:
                q[0] -= rig["bl"]
            x[i] = rig["fx"] * q[0] / q[2] + rig["cx"] + rng.normal(0, pix_noise)
            y[i] = rig["fy"] * q[1] / q[2] + rig["cy"] + rng.normal(0, pix_noise)
        return slot, x, y, octv

    sl, xl, yl, ol = side(n_left, False)
    sr, xr, yr, orr = side(n_right, True)
    return dict(rig=rig, numb=7, key_pose=key_new, ref_pose=ref, cur_pose_inv=np.linalg.inv(cur), slotL=sl, slotR=sr,
                kL=(xl, yl, ol), kR=(xr, yr, orr), lm=lm, kdx=kdx, outlier=outlier, cur=cur)


# ---- bench workload: a long corridor, rendered on the GPU ----------------------------------------------------------------
# The parity tests use the small scene above (numpy renderer, identical on every host).  The benchmark needs hundreds of
# DISTINCT poses (a session must not revisit the places its map already covers), which the small scene cannot hold and the
# numpy renderer cannot produce in a bench's set-up time: the same plane model, extended along z, rendered by the same
# arithmetic on torch tensors.  Bench data generation only - nothing of the product or of the parity tests goes through it.

def make_corridor(seed=11, length=48.0, boxes_per_m=2.2):
    """Planes of a corridor along +z: floor, ceiling, two side walls, end wall, textured boxes every ~0.45 m of depth."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L2 = length + 20.0
    planes = [(np.array([0, 0, L2]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), 30, 30, (100, 100), 14.0),
              (np.array([0, 1.6, L2 / 2]), np.array([1.0, 0, 0]), np.array([0, 0.0, 1.0]), 30, L2, (900, 300), 150.0),
              (np.array([0, -2.6, L2 / 2]), np.array([1.0, 0, 0]), np.array([0, 0.0, 1.0]), 30, L2, (500, 1500), 130.0),
              (np.array([-6.0, 0, L2 / 2]), np.array([0, 0, 1.0]), np.array([0, 1.0, 0]), L2, 30, (300, 1200), 140.0),
              (np.array([6.0, 0, L2 / 2]), np.array([0, 0, 1.0]), np.array([0, 1.0, 0]), L2, 30, (1300, 800), 140.0)]
    for _ in range(int(length * boxes_per_m)):
        z = rng.uniform(1.8, length)
        # (outside the tube the camera flies through: |x| <= 0.6 m + the boxes' half width, so no plane is ever crossed)
        c = np.array([rng.uniform(1.7, 4.8) * (1.0 if rng.random() < 0.5 else -1.0), rng.uniform(-1.6, 1.3), z])
        yaw = rng.uniform(-0.5, 0.5)
        ex = np.array([np.cos(yaw), 0, np.sin(yaw)])
        planes.append((c, ex, np.array([0, 1.0, 0]), rng.uniform(0.25, 0.8), rng.uniform(0.25, 0.8),
                       (int(rng.integers(0, 1500)), int(rng.integers(0, 1500))), rng.uniform(250, 450)))
    return planes


def corridor_pose(i, fps=20.0, speed=0.5):
    """world <- left camera at frame i: `speed` m/s along the corridor, yaw sweeping at up to ~20 deg/s, small pitch / roll and
    lateral motion (smooth 6-DoF, 0.5 m/s, <= 20 deg/s)."""
    s = i / fps
    yaw = 0.70 * np.sin(0.5 * s)
    pitch = 0.05 * np.sin(0.45 * s + 0.3)
    roll = 0.04 * np.sin(0.6 * s + 1.0)
    cy_, sy_ = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rx @ Rz
    T[:3, 3] = [0.6 * np.sin(0.35 * s), 0.08 * np.sin(0.9 * s), speed * s + 0.10 * np.sin(0.4 * s)]
    return T


class TorchPlanes:
    """the plane model as tensors on `device` (built once per sequence)"""

    def __init__(self, planes, device, dtype):
        import torch
        def t(v):
            return torch.as_tensor(np.array(v, np.float64), dtype=dtype, device=device)
        self.np_planes = planes
        self.o = t([p[0] for p in planes]); self.ex = t([p[1] for p in planes]); self.ey = t([p[2] for p in planes])
        self.n = t([np.cross(p[1], p[2]) for p in planes])
        self.hw = t([p[3] for p in planes]); self.hh = t([p[4] for p in planes])
        self.toff = t([p[5] for p in planes]); self.tsc = t([p[6] for p in planes])
        self.z = np.array([p[0][2] for p in planes]); self.small = np.array([p[3] < 5 for p in planes])


def render_torch(planes, tex_t, rig, T_wc, device, noise_seed=None, noise_sigma=2.0, chunk=16):
    """render() on torch tensors: one u8 [h, w] image on `device`.  Planes are intersected `chunk` at a time (elementwise ops on
    [chunk, h, w] tensors, float32), the nearest hit per pixel kept, ONE bilinear texture fetch per pixel at the end."""
    import torch
    P = planes if isinstance(planes, TorchPlanes) else TorchPlanes(planes, device, torch.float32)
    w, h = rig["w"], rig["h"]
    fx, fy, cx, cy = rig["fx"], rig["fy"], rig["cx"], rig["cy"]
    f32 = torch.float32
    v, u = torch.meshgrid(torch.arange(h, dtype=torch.float64, device=device), torch.arange(w, dtype=torch.float64, device=device), indexing="ij")
    R = T_wc[:3, :3]
    xc, yc = (u - cx) / fx, (v - cy) / fy
    d = [(R[k, 0] * xc + R[k, 1] * yc + R[k, 2]).to(f32) for k in range(3)]      # world ray directions (camera z = 1)
    t = [float(T_wc[k, 3]) for k in range(3)]
    tz = t[2]
    keep = np.nonzero(~(P.small & ((P.z < tz - 1.0) | (P.z > tz + 30.0))) if len(P.z) > 40 else np.ones(len(P.z), bool))[0]
    best = torch.full((h, w), float("inf"), dtype=f32, device=device)
    bA = torch.zeros((h, w), dtype=f32, device=device); bB = torch.zeros_like(bA)
    bI = torch.zeros((h, w), dtype=torch.long, device=device)
    tv = torch.as_tensor(t, dtype=f32, device=device)
    for c0 in range(0, len(keep), chunk):
        idx = torch.as_tensor(keep[c0:c0 + chunk], device=device)
        o, n, ex, ey = P.o[idx], P.n[idx], P.ex[idx], P.ey[idx]
        cc = ((o - tv) * n).sum(1)                                                  # [C]
        den = d[0] * n[:, 0, None, None] + d[1] * n[:, 1, None, None] + d[2] * n[:, 2, None, None]
        lam = cc[:, None, None] / den
        px = tv[0] + lam * d[0] - o[:, 0, None, None]; py = tv[1] + lam * d[1] - o[:, 1, None, None]; pz = tv[2] + lam * d[2] - o[:, 2, None, None]
        a = px * ex[:, 0, None, None] + py * ex[:, 1, None, None] + pz * ex[:, 2, None, None]
        b = px * ey[:, 0, None, None] + py * ey[:, 1, None, None] + pz * ey[:, 2, None, None]
        hit = (lam > 0.05) & (a.abs() <= P.hw[idx][:, None, None]) & (b.abs() <= P.hh[idx][:, None, None]) & torch.isfinite(lam)
        lam = torch.where(hit, lam, torch.full_like(lam, float("inf")))
        lc, ic = lam.min(0)
        upd = lc < best
        ac = torch.gather(a, 0, ic[None])[0]; bc = torch.gather(b, 0, ic[None])[0]
        best = torch.where(upd, lc, best); bA = torch.where(upd, ac, bA); bB = torch.where(upd, bc, bB); bI = torch.where(upd, idx[ic], bI)
    ts = tex_t.shape[0]
    hitAny = torch.isfinite(best)
    tu = torch.remainder(P.toff[bI, 0] + (bA + P.hw[bI]) * P.tsc[bI], ts - 1)
    tw = torch.remainder(P.toff[bI, 1] + (bB + P.hh[bI]) * P.tsc[bI], ts - 1)
    tu = torch.where(hitAny, tu, torch.zeros_like(tu)); tw = torch.where(hitAny, tw, torch.zeros_like(tw))
    x0 = tu.floor().long(); y0 = tw.floor().long()
    fxr = tu - x0; fyr = tw - y0
    x1 = torch.clamp(x0 + 1, max=ts - 1); y1 = torch.clamp(y0 + 1, max=ts - 1)
    out = (tex_t[y0, x0] * (1 - fxr) * (1 - fyr) + tex_t[y0, x1] * fxr * (1 - fyr) + tex_t[y1, x0] * (1 - fxr) * fyr + tex_t[y1, x1] * fxr * fyr)
    out = torch.where(hitAny, out, torch.zeros_like(out))
    if noise_seed is not None and noise_sigma > 0:
        g = torch.Generator(device=device)
        g.manual_seed(int(noise_seed))
        out = out + torch.randn(out.shape, generator=g, device=device, dtype=torch.float32) * noise_sigma
    return torch.clamp(torch.round(out), 0, 255).to(torch.uint8)


def corridor_sequence(rig_name, n_frames, device, frame_step=1, speed=0.5, first=0, scene_seed=11, tex_seed=0xC0FFEE):
    """n_frames stereo pairs along the corridor on `device`: (left [n, h, w] u8, right [n, h, w] u8, poses [n, 4, 4], frame indices)."""
    import torch
    rig = RIGS[rig_name]
    planes = TorchPlanes(make_corridor(scene_seed), device, torch.float32)
    tex_t = torch.from_numpy(texture(tex_seed).astype(np.float32)).to(device)
    ext = np.eye(4); ext[0, 3] = rig["bl"]
    idx = [first + frame_step * j for j in range(n_frames)]
    Ls = torch.empty((n_frames, rig["h"], rig["w"]), dtype=torch.uint8, device=device)
    Rs = torch.empty_like(Ls)
    poses = np.zeros((n_frames, 4, 4))
    for j, i in enumerate(idx):
        T = corridor_pose(i, rig["fps"], speed)
        poses[j] = T
        Ls[j] = render_torch(planes, tex_t, rig, T, device, noise_seed=1000 + 2 * i)
        Rs[j] = render_torch(planes, tex_t, rig, T @ ext, device, noise_seed=1001 + 2 * i)
    return Ls, Rs, poses, idx
