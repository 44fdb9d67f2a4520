"""TEST INFRASTRUCTURE (oracle): restatement of the dataset bookkeeping of the reference's main()
(src/VIOSlam.cpp:23-139 readers, :238-274 per-frame IMU buckets and the gravity guess)."""
import os


def read_image_csv(path):
    names, stamps = [], []
    with open(path) as f:
        f.readline()
        for line in f:
            line = line.rstrip("\n")
            tok = line.split(",")
            if not tok[0]:
                continue
            stamps.append(float(tok[0]))
            n = tok[1]
            if n.endswith("\r"):
                n = n[:-1]
            names.append(n)
    return names, stamps


def read_imu_csv(path):
    T, W, A = [], [], []
    with open(path) as f:
        f.readline()
        for line in f:
            tok = line.rstrip("\n").split(",")
            if not tok[0]:
                continue
            T.append(float(tok[0])); W.append([float(v) for v in tok[1:4]]); A.append([float(v) for v in tok[4:7]])
    return T, W, A


def kitti_names(directory):
    n = sum(1 for e in os.listdir(directory) if e.endswith(".png") and os.path.isfile(os.path.join(directory, e)))
    return ["%06d.png" % i for i in range(n)]


def imu_buckets(stamps, T, W, A):
    nF = len(stamps)
    buckets = [dict(ts=[], gyr=[], acc=[]) for _ in range(nF)]
    frame = 0
    ft, nt = stamps[0], stamps[1]
    for i, t in enumerate(T):
        if t > ft and t > nt:
            if frame + 1 >= nF:
                break
            frame += 1
            ft = stamps[frame]
            nt = stamps[frame + 1] if frame + 1 < nF else float("inf")
        if ft < t < nt:
            b = buckets[frame]
            b["ts"].append(t); b["gyr"].append(W[i]); b["acc"].append(A[i])
    g = None
    if buckets and buckets[0]["ts"]:
        a = buckets[0]["acc"][0]
        g = (a[1], -a[0], a[2])
    return buckets, g
