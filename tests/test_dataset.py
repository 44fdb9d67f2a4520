"""N3 dataset bookkeeping (host only, runs without a GPU): the library's EuRoC / KITTI readers and per-frame IMU bucketing
against the oracle's restatement of the reference's main() (src/VIOSlam.cpp:23-139, 238-274) on small fixture trees
written here (the reference ships no data files)."""
import os
import numpy as np
import pytest


def _euroc_tree(root, n_frames=6, imu_per_frame=10, crlf=False):
    os.makedirs(os.path.join(root, "cam0", "data")); os.makedirs(os.path.join(root, "cam1", "data")); os.makedirs(os.path.join(root, "imu0"))
    t0, dt = 1403636579763555584, 50000000
    eol = "\r\n" if crlf else "\n"
    with open(os.path.join(root, "cam0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp [ns],filename" + eol)
        for i in range(n_frames):
            f.write("%d,%d.png%s" % (t0 + i * dt, t0 + i * dt, eol))
    rng = np.random.default_rng(3)
    with open(os.path.join(root, "imu0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_RS_S_x,w_RS_S_y,w_RS_S_z,a_RS_S_x,a_RS_S_y,a_RS_S_z\n")
        t = t0 - 2 * dt // imu_per_frame + 1234          # a few samples before the first frame, none exactly on a frame
        while t < t0 + (n_frames + 1) * dt:
            v = rng.normal(size=6)
            f.write("%d,%s\n" % (t, ",".join("%.9f" % x for x in v)))
            t += dt // imu_per_frame
    return root + "/", os.path.join(root, "imu0") + "/"


@pytest.mark.parametrize("crlf", [False, True])
def test_euroc_reader_and_imu_buckets(capi, tmp_path, crlf):
    import dataset as od
    images, imu = _euroc_tree(str(tmp_path / "mav0"), crlf=crlf)
    names, stamps = od.read_image_csv(images + "cam0/data.csv")
    T, W, A = od.read_imu_csv(imu + "data.csv")
    ref, g = od.imu_buckets(stamps, T, W, A)
    ds = capi.Dataset(0, images, imu)
    assert len(ds) == len(names) == 6
    for i in range(len(ds)):
        l, r, t = ds.frame(i)
        assert l == images + "cam0/data/" + names[i] and r == images + "cam1/data/" + names[i] and t == stamps[i]
        assert not l.endswith("\r")
        acc, gyr, ts = ds.imu_bucket(i)
        assert np.array_equal(ts, np.array(ref[i]["ts"])) and np.array_equal(acc.reshape(-1, 3), np.array(ref[i]["acc"]).reshape(-1, 3))
        assert np.array_equal(gyr.reshape(-1, 3), np.array(ref[i]["gyr"]).reshape(-1, 3))
    assert all(len(ref[i]["ts"]) in (9, 10) for i in range(5))            # strictly between the frame stamps
    valid, grav = ds.gravity()
    assert valid and grav == g
    ds.close()


def test_kitti_listing_counts_png_files(capi, tmp_path):
    import dataset as od
    root = str(tmp_path / "kitti")
    os.makedirs(os.path.join(root, "image_0")); os.makedirs(os.path.join(root, "image_1"))
    for n in ("000000.png", "000001.png", "junk.txt", "000007.png"):        # names are GENERATED from the png count (:129-137)
        open(os.path.join(root, "image_0", n), "w").close()
    os.makedirs(os.path.join(root, "image_0", "dir.png"))                  # not a regular file
    ds = capi.Dataset(1, root + "/")
    assert len(ds) == 3 == len(od.kitti_names(os.path.join(root, "image_0")))
    assert [os.path.basename(ds.frame(i)[0]) for i in range(3)] == ["000000.png", "000001.png", "000002.png"]
    assert ds.frame(2)[1] == root + "/image_1/000002.png" and ds.gravity()[0] is False
    ds.close()


def test_dataset_errors(capi, tmp_path):
    with pytest.raises(RuntimeError):
        capi.Dataset(0, str(tmp_path) + "/nothing/")
    with pytest.raises(RuntimeError):
        capi.Dataset(2, str(tmp_path) + "/")
