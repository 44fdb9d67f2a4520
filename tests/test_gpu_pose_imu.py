"""GPU parity of the stereo + IMU (C2) pose solve — IMU pre-integration, CombinedImuFactor, 15-dof LM and the
chi2 inlier pass in HIP (through the C ABI) vs the CPU oracle.  Bar: pose within 1e-6 relative (asserted 1e-8
absolute on T_cw), velocity / bias within 1e-8, identical iteration counts and inlier decisions."""
import numpy as np
import pytest
import synth
from test_gpu_pose import _scene

pytestmark = pytest.mark.gpu
G = (0.0, 9.81, 0.0)
NOISE = (1.6968e-4, 1.9393e-5, 2.0e-3, 3.0e-3)      # gyro density, gyro walk, acc density, acc walk


@pytest.mark.parametrize("seed,bias", [(0, 0.0), (3, 0.01)])
def test_pose_imu_parity(oracle, capi, seed, bias):
    frame = 6
    rig, oL, (kL, dL, kR, dR), st, m, pts, matches, inF, inFR, mpo, out0, T_wc = _scene(oracle, capi, frame=frame, seed=seed)
    T_prev = synth.pose_at(frame - 1)
    h = 1e-4
    v_prev = (synth.pose_at(frame - 1 + h * 20)[:3, 3] - synth.pose_at(frame - 1 - h * 20)[:3, 3]) / (2 * h)
    b_prev = np.full(6, bias) * np.array([1, -1, 0.5, 0.1, -0.1, 0.05])
    S, dts, _ = synth.imu_samples(frame - 1, frame, noise_seed=seed + 1, bias=b_prev)
    ts = np.arange(len(dts)) * 5e6                                     # 200 Hz timestamps in ns
    prm = oracle.imu_params(G, NOISE[0], NOISE[2], NOISE[1], NOISE[3], synth.T_BC1)
    ref = oracle.estimate_pose_imu(rig, oL.InvSigmaFactor, pts, inF, inFR, mpo, matches, out0, kL, kR, st["rightIdxs"],
                                   st["leftIdxs"], st["depth"], st["close"], prm, T_prev, v_prev, b_prev, S, dts)
    got = capi.estimate_pose_imu(m, pts, inF, inFR, mpo, matches, out0, G, NOISE, synth.T_BC1, T_prev, v_prev, b_prev,
                                 S[:, :3], S[:, 3:], ts, 200)
    assert ref["iterations"] >= 2
    assert (got["iterations"], got["inner"]) == (ref["iterations"], ref["inner"])
    assert np.abs(got["T_cw"] - ref["T_cw"]).max() < 1e-8
    assert np.abs(got["vel"] - ref["vel"]).max() < 1e-8 and np.abs(got["bias"] - ref["bias"]).max() < 1e-9
    assert abs(got["finalError"] - ref["finalError"]) <= 1e-8 * max(1.0, ref["finalError"])
    assert abs(got["initialError"] - ref["initialError"]) <= 1e-9 * max(1.0, ref["initialError"])
    assert (got["nIn"], got["nStereo"]) == (ref["nIn"], ref["nStereo"])
    assert np.array_equal(got["outliers"], ref["outliers"]) and np.array_equal(got["matches"], ref["matches"])
    st2 = m.stereo_fetch(len(kL), len(kR))
    assert np.array_equal(st2["rightIdxs"], ref["rightIdxs"]) and np.array_equal(st2["close"], ref["close"])


def test_pose_imu_single_sample_bucket_and_no_vision(oracle, capi):
    """One IMU sample (dt = 1/Hz rule) and no vision factor: the solve stays at the IMU prediction."""
    frame = 6
    rig, oL, (kL, dL, kR, dR), st, m, pts, matches, inF, inFR, mpo, out0, T_wc = _scene(oracle, capi, frame=frame)
    T_prev = synth.pose_at(frame - 1)
    S, dts, _ = synth.imu_samples(frame - 1, frame)
    none = np.full_like(matches, -1)
    prm = oracle.imu_params(G, NOISE[0], NOISE[2], NOISE[1], NOISE[3], synth.T_BC1)
    ref = oracle.estimate_pose_imu(rig, oL.InvSigmaFactor, pts, inF, inFR, mpo, none, out0, kL, kR, st["rightIdxs"],
                                   st["leftIdxs"], st["depth"], st["close"], prm, T_prev, [0.1, 0, 0.2], np.zeros(6), S[:1], [1.0 / 200])
    got = capi.estimate_pose_imu(m, pts, inF, inFR, mpo, none, out0, G, NOISE, synth.T_BC1, T_prev, [0.1, 0, 0.2], np.zeros(6),
                                 S[:1, :3], S[:1, 3:], np.array([0.0]), 200)
    assert got["iterations"] == 0 == ref["iterations"]
    assert np.abs(got["T_cw"] - ref["T_cw"]).max() < 1e-12 and np.abs(got["vel"] - ref["vel"]).max() < 1e-12
    with pytest.raises(capi.VslamError):
        capi.estimate_pose_imu(m, pts, inF, inFR, mpo, none, out0, G, NOISE, synth.T_BC1, T_prev, [0, 0, 0], np.zeros(6),
                               S[:0, :3], S[:0, 3:], np.zeros(0), 200)
