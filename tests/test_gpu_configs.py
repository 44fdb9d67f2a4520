"""The tracking stages on the BASELINE configs other than the EuRoC rig (BASELINE.json configs):
  C3  KITTI-00-like  1241x376, 2000 features, 64x20 matching grid, fx 718.9, baseline 0.537
  C5  synthetic      1920x1200, 4000 features (~4400 keypoints per image)
Stereo matching, projection matching, the pose solve, the whole tracking loop and the new-point pipeline run through
the C ABI and are compared with the oracle exactly as on the EuRoC rig (same helpers, other rig / feature count)."""
import numpy as np
import pytest
import synth

from test_gpu_proj import _projection_parity
from test_gpu_pose import _pose_lm_parity
from test_gpu_track import _track_frame_parity
from test_gpu_newpts import _find_new_points_parity
from test_gpu_stereo import _run_pair, _assert_same

pytestmark = pytest.mark.gpu

CONFIGS = [("kitti", 2000), ("synthetic", 4000)]


@pytest.mark.parametrize("rig_name,nfeat", CONFIGS)
def test_stereo_parity_config(oracle, capi, rig_name, nfeat):
    L, R, _ = synth.stereo_frame(3, rig_name)
    ref, got = _run_pair(oracle, capi, L, R, rig_name, nfeat)
    _assert_same(ref, got)
    assert (ref["rightIdxs"] >= 0).sum() > 300


@pytest.mark.parametrize("rig_name,nfeat", CONFIGS)
@pytest.mark.parametrize("rad,jitter,dup", [(10.0, 6.0, 0), (4.0, 2.0, 150), (120.0, 40.0, 300)])
def test_projection_parity_config(oracle, capi, rig_name, nfeat, rad, jitter, dup):
    # C5: more map points than the 1024 the parallel fixed point holds in registers -> the blocked / sequential walk
    _projection_parity(oracle, capi, rad, jitter, dup, rig_name=rig_name, nfeat=nfeat, n_mps=900 if rig_name == "kitti" else 2600)


@pytest.mark.parametrize("rig_name,nfeat", CONFIGS)
@pytest.mark.parametrize("seed,shared", [(0, 0), (1, 40)])
def test_pose_lm_parity_config(oracle, capi, rig_name, nfeat, seed, shared):
    _pose_lm_parity(oracle, capi, seed, shared, rig_name=rig_name, nfeat=nfeat)


@pytest.mark.parametrize("rig_name,nfeat", CONFIGS)
@pytest.mark.parametrize("f0,f1,frame_number", [(4, 5, 7), (6, 7, 1)])
def test_track_frame_parity_config(oracle, capi, rig_name, nfeat, f0, f1, frame_number):
    ge, rep = _track_frame_parity(oracle, capi, f0, f1, frame_number, rig_name=rig_name, nfeat=nfeat)
    # the suppression of every level ran in k_ssc: no host redo (1920x1200 level 0 has > 8192 FAST candidates)
    assert ge.ssc_stats() == (1, 0)


@pytest.mark.parametrize("rig_name,nfeat,frames", [("kitti", 2000, (15, 12, 9, 6, 3)), ("synthetic", 4000, (30, 24, 18, 12, 6))])
def test_find_new_points_parity_config(oracle, capi, rig_name, nfeat, frames):
    _find_new_points_parity(oracle, capi, frames, 0, rig_name=rig_name, nfeat=nfeat)
