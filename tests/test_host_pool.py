"""Host-side worker pool of the local-BA preparation (gtsam-vslam_amd/csrc/ba_pool.hpp) under ThreadSanitizer:
back-to-back run() calls must execute every task exactly once and never hang (CPU build only; the GPU pool
offers no sanitizers)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ba_pool_tsan():
    src = os.path.join(ROOT, "tests", "native", "pool_tsan.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "pool_tsan")
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.join(ROOT, "gtsam-vslam_amd", "csrc"),
                        src, "-o", exe, "-lpthread"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.startswith("ok") and "WARNING: ThreadSanitizer" not in r.stderr


def test_job_engine_tsan():
    """The cohort engine of the lockstep groups' local mapping (job_engine.hpp: release of a phase's jobs by several producer threads,
    cohorts taken by two kinds of engine threads, drain at shutdown) under ThreadSanitizer."""
    src = os.path.join(ROOT, "tests", "native", "engine_tsan.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "engine_tsan")
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.join(ROOT, "gtsam-vslam_amd", "csrc"),
                        src, "-o", exe, "-lpthread"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.startswith("ok") and "WARNING: ThreadSanitizer" not in r.stderr
