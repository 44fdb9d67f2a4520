"""The claim the device-side factor ordering rests on (DESIGN §0 "C5", k_ord_scatter / k_ord_rank), checked on the CPU with numpy:
filling a landmark's bucket in ANY arrival order and then placing every factor at the number of bucket entries that precede it in
(free index, 2 * pair + side) order gives exactly the list the host path builds - a counting sort by landmark that keeps the pair
order, followed by a stable insertion sort of each bucket by free index (BaPassHost::fill).  The GPU tests compare the two paths'
results end to end (tests/test_gpu_ba.py::test_ba_device_factor_ordering_equals_host_ordering); this one pins the argument itself."""
import numpy as np


def _host_order(pair_lm, pair_kf, pair_flags, fidx, n_lm):
    buckets = [[] for _ in range(n_lm)]
    for p in range(len(pair_lm)):                      # pair order, side 0 before side 1
        for side in range(2):
            if (pair_flags[p] >> side) & 1:
                buckets[pair_lm[p]].append((fidx[pair_kf[p]], 2 * p + side))
    out = []
    for b in buckets:                                  # stable insertion sort by free index
        for i in range(1, len(b)):
            k = b[i]
            j = i - 1
            while j >= 0 and b[j][0] > k[0]:
                b[j + 1] = b[j]
                j -= 1
            b[j + 1] = k
        out.extend(b)
    return out


def _device_order(pair_lm, pair_kf, pair_flags, fidx, n_lm, rng):
    entries = [(pair_lm[p], fidx[pair_kf[p]], 2 * p + side) for p in range(len(pair_lm)) for side in range(2) if (pair_flags[p] >> side) & 1]
    rng.shuffle(entries)                               # the atomics' arrival order
    buckets = [[] for _ in range(n_lm)]
    for l, k, v in entries:
        buckets[l].append((k, v))
    out = []
    for b in buckets:
        placed = [None] * len(b)
        for k, v in b:                                 # k_ord_rank: count the entries that precede (k, v)
            r = sum(1 for kg, vg in b if kg < k or (kg == k and vg < v))
            placed[r] = (k, v)
        out.extend(placed)
    return out


def test_rank_by_total_order_equals_stable_counting_sort():
    rng = np.random.default_rng(5)
    for trial in range(20):
        n_lm, n_kf, n_pairs = int(rng.integers(1, 40)), int(rng.integers(2, 12)), int(rng.integers(0, 400))
        pair_lm = rng.integers(0, n_lm, n_pairs)
        pair_kf = rng.integers(0, n_kf, n_pairs)
        pair_flags = rng.integers(0, 4, n_pairs)
        fixed = rng.random(n_kf) < 0.3
        fidx = np.where(fixed, -1, np.cumsum(~fixed) - 1)
        assert _device_order(pair_lm, pair_kf, pair_flags, fidx, n_lm, rng) == _host_order(pair_lm, pair_kf, pair_flags, fidx, n_lm)
