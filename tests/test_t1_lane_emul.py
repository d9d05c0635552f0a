"""CPU: the per-lane EBCOT Tier-1 code the HIP kernels run (qsvc_amd/csrc/t1_core.h, compiled for the host by
tests/emul) against the restatement's Tier-1: fused form and split form (bit-plane-parallel context modelling
from magnitude-derived initial flags + MQ over the decision segments)."""
import numpy as np
import pytest

J = pytest.importorskip("oracle.j2k")


@pytest.mark.parametrize("split", [False, True, "cm2"])
def test_lane_t1_equals_restatement(split):
    """split: False = fused lane form, True = bit-plane-parallel lane form, "cm2" = sample-parallel context modelling
    (t1_cm2.h: membership recurrence + independent stripe columns + prefix-summed emission), all followed by the MQ coder."""
    from emul import emul as E
    rng = np.random.default_rng(3 + (2 if split == "cm2" else int(split)))
    for trial in range(120):
        w = int(rng.choice([64, 64, 64, 33, 17, 5, 1, 48]))
        h = int(rng.choice([64, 64, 32, 7, 4, 3, 1, 61]))
        orient = int(rng.integers(0, 4))
        mode = trial % 5
        if mode == 0:
            v = rng.integers(-300, 300, size=(h, w))
        elif mode == 1:
            v = rng.integers(-3, 4, size=(h, w)) * (rng.random((h, w)) < 0.05)
        elif mode == 2:
            v = rng.integers(-32767, 32768, size=(h, w))
        elif mode == 3:
            v = np.rint(rng.laplace(0, 8, size=(h, w))).astype(np.int64)
        else:
            v = np.zeros((h, w), np.int64)
            v[rng.integers(0, h), rng.integers(0, w)] = int(rng.integers(-1000, 1000))
        v = v.astype(np.int32)
        a, b = J.t1_block(v, orient), E.t1_block(v, orient, split=split is True, cm2=split == "cm2")
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (trial, w, h, orient, mode)
        np.testing.assert_array_equal(a[3], b[3])
