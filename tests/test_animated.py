"""AnimatedPrimitive (cpu/primitive.cpp:133-158): AnimatedTransform::Interpolate (util/transform.cpp:
1062-1081) restated by the oracle and pinned to the compiled reference, and the device's two-level
traversal with time-interpolated instance transforms against the oracle."""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
REF = os.path.join(HERE, "..", "oracle", "_ref", "ref_anim")


def anims_from_reference_output(rec, out):
    """ANIM_DTYPE records from a ref_anim output row (the members the reference object holds)."""
    o = out.view(np.float32)
    a = np.zeros(len(rec), ob.ANIM_DTYPE)
    a["start_m"], a["end_m"] = rec[:, :16], rec[:, 16:32]
    a["start_minv"], a["end_minv"] = o[:, 48:64], o[:, 64:80]
    a["T"] = o[:, 2:8].reshape(-1, 2, 3)
    a["R"] = o[:, 8:16].reshape(-1, 2, 4)
    a["S"] = o[:, 16:48].reshape(-1, 2, 16)
    a["start_time"], a["end_time"] = rec[:, 32], rec[:, 33]
    a["actually_animated"] = o[:, 0] != 0
    return a


def test_interpolate_matches_reference_vectors_bit_exact():
    g = np.load(os.path.join(HERE, "golden", "anim_interpolate.npz"))
    rec, out = g["inputs"], g["outputs"]
    a = anims_from_reference_output(rec, out)
    got = ob.anim_interpolate(a, rec[:, 34])
    exp = out[:, 80:112]
    bad = np.nonzero((got.view(np.uint32) != exp).any(1))[0]
    assert len(bad) == 0, f"{len(bad)} of {len(rec)} interpolated transforms differ, first {bad[:5]}"
    inside = (rec[:, 34] > rec[:, 32]) & (rec[:, 34] < rec[:, 33]) & (a["actually_animated"] != 0)
    assert inside.sum() > 800  # most cases exercise the Slerp / lerp / inverse path


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.isdir("/root/reference")),
                    reason="compiled reference harness only exists in the build container")
@pytest.mark.parametrize("seed", [1, 2])
def test_interpolate_equals_reference_live(seed):
    from make_anim_golden import cases, run_ref
    rec = cases(6000, seed)
    out = run_ref(rec)
    a = anims_from_reference_output(rec, out)
    got = ob.anim_interpolate(a, rec[:, 34])
    assert np.array_equal(got.view(np.uint32), out[:, 80:112])
