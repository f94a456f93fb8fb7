"""CPU: the oracle (oracle/rt_oracle.c) against the golden vectors produced by the reference's own kernels.

Bar: exact float equality (helpers.same_floats), both contraction flavours, every fixture (geometry AND colour - on the CPU both sides use
the same libm powf)."""
import numpy as np
import pytest

from helpers import count_float_mismatches, fixture_names, load_fixture, same_floats

NAMES = fixture_names()


def test_fixture_set_present():
    assert len(NAMES) >= 60


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", NAMES)
def test_restatement_matches_reference_golden(name, fused, restatement):
    fx = load_fixture(name)
    res = restatement[fused].render(fx["kernel"], fx["objs"], fx["lights"], fx["rays"], fx["max_bounces"])
    got = res["out"] if fx["kernel"] == 0 else np.ascontiguousarray(res["out"][:, :3])
    want = fx["out_fused"] if fused else fx["out_unfused"]
    assert got.shape == want.shape
    assert same_floats(got, want), f"{name}: {count_float_mismatches(got, want)} differing values"


def test_index_recovery_fixture_pins_hit_index(restatement):
    """In the index-recovery scene the reference's own `shade` output IS the hit index
    (light ambient 1, material ambient = (i+1)/256): compare it with the restatement's index output."""
    fx = load_fixture("index_recovery_shade")
    res = restatement[True].render(1, fx["objs"], fx["lights"], fx["rays"], 0)
    ref_rgb = fx["out_fused"]
    ref_index = np.rint(ref_rgb[:, 0] * 256 + ref_rgb[:, 1] * 65536).astype(np.int64) - 1
    assert np.array_equal(ref_index, res["hit_index"].astype(np.int64))
    assert (ref_index >= 0).sum() > 500


def test_tie_break_rules_from_reference():
    """Q3: later sphere wins ties, earlier box wins ties - read off the reference's outputs."""
    two_s = load_fixture("tie_two_spheres")["out_fused"]
    two_b = load_fixture("tie_two_boxes")["out_fused"]
    sbs = load_fixture("tie_sphere_box_sphere")["out_fused"]
    assert set(np.unique(two_s[:, 0])) == {0.0, np.float32(2 / 256)}
    assert set(np.unique(two_b[:, 0])) == {0.0, np.float32(1 / 256)}
    # unit sphere inside its bounding box: the box face is nearer everywhere except the tangent points,
    # where a tie would go to the LAST sphere (index 2), never to the first (index 0)
    vals = set(np.unique(sbs[:, 0]))
    assert np.float32(1 / 256) not in vals and np.float32(2 / 256) in vals
