"""next-4 global augmentation: the numpy restatement and the host-side draw order against outputs of the reference's own functions
(tests/golden/augment.npz, made by tests/golden/make_golden.py::gen_augment under seeded np.random)."""
import numpy as np
import pytest

from oracle import augment_oracle as A
from partner_amd.augment import GlobalAugment

CASES = [0, 1, 2, 3]


def draw_for(g, case):
    std = g[f"std_{case}"]
    aug = GlobalAugment(global_rot_noise=[-0.78539816, 0.78539816], global_scale_noise=(0.95, 1.05),
                        global_translate_std=float(std[0]) if std.size == 1 else list(std))
    np.random.seed(int(g[f"seed_{case}"]))
    return aug.draw()


@pytest.mark.parametrize("case", CASES)
def test_oracle_and_draw_order_match_the_reference(golden, case):
    g = golden("augment.npz")
    d = draw_for(g, case)
    p, b = A.global_augment(g[f"pts_{case}"], g[f"boxes_{case}"], d.flip_y, d.flip_x, d.rotation, d.scale, d.translate)
    np.testing.assert_array_equal(p, g[f"pts_out_{case}"])          # same numpy operations in the same order: bit for bit
    np.testing.assert_array_equal(b, g[f"boxes_out_{case}"])
    assert (d.translate is None) == (not g[f"std_{case}"].any())


def test_draws_cover_both_flip_states(golden):
    g = golden("augment.npz")
    flips = {(draw_for(g, c).flip_y, draw_for(g, c).flip_x) for c in CASES}
    assert len(flips) >= 2
