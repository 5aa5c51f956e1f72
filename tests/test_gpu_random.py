"""Seeded random sweep of the two-phase encoder against the oracle: models x block sizes (1 .. > 64 KiB, ragged tails) x data
kinds (text, runs, random bytes, few symbols).  Catches indexing slips in the batched / tiled predict kernels that the
fixed-shape tests would miss."""
import numpy as np
import pytest

from tests.synth import lcg_text, markov_text, mixed_bytes
from tests.test_gpu_parity import check_blocks, ctx, decode_both, pair  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

import os

SOAK = int(os.environ.get("W3_RANDOM_SOAK", "1"))   # W3_RANDOM_SOAK=8: eight times as many cases (a one-off soak, not the driver's run)

MODELS = ["order0", "order1", "order2", "best012", "best_ac_wide", "main_default", "ordern_5_3", "best_right"]


def make_data(rng, kind, n):
    if kind == 0:
        return markov_text(n, seed=int(rng.integers(1, 1 << 30)))
    if kind == 1:
        return lcg_text(n, seed=int(rng.integers(1, 1 << 30)))
    if kind == 2:
        return mixed_bytes(n, seed=int(rng.integers(1, 1 << 30)))
    if kind == 3:                                   # long runs of few symbols: big groups, Counter counts in the thousands
        sym = rng.integers(0, 256, 3, dtype=np.uint8)
        return bytes(np.repeat(sym[rng.integers(0, 3, n // 50 + 1)], 50)[:n])
    return rng.integers(0, 256, n, dtype=np.uint8).tobytes()   # incompressible: every group tiny


@pytest.mark.parametrize("case", range(36 * SOAK))
def test_random_shapes(ctx, oracle, case):  # noqa: F811
    rng = np.random.default_rng(1000 + case)
    name = MODELS[case % len(MODELS)]
    n = int(rng.integers(4, 200_000))
    bs = int(rng.choice([1, 3, 17, 64, 100, 511, 512, 2047, 2048, 2049, 4096, 10_000, 65_535, 65_536, 65_537, 70_001, 150_000]))
    if n // bs > 3000:                               # keep the oracle run short
        bs = max(bs, n // 3000 + 1)
    data = make_data(rng, case % 5, n)
    out, lens = check_blocks(ctx, oracle, name, data, bs, "twophase")
    # and back: k_decode_spec (sixteen lanes per block) and the lane-per-block decoder must both return the input
    assert decode_both(ctx, pair(oracle, name)[0](), out, lens, bs, len(data)).tobytes() == data


CM_MODELS = ["slot1", "slot2", "slot_mix", "o012_apm", "apm_chain4", "full_cm_small_tables", "apm1_order0_r3", "slot7"]


@pytest.mark.parametrize("case", range(16 * SOAK))
def test_random_shapes_cm(ctx, oracle, case):  # noqa: F811
    """The same sweep over the CM models: slot-state leaves on both forms (sorted replay and k_slot), APM chains, the two-phase encoder
    against k_cm, both decoders (tests/test_gpu_cm.py::check does all of that per case)."""
    from tests.test_gpu_cm import check
    rng = np.random.default_rng(5000 + case)
    name = CM_MODELS[case % len(CM_MODELS)]
    n = int(rng.integers(8, 120_000))
    bs = int(rng.choice([8, 17, 64, 100, 511, 512, 2049, 4096, 10_000, 65_536, 70_001]))
    if n // bs > 1500:
        bs = max(bs, n // 1500 + 1)
    data = make_data(rng, case % 5, n)
    check(ctx, oracle, name, data, bs)
