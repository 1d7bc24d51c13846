"""Seeded fuzz of both curves against the oracles: random sizes, repeated / opposite / identical points,
zero and tiny scalars, scalars drawn from a small pool (long rows -> split work items and merges), digit
patterns at the window boundaries.  Bit-exact.  Run with `pytest -m gpu`."""
import random

import pytest

import pyref as R
import util

pytestmark = pytest.mark.gpu


def fuzz_case(rnd, n, order, pick_points):
    pts = pick_points(rnd, n)
    mode = rnd.randrange(5)
    pool = [rnd.randrange(order) for _ in range(rnd.choice([1, 2, 5, 17]))]
    ks = []
    for _ in range(n):
        r = rnd.random()
        if mode == 0 or r < 0.5:
            k = rnd.randrange(order)
        elif r < 0.6:
            k = 0
        elif r < 0.7:
            k = rnd.randrange(1 << 16)
        elif r < 0.8:
            k = sum(rnd.choice([0, 1, 0x7FFF, 0x8000, 0x8001, 0xFFFF]) << (16 * w) for w in range(15)) % order
        else:
            k = rnd.choice(pool)
        ks.append(k)
    return pts, ks


def g1_points(oracle):
    base = R.decode_points(util.oracle_gen_points(oracle, 64, 0xF00D, 0xBEEF))

    def pick(rnd, n):
        out = []
        for _ in range(n):
            p = rnd.choice(base)
            out.append(R.neg(p) if rnd.random() < 0.3 else p)
        return out

    return pick


def ed_points(oracle):
    raw = util.oracle_ed_gen_points(oracle, 64, 0xF00D, 0xBEEF)
    base = [(int.from_bytes(raw[64 * i : 64 * i + 32], "little"), int.from_bytes(raw[64 * i + 32 : 64 * i + 64], "little")) for i in range(64)]

    def pick(rnd, n):
        out = []
        for _ in range(n):
            p = rnd.choice(base)
            out.append(R.ed_neg(p) if rnd.random() < 0.3 else p)
        return out

    return pick


@pytest.mark.parametrize("seed", range(24))
def test_g1_fuzz(engine, oracle, seed):
    rnd = random.Random(0x377000 + seed)
    n = rnd.choice([1, 2, 3, 7, 63, 64, 65, 129, 500, 1500, 3000])
    pts, ks = fuzz_case(rnd, n, R.R_ORDER, g1_points(oracle))
    pb, sb = R.encode_points(pts), R.encode_scalars(ks)
    exp = util.oracle_msm(oracle, pb, sb)
    try:
        for form, glv in (("edwards", "auto"), ("weierstrass", True), ("weierstrass", False)):
            engine.set_g1_form(form)  # twisted Edwards form (default); Weierstrass XYZZ behind GLV; plain 16 windows
            engine.set_glv(glv)
            assert engine.msm(pb, sb) == exp, (form, glv)
    finally:
        engine.set_g1_form("edwards")
        engine.set_glv("auto")


@pytest.mark.parametrize("seed", range(12))
def test_ed_fuzz(engine, oracle, seed):
    rnd = random.Random(0xED0000 + seed)
    n = rnd.choice([1, 2, 5, 64, 65, 200, 1000, 2500])
    pts, ks = fuzz_case(rnd, n, R.ED_SUBGROUP, ed_points(oracle))
    pb, sb = R.ed_encode_points(pts), R.encode_scalars(ks)
    assert engine.ed_msm(pb, sb) == util.oracle_ed_msm(oracle, pb, sb)
