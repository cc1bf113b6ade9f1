"""Reference-noise variants of the oracle (oracle/ref_noise.hpp, ORC_VAR_*) and the committed envelope
tests/golden/noise_envelope.json (made by scripts/make_noise_envelope.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import rot_trans_err


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "closed_forms.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def envelope():
    with open(os.path.join(GOLDEN, "noise_envelope.json")) as f:
        return json.load(f)


def test_francis_qr_eigenvalues_match_numpy(oracle):
    rng = np.random.default_rng(3)
    for n in (2, 3, 4):
        for _ in range(100):
            A = rng.normal(size=(n, n))
            ref = np.linalg.eigvals(A)
            rc, ev = oracle.test_eigenvalues(A)                       # instantiated in double
            assert rc == 0
            assert max(min(abs(e - r) for r in ref) for e in ev) < 1e-10
            rc, ev = oracle.test_eigenvalues(A, use_f32=True)         # the same code in float: f32-sized noise, nothing worse
            assert rc == 0
            assert max(min(abs(e - r) for r in ref) for e in ev) < 3e-4 * max(1.0, np.abs(ref).max())
    # real eigenvalues come out with an imaginary part of exactly zero (what cvo.cpp:326 tests)
    rc, ev = oracle.test_eigenvalues(np.array([[6.0, -11.0, 6.0], [1, 0, 0], [0, 1, 0]]), use_f32=True)
    assert rc == 0 and np.all(ev.imag == 0) and np.allclose(sorted(ev.real), [1, 2, 3], rtol=1e-5)


def test_pade_logm_matches_scipy(oracle):
    from scipy.linalg import expm, logm
    rng = np.random.default_rng(4)
    worst = 0.0
    for _ in range(100):
        X = rng.normal(size=(4, 4)) * rng.uniform(0.001, 1.2)
        A = expm(X)
        ref = logm(A)
        if np.abs(np.imag(ref)).max() > 1e-9:
            continue
        rc, L = oracle.test_logm(A)
        assert rc == 0
        worst = max(worst, np.abs(L - ref.real).max() / max(1e-3, np.abs(ref).max()))
    assert worst < 1e-6          # single-precision Pade thresholds (degree <= 5): ~1e-8 in double, never more


def test_f32_companion_roots_follow_the_exact_roots(oracle, kat):
    # cvo.cpp:76-92,324-330 in f32: the same step as the exact solver up to the eigen-solver's f32 noise (percents when
    # the wanted root is six orders of magnitude smaller than the other two: the golden set holds such cases), same fallbacks
    rel = []
    for case in kat["cubic"]:
        c = case["coef"]
        a, b = oracle.cubic_step(*c), oracle.cubic_step_f32eig(*c)
        rel.append(abs(a - b) / max(abs(a), 1e-12))
        assert abs(a - b) <= 1e-1 * max(abs(a), 1e-6), (c, a, b)
    assert np.median(rel) < 1e-5
    assert oracle.cubic_step_f32eig(0.0, 1.0, -1.0, 0.5, 0.2) == pytest.approx(0.2)      # E = 0: no finite companion matrix


def test_f32_logm_distance_noise(oracle, kat):
    # cvo.cpp:94-104 in f32: ~1e-7 absolute noise on a norm compared with eps_2 = 1e-5
    for case in kat["dist"]:
        got = oracle.dist_se3_f32logm(case["dR"], case["dT"])
        assert got == pytest.approx(case["frob_log"], rel=2e-3, abs=3e-7), case


def _align(oracle, pair, **kw):
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, **kw)
    o.set_pcd(pair.fixed.xyz, pair.fixed.feat); o.set_pcd(pair.moving.xyz, pair.moving.feat)
    assert o.align()[0] == 0
    st = o.get_state()
    return st["transform"], st["iter"]


def test_variants_reproduce_the_committed_envelope(oracle, envelope):
    """The parity-build variants are deterministic: re-running them on the fixture's first small pair gives the recorded
    distances; the envelope file is what the script writes (64 bench pairs, every variant)."""
    from cvo_slam_amd import synth
    assert len(envelope["tum64"]["per_pair"]) == 64 and len(envelope["variants"]) == 14
    names = {v["name"]: v for v in envelope["variants"]}
    assert names["row_eigen337_lazy16"]["flags"] == 8 and names["row_stride4"]["flags"] == 16 and names["row_stride8"]["flags"] == 32     # ORC_VAR_ROW_*
    assert names["row_alpha_first"]["flags"] == 64 and names["feat_hadd"]["flags"] == 128 and names["feat_movehl"]["flags"] == 256
    row = envelope["small"]["per_pair"][0]
    pair = synth.make_small_pair(row["pair"], n=600)
    base, it0 = _align(oracle, pair)
    assert it0 == row["base_iter"]
    worst_r = worst_t = 0.0
    for v in envelope["variants"]:
        if v["build"] != "parity":
            continue                     # the -march=native build belongs to the machine that made the fixture
        tf, _ = _align(oracle, pair, variant=v["flags"], shuffle_seed=v["shuffle_seed"])
        r, t = rot_trans_err(tf, base)
        worst_r, worst_t = max(worst_r, r), max(worst_t, t)
    assert worst_r <= row["max_rot_rad"] + 1e-12 and worst_t <= row["max_trans_m"] + 1e-12


def test_fast_build_agrees_within_the_reference_noise(oracle):
    """-O3 -march=native -ffp-contract=fast build of the same source (the FMA member of the envelope and the timed CPU
    baseline): same answer up to the contraction noise, far inside 1e-3."""
    from cvo_slam_amd import synth
    pair = synth.make_small_pair(12, n=600)
    base, _ = _align(oracle, pair)
    fast, _ = _align(oracle, pair, flavor="fast")
    r, t = rot_trans_err(fast, base)
    assert r < 1e-3 and t < 1e-3


def test_row_and_feature_order_variants(oracle):
    """ORC_VAR_ROW_* / ORC_VAR_FEAT_* (cvo.cpp:222-223, :169, :662 are Eigen reductions whose order Eigen picks).  On features as the
    reference's generator makes them -- 8-bit B, G, R and half-integer gradients (pcd_generator.cpp:601-609) -- every term of the colour
    reductions is a multiple of 1/4 below 2^18 and every partial sum is exact: the feature orders give the base oracle's bits.  On
    arbitrary float features (the bench's synthetic gradients come from a float gray image) they do not; the row orders never do."""
    from cvo_slam_amd import synth
    import copy
    pair = synth.make_small_pair(13, n=600)
    base, it0 = _align(oracle, pair)
    q = copy.deepcopy(pair)
    for cl in (q.fixed, q.moving):
        cl.feat[:3] = np.round(cl.feat[:3]); cl.feat[3:] = np.round(2 * cl.feat[3:]) / 2
    qbase, qit = _align(oracle, q)
    for flags in (128, 256):
        tf, it = _align(oracle, q, variant=flags)
        assert it == qit and np.array_equal(tf, qbase), flags                                   # exact arithmetic: same bits
    moved = 0
    for flags in (8, 16, 32, 64, 128, 256):
        tf, it = _align(oracle, pair, variant=flags)
        r, t = rot_trans_err(tf, base)
        assert r < 1e-3 and t < 1e-3, (flags, r, t)                                             # noise, not another answer
        tf2, it2 = _align(oracle, pair, variant=flags)
        assert np.array_equal(tf, tf2) and it == it2                                            # deterministic
        moved += int(not np.array_equal(tf, base))
    assert moved >= 4                                                                           # they really are other float sequences


def test_envelope_statement(envelope):
    """What DESIGN.md section 2 says about the envelope is what the file holds."""
    e = envelope["tum64"]
    assert e["per_variant"]["shuffled_reduction_1"]["max_rot_rad"] < 1e-6          # f64 cross-row sums: order does not reach the pose
    assert e["max_rot_rad"] < 1e-3 and e["max_trans_m"] < 1e-3
    # the within-row / within-feature orders are as loud as FMA contraction and the f32 roots: the same 1e-4 scale, never 1e-3
    for name in ("row_eigen337_lazy16", "row_alpha_first", "row_stride4", "row_stride8", "feat_hadd", "feat_movehl"):
        assert 1e-6 < e["per_variant"][name]["max_trans_m"] < 1e-3, name
    assert e["pairs_beyond_1e-4"] >= 5                                              # the north-star tolerance is tighter than the reference's own freedom on some pairs
    assert e["max_trans_m"] > 1e-4                                                 # the reference's own noise exceeds the north-star tolerance
