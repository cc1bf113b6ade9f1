"""Device known-answer tests of the pair arithmetic itself (cvo.cpp:166-175) and of the float routines the epilogue and the gates call.

`cvo_selftest_pair_values` runs the align kernel's own device functions -- se_kernel_value (dense fallback), colour_factors +
se_kernel_value_ck (lists outside the polynomial's range), colour_factors + se_kernel_values_flat (12-term chain) and colour_factors +
se_kernel_values_flat7 (degree-7 polynomial with its rounding guard: what the steady candidate walk evaluates) -- on caller-supplied pairs.
Every route must give, BIT FOR BIT, what the oracle's `(float)(s2*exp(-d2/(2.0*l*l)))` sequence gives (oracle/cvo_oracle.cpp: orc_pair_values =
the statements of se_kernel_clouds on given distances): >= 1e7 samples over the four length-scales of cvo.cpp:810-812, among them pairs within a
few ulps of d2_thres, of d2_c_thres and of a == sp_thres.  The kernels use a reciprocal multiply where the reference divides, an inline
polynomial where it calls exp, OCML where it calls glibc: this is where "equal on all tested inputs" is checked one pair at a time
instead of through the sums of an alignment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

f32 = np.float32
ELLS = (0.15, 0.10, 0.06, 0.03)


def _device_d2(y):
    d2 = y[:, 0] * y[:, 0]; d2 = d2 + y[:, 1] * y[:, 1]; d2 = d2 + y[:, 2] * y[:, 2]          # nanoflann.hpp:403-406
    return d2


def _device_d2c(g):
    t = g * g
    return (t[:, 0] + t[:, 1]) + (t[:, 2] + (t[:, 3] + t[:, 4]))


def _gates(oracle, ell):
    p = oracle.default_params()
    s2 = f32(p.sigma) * f32(p.sigma)
    d2_thres = f32(-2.0 * float(f32(ell)) * float(f32(ell)) * float(np.log(f32(f32(p.sp_thres) / s2))))
    d2c_thres = f32(-2.0 * float(p.c_ell) * float(p.c_ell) * float(np.log(f32(f32(f32(p.sp_thres) / f32(p.c_sigma)) / f32(p.c_sigma)))))
    return p, s2, d2_thres, d2c_thres


def _split(rng, total, parts):
    """`parts` non-negative float32 squares-to-be that add up (roughly) to `total`: returns their square roots."""
    w = rng.random((total.shape[0], parts)).astype(np.float64) + 1e-3
    w = w / w.sum(axis=1, keepdims=True)
    return np.sqrt(w * total[:, None].astype(np.float64)).astype(f32)


def _near(centre, ulps):
    """float32 neighbours of `centre`: centre + k ulp for k in -ulps .. ulps."""
    out = [f32(centre)]
    lo = f32(centre); hi = f32(centre)
    for _ in range(ulps):
        lo = np.nextafter(lo, f32(-np.inf), dtype=f32); hi = np.nextafter(hi, f32(np.inf), dtype=f32)
        out += [lo, hi]
    return np.array(sorted(out), f32)


def _hit_d2(rng, target, tries=4000):
    """A point y with the device's d2(y) == target exactly (two components; found by search), or None."""
    t = float(target)
    e0 = (np.sqrt(t) * rng.uniform(0.3, 0.9, tries)).astype(f32)
    rest = np.maximum(t - (e0 * e0).astype(np.float64), 0.0)
    e1 = np.sqrt(rest).astype(f32)
    for dk in (0, 1, -1, 2, -2):
        e1k = e1.copy()
        for _ in range(abs(dk)):
            e1k = np.nextafter(e1k, f32(np.inf if dk > 0 else -np.inf), dtype=f32)
        d2 = e0 * e0 + e1k * e1k
        hit = np.nonzero(d2 == f32(target))[0]
        if hit.size:
            return np.array([e0[hit[0]], e1k[hit[0]], 0.0], f32)
    return None


@pytest.mark.parametrize("ell", ELLS)
def test_pair_values_are_bit_equal_to_the_oracle(hiplib, oracle, ell):
    from cvo_slam_amd import api as ca
    p, s2, d2_thres, d2c_thres = _gates(oracle, ell)
    rng = np.random.default_rng(20240 + int(ell * 1000))
    blocks = []
    # (1) pairs all over the neighbourhood and a little beyond it; colour distances over the range 8-bit colours + gradients can have, and beyond the gate
    n = 2_400_000
    d2_t = (rng.random(n) ** 1.5 * 1.3 * float(d2_thres)).astype(f32)
    d2c_t = np.where(rng.random(n) < 0.9, rng.random(n) ** 2 * 60000.0, rng.random(n) * 1.15 * float(d2c_thres)).astype(f32)
    blocks.append(np.concatenate([_split(rng, d2_t, 3) * rng.choice([-1, 1], (n, 3)).astype(f32), _split(rng, d2c_t, 5)], axis=1))
    # (2) the membership boundary a == sp_thres: for a colour distance, the geometric distance at which ck * k crosses sp_thres, and its neighbourhood
    m = 150_000
    d2c_b = (rng.random(m) * 17000.0).astype(f32)                                              # beyond 2 c_ell^2 ln(s2/sp) = 17 851 no distance is a member
    ck = np.exp(-d2c_b.astype(np.float64) / (2.0 * float(p.c_ell) ** 2))
    d2_b = 2.0 * float(f32(ell)) ** 2 * np.log(float(s2) * ck / float(f32(p.sp_thres)))
    d2_b = np.clip(d2_b * (1.0 + rng.normal(0, 2e-7, m)), 0, None).astype(f32)                  # within a few float ulps of the crossing
    r = np.sqrt(d2_b.astype(np.float64)).astype(f32)
    yb = np.zeros((m, 3), f32); yb[:, 0] = r
    gb = np.zeros((m, 5), f32); gb[:, 0] = np.sqrt(d2c_b.astype(np.float64)).astype(f32)
    blocks.append(np.concatenate([yb, gb], axis=1))
    # (3) the radius gate: d2 == d2_thres +- k ulp, exactly (points found by search), over a range of colour distances
    hit = [y for y in (_hit_d2(rng, t) for t in _near(d2_thres, 4)) if y is not None]
    assert len(hit) >= 5, "the search finds points for most of the nine neighbours of d2_thres"
    yk = np.repeat(np.array(hit, f32), 200, axis=0)
    gk = np.zeros((yk.shape[0], 5), f32); gk[:, 1] = np.tile(np.linspace(0, 130, 200, dtype=f32), len(hit))
    blocks.append(np.concatenate([yk, gk], axis=1))
    # (4) the colour gate: d2c == d2_c_thres +- k ulp (single-component squares that land there), small distances
    cands = _near(f32(np.sqrt(float(d2c_thres))), 40)
    gc = np.zeros((cands.size, 5), f32); gc[:, 2] = cands
    yc = np.zeros((cands.size, 3), f32); yc[:, 1] = f32(0.2 * np.sqrt(float(d2_thres)))
    blocks.append(np.concatenate([yc, gc], axis=1))
    cases = np.ascontiguousarray(np.concatenate(blocks, axis=0), f32)

    a_dev, d2_dev, d2c_dev = ca.selftest_pair_values(cases, ell)
    d2 = _device_d2(cases[:, :3]); d2c = _device_d2c(cases[:, 3:])
    assert np.array_equal(d2_dev, d2) and np.array_equal(d2c_dev, d2c), "the distances are formed with the association the tests assume"
    a_ref, k_ref, ck_ref = oracle.pair_values(d2, d2c, ell)
    members = int(np.count_nonzero(a_ref))
    assert members > 0.2 * cases.shape[0] and members < 0.9 * cases.shape[0]                   # both outcomes are well represented
    near_sp = np.abs(ck_ref.astype(np.float64) * k_ref - float(f32(p.sp_thres))) <= 4 * float(np.spacing(f32(p.sp_thres)))
    assert int(near_sp.sum()) > 1000, "thousands of cases sit within 4 ulps of a == sp_thres"
    assert int((np.abs(d2.astype(np.float64) - float(d2_thres)) <= 4 * float(np.spacing(d2_thres))).sum()) >= 1000
    assert int((np.abs(d2c.astype(np.float64) - float(d2c_thres)) <= 4 * float(np.spacing(d2c_thres))).sum()) >= 2
    for route, name in enumerate(("se_kernel_value", "colour_factors + se_kernel_value_ck", "12-term chain", "degree 7 + guard")):
        bad = np.nonzero(a_dev[:, route].view(np.uint32) != a_ref.view(np.uint32))[0]
        assert bad.size == 0, (f"{name}: {bad.size} of {cases.shape[0]} pairs differ from the oracle at ell = {ell}; first: d2 = {d2[bad[0]]!r}, d2c = {d2c[bad[0]]!r}, "
                               f"device {a_dev[bad[0], route]!r}, oracle {a_ref[bad[0]]!r}")


def test_pair_values_with_other_parameters(hiplib, oracle):
    """Parameters for which the exponent leaves the polynomial's range (Gates::poly_ok false): the list routes fall back to the library exp,
    the branch-free routes are not used (they report 0)."""
    from cvo_slam_amd import api as ca
    p = oracle.default_params(); p.sigma = 0.3; p.sp_thres = 2e-3; p.c_ell = 120.0
    cp = ca.default_params(); cp.sigma = 0.3; cp.sp_thres = 2e-3; cp.c_ell = 120.0
    rng = np.random.default_rng(7)
    n = 400_000
    ell = 0.08
    d2_thres = -2.0 * ell * ell * np.log(2e-3 / 0.09)
    d2_t = (rng.random(n) * 1.2 * d2_thres).astype(f32)
    d2c_t = (rng.random(n) ** 2 * 120000.0).astype(f32)
    cases = np.ascontiguousarray(np.concatenate([_split(rng, d2_t, 3), _split(rng, d2c_t, 5)], axis=1), f32)
    a_dev, d2_dev, d2c_dev = ca.selftest_pair_values(cases, ell, params=cp)
    a_ref, _, _ = oracle.pair_values(d2_dev, d2c_dev, ell, params=p)
    assert 0.1 * n < np.count_nonzero(a_ref) < 0.95 * n
    for route in (0, 1):
        assert np.array_equal(a_dev[:, route].view(np.uint32), a_ref.view(np.uint32)), route
    assert not a_dev[:, 2].any() and not a_dev[:, 3].any()


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_device_float_routines(hiplib, oracle):
    """The float routines of the scalar epilogue and the gates, element by element on the arguments the kernel produces.

    Exp_SEK3's sine and cosine (LieGroup.cpp:174-175; dt * theta with step <= 0.8 and |omega| from the stop threshold 5e-5 up to a few tenths, and
    larger arguments for the other branch): the reference calls its libm's float routines, whose last bit differs between libms, so oracle and device
    both take the correctly rounded float (cvo_math.hpp: sin_f32_cr, cos_f32_cr; oracle: the double routine rounded once) -- they must agree bit for
    bit.  What that choice replaces is measured beside it: glibc's sinf / cosf and OCML's against the correctly rounded value.  The logarithm of the gates
    (cvo.cpp:125-126, 395-396) likewise: log_f32_cr on the device against the oracle's, bit for bit; OCML's and glibc's logf beside it (the three gate arguments of the
    default parameters give the same floats in all three)."""
    from cvo_slam_amd import api as ca
    rng = np.random.default_rng(11)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-7), np.log(0.5), 2_000_000)), rng.uniform(0, 0.5, 1_000_000), rng.uniform(0.5, 6.5, 200_000),
                        -rng.uniform(0, 1.0, 100_000)]).astype(f32)
    dev = ca.selftest_libm(x)
    report = {}
    xl = np.concatenate([rng.uniform(1e-4, 1.0, 500_000), np.exp(rng.uniform(np.log(1e-6), np.log(10.0), 500_000))]).astype(f32)   # arguments of the gates' logarithm: sp / s2 and the like
    devl6 = ca.selftest_libm(xl)
    for col, kind, arg, out in ((3, "sin_cr", x, dev), (4, "cos_cr", x, dev), (5, "log_cr", xl, devl6)):
        ref = oracle.libm_f32(kind, arg)
        bad = np.nonzero(out[:, col].view(np.uint32) != ref.view(np.uint32))[0]
        assert bad.size == 0, f"{kind}: {bad.size} of {arg.size} arguments differ; first x = {arg[bad[0]]!r}: device {out[bad[0], col]!r}, oracle {ref[bad[0]]!r}"
    small = np.abs(x) < 1e-3                                                            # where a converging alignment's dt * theta lives
    for col, kind in ((0, "sin"), (1, "cos")):
        cr = oracle.libm_f32(kind + "_cr", x); gl = oracle.libm_f32(kind, x)
        report[kind] = dict(glibc_vs_cr=int((gl != cr).sum()), ocml_vs_cr=int((dev[:, col] != cr).sum()), ocml_vs_glibc=int((dev[:, col] != gl).sum()),
                            glibc_vs_cr_below_1e3=int(((gl != cr) & small).sum()), ocml_vs_cr_below_1e3=int(((dev[:, col] != cr) & small).sum()))
        assert _ulps(gl, cr).max() <= 1 and _ulps(dev[:, col], cr).max() <= 2, report       # the libms are good to an ulp or two -- and not to the last bit
        assert report[kind]["glibc_vs_cr_below_1e3"] <= 1e-5 * small.sum()                  # (rare where alignments live: why the bench's poses never showed it)
    print("float routines against the correctly rounded value, of", x.size, "arguments:", report)
    p = oracle.default_params()
    args = np.array([f32(p.sp_thres) / (f32(p.sigma) * f32(p.sigma)), f32(p.sp_thres) / f32(p.sigma) / f32(p.sigma),
                     f32(p.sp_thres) / f32(p.c_sigma) / f32(p.c_sigma)], f32)
    more = np.concatenate([args, rng.uniform(1e-4, 1.0, 500_000).astype(f32)])
    d6 = ca.selftest_libm(more); devl = d6[:, 2]
    ref = oracle.libm_f32("log", more); cr = oracle.libm_f32("log_cr", more)
    # the gates of the default parameters: glibc's logf, OCML's and the correctly rounded value (what device, host and oracle use) are the same floats
    assert np.array_equal(devl[:3].view(np.uint32), ref[:3].view(np.uint32)) and np.array_equal(cr[:3].view(np.uint32), ref[:3].view(np.uint32)), (devl[:3], ref[:3], cr[:3])
    assert np.array_equal(d6[:, 5].view(np.uint32), cr.view(np.uint32))
    assert _ulps(devl, ref).max() <= 2 and _ulps(ref, cr).max() <= 1
    print("logf: OCML against glibc", int((devl != ref).sum()), "of", more.size, "arguments differ (by at most", int(_ulps(devl, ref).max()), "ulp); glibc against the correctly rounded value",
          int((ref != cr).sum()))
