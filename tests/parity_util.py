"""Per-lane parity gates shared by the GPU parity tests.

A percentile gate lets the rare-path lanes (self-collision instantiation, the larger humanoid sweep sizes, capped line
searches) be arbitrarily wrong.  Here EVERY lane has to satisfy
    err <= tol                      the stated fp32 tolerance, or
    err <= K * sens  and  err <= cap
where `sens` is how far the fp64 oracle's OWN output moves when its inputs move by fp32 rounding (oracle_bindings.
oracle_sensitivity): the only accepted explanation for an error above the tolerance is that the oracle itself is that
ill-conditioned in this lane (an active-set switch -- contact margin, joint limit, friction-cone edge, the PGS cap -- within
rounding of the inputs).  `cap` bounds even the explained lanes.  Unexplained outliers fail, with their lane indices."""
import contextlib
import os

import numpy as np


def assert_lanes_explained(err, sens, tol, cap, K=64.0, label=""):
    err = np.asarray(err, dtype=np.float64); sens = np.asarray(sens, dtype=np.float64)
    assert np.isfinite(err).all(), "%s: non-finite error in lanes %s" % (label, np.where(~np.isfinite(err))[0][:10])
    over = err > tol
    explained = err <= K * sens
    bad = np.where(over & ~explained)[0]
    too_big = np.where(err > cap)[0]
    msg = ("%s: max %.3e p99 %.3e median %.3e | %d of %d lanes above tol %.1e, %d unexplained %s, %d above the cap %.1e %s"
           % (label, err.max(), np.percentile(err, 99), np.median(err), int(over.sum()), err.size, tol, bad.size,
              [(int(i), float("%.2e" % err[i]), float("%.2e" % sens[i])) for i in bad[:8]], too_big.size, cap,
              [(int(i), float("%.2e" % err[i])) for i in too_big[:8]]))
    print(msg)
    assert bad.size == 0 and too_big.size == 0, msg
    return int(over.sum())


def assert_done_explained(done_gpu, done_ref, margins, eps, label=""):
    """`done` may differ from the oracle only where the oracle's own state sits within `eps` of a termination threshold
    (`margins`: per lane, the smallest distance of the reference state to any threshold of the done rule)."""
    mism = np.where(np.asarray(done_gpu, bool) != np.asarray(done_ref, bool))[0]
    bad = [int(i) for i in mism if not (margins[i] <= eps)]
    print("%s: %d done mismatches, margins %s" % (label, mism.size, [float("%.2e" % margins[i]) for i in mism[:8]]))
    assert not bad, "%s: done differs away from every threshold in lanes %s (margins %s)" % (
        label, bad[:10], [float(margins[i]) for i in bad[:10]])


@contextlib.contextmanager
def lanes_per_block(n):
    """Launch shape of the handles created inside: REX_LANES is read once in rex_create (None = the default rule:
    32-lane blocks up to 32 768 envs, 64-lane blocks -- and the > 64 KB dynamic-LDS opt-in of the humanoid's one-lane kernels -- past that)."""
    with create_knobs(REX_LANES=n):
        yield


@contextlib.contextmanager
def create_knobs(**knobs):
    """Environment knobs librex reads once in rex_create (REX_FAST, REX_LANES, ...) for the handles created inside.  The library honours a
    knob only beside REX_ALLOW_TUNING=1 and refuses to create a handle otherwise (include/rex.h), so that rides along while one is set."""
    if any(v is not None for v in knobs.values()):
        knobs = dict(knobs, REX_ALLOW_TUNING=1)
    old = {k: os.environ.get(k) for k in knobs}
    try:
        for k, v in knobs.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
