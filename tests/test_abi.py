"""The C-ABI library loads and exports every symbol include/rex.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "rex.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rex_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    from random_envs_amd import _native
    assert sorted(_native.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from random_envs_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    assert sorted(_native.exported_symbols()) == _declared()


def test_no_cpu_fallback_in_product():
    """the product package never imports the oracle or the host harness"""
    pkg = os.path.join(ROOT, "random-envs_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "oracle" not in src.replace("oracle/", "").lower() or f == "__init__.py" and False, f
            assert "host_harness" not in src and "libmjo" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from random_envs_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.RexError, match="no CPU fallback"):
        _native.lib()


# ------------------------------------------------------------------------------------------------- source-level contracts of the C-ABI
_SRC = os.path.join(ROOT, "random-envs_amd", "csrc", "rex_hip.hip")
_DEVICE_WORK = re.compile(r"hipLaunchKernelGGL|hipMemcpy|hipMemset|hipDeviceSynchronize|hipEvent(Record|Create|Synchronize)|hipMalloc|hipFree|"
                          r"\bcopy_rows\(|\bdo_reset\(|\blaunch_[a-z_]+(<[A-Za-z0-9_]+>)?\(|\bensure_replay_scratch\(")


def _exported_bodies():
    """name -> body of every extern "C" function defined in rex_hip.hip (brace matching; diagnostic-only rex_debug_* excluded)."""
    src = open(_SRC).read()
    out = {}
    for m in re.finditer(r'extern "C"\s+[^;{(]*?\b(rex_[a-z_0-9]+)\s*\(([^)]*)\)\s*\{', src):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0); i += 1
        if not m.group(1).startswith("rex_debug_"):
            out[m.group(1)] = (m.group(2), src[m.end():i])
    return out


def test_every_entry_point_that_touches_the_device_selects_the_handles_device_first():
    """One process may drive one handle per GPU from one thread (SURVEY 8(b) "Threading"): every exported function that takes a handle and
    launches, copies, allocates or synchronises must make the handle's device current BEFORE it does (REX_ENTER, or hipSetDevice(h->device)
    ahead of the first such call).  Round 3 shipped half of them without it -- invisible on a one-GPU box."""
    bodies = _exported_bodies()
    assert len(bodies) >= 30
    missing = []
    for name, (args, body) in bodies.items():
        if "rex_t*" not in args.replace("rex_t *", "rex_t*") or "rex_t**" in args.replace(" ", ""):
            continue   # no handle argument (rex_get_dims, rex_last_error ...) / rex_create (selects device_id itself)
        work = _DEVICE_WORK.search(body)
        if not work:
            continue   # pure host bookkeeping (rex_set_flags, rex_step_count ...)
        sel = re.search(r"REX_ENTER\(|hipSetDevice\(\s*\(?h\)?->device", body)
        if not sel or sel.start() > work.start():
            missing.append(name)
    assert not missing, "entry points that touch the device before selecting it: %s" % missing
    create = open(_SRC).read()
    assert re.search(r"HIP_TRY\(hipSetDevice\(device_id\)\);\s*rex_env\* h = new", create), "rex_create selects device_id before allocating"


def test_the_product_reads_no_environment_variable_outside_the_gated_knobs():
    """A stray variable must not change what a production process computes: every getenv of the library sits in the three knob helpers
    (honoured only beside REX_ALLOW_TUNING=1, refused by rex_create otherwise), the physics-changing diagnostics are compiled in by -DREX_TUNING
    only, and build() never defines it."""
    src = open(_SRC).read()
    for inc in ("planar_engine.hpp", "planar_model.hpp", "planar_spec.hpp", "humanoid_engine.hpp", "humanoid_pair.hpp", "humanoid_model.hpp"):
        assert "getenv" not in open(os.path.join(ROOT, "random-envs_amd", "csrc", inc)).read(), inc
    lines = [l for l in src.splitlines() if "getenv(" in l and not l.lstrip().startswith("//")]
    allowed = ('getenv("REX_ALLOW_TUNING")', "if (getenv(k)) return k", "getenv(name)", 'getenv("REX_DIAG_NOCONTACT") || getenv("REX_HUM_ITERS")')
    stray = [l.strip() for l in lines if not any(a in l for a in allowed)]
    assert not stray, stray
    for name in ("REX_DIAG_NOCONTACT", "REX_HUM_ITERS"):   # physics-changing: only under #if defined(REX_TUNING)
        for m in re.finditer(r'knob\("%s"\)' % name, src):
            before = src[:m.start()]
            assert before.rfind("#if defined(REX_TUNING)") > before.rfind("#endif"), name
    import __graft_entry__ as g
    assert not any("REX_TUNING" in f for f in g.HIPCC_FLAGS)


def test_rex_create_has_one_cleanup_path():
    """Every failure after the handle exists goes through rex_destroy (which frees whatever was allocated): create_body only RETURNS error
    codes, rex_create destroys the handle on any of them, and the humanoid's process-wide model is a thread-safe magic static."""
    src = open(_SRC).read()
    body = src[src.index("static int create_body("):src.index('extern "C" int rex_create(')]
    assert "delete h" not in body and "new (std::nothrow) rex_env" not in body
    create = _exported_bodies()["rex_create"][1]
    assert "create_body(" in create and "rex_destroy(h)" in create and "*out = nullptr" in create
    assert "static bool built" not in src and "static const HumModels* m = [] {" in src
