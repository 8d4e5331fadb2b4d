"""The oracle's model builders (oracle/mjo_core.c) against what the reference's MJCF templates SAY
(tests/golden/mjcf_tables.json, rendered with jinja2 + parsed by tests/golden/gen_mjcf_tables.py): joint
anchors / axes / ranges / armature / damping / stiffness, capsule end points and radii, frictions, condim,
margins, motors, options -- element by element, in world coordinates at qpos0."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle_bindings import KINDS, _p, lib

CASES = [("hopper", "hopper", None), ("walker2d", "walker2d", None), ("walker2d_alt", "walker2d", [.5, .3, .7, .25]),
         ("halfcheetah", "halfcheetah", None), ("humanoid", "humanoid", None)]
JT = {"free": 0, "slide": 2, "hinge": 3}


def _dump(kind, size):
    jnt = np.zeros(14 * 24); geom = np.zeros(12 * 24); act = np.zeros(4 * 20); opt = np.zeros(32); dims = np.zeros(5, dtype=np.int32)
    s = None if size is None else np.ascontiguousarray(size, dtype=np.float64)
    rc = lib().mjo_model_dump(KINDS[kind], _p(s), _p(jnt), _p(geom), _p(act), _p(opt), dims.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return jnt.reshape(-1, 14)[:dims[1]], geom.reshape(-1, 12)[:dims[2]], act.reshape(-1, 4)[:dims[3]], opt, dims


@pytest.mark.parametrize("key,kind,size", CASES)
def test_oracle_model_equals_rendered_mjcf(golden_dir, key, kind, size):
    t = json.load(open(os.path.join(golden_dir, "mjcf_tables.json")))[key]
    jnt, geom, act, opt, dims = _dump(kind, size)
    assert dims[0] == len(t["bodies"]) + 1 and dims[1] == len(t["joints"]) and dims[2] == len(t["geoms"]) and dims[3] == len(t["motors"])
    names = [j["name"] for j in t["joints"]]
    for r, j in zip(jnt, t["joints"]):
        assert int(r[0]) == JT[j["type"]], j["name"]
        if j["type"] != "slide":   # the anchor of a slide joint does not enter the kinematics
            assert np.allclose(r[1:4], j["pos"], atol=1e-12), (j["name"], r[1:4], j["pos"])
        if j["type"] != "free":
            assert np.allclose(r[4:7], j["axis"], atol=1e-12), j["name"]
            assert bool(r[7]) == j["limited"] and (not j["limited"] or np.allclose(r[8:10], j["range"], atol=1e-12)), j["name"]
            assert r[13] == j["ref"]
        assert r[10] == j["armature"] and r[11] == j["damping"] and r[12] == j["stiffness"], j["name"]
    floor_fr = [g for g in t["geoms"] if g["type"] == "plane"][0]
    for r, g in zip(geom, t["geoms"]):
        if g["type"] == "plane":
            assert int(r[0]) == 0 and int(r[9]) == g["condim"]
            continue
        assert r[7] == g["radius"] and int(r[9]) == g["condim"] and r[10] == g["margin"], g["name"]
        if g["friction"] is not None:
            assert r[8] == g["friction"][0], g["name"]
        if g["type"] == "capsule":
            ends = sorted([tuple(np.round(r[1:4], 10)), tuple(np.round(r[4:7], 10))])
            want = sorted([tuple(np.round(g["p0"], 10)), tuple(np.round(g["p1"], 10))])
            assert np.allclose(ends, want, atol=1e-10), (g["name"], ends, want)
        else:
            assert int(r[0]) == 2 and np.allclose(r[1:4], g["center"], atol=1e-12), g["name"]
    for r, mtr in zip(act, t["motors"]):
        assert names[int(r[0])] == mtr["joint"] and r[1] == mtr["gear"] and list(r[2:4]) == mtr["ctrlrange"]
    assert opt[0] == float(t["option"]["timestep"])
    assert int(opt[1]) == (1 if t["option"].get("integrator") == "RK4" else 0)
    assert int(opt[2]) == (0 if t["option"].get("solver") == "PGS" else 2)
    if "iterations" in t["option"]:
        assert int(opt[3]) == int(t["option"]["iterations"])
    assert t["flags"].get("warmstart") == "disable"          # every XML disables warmstart: the solver starts at qacc_smooth
    for k, p in enumerate(t["pairs"]):                        # explicit <pair>s come first in the compiled list
        assert int(opt[5 + 3 * k]) == 1 and opt[6 + 3 * k] == p["friction"][0] and int(opt[7 + 3 * k]) == p["condim"]
    if t["settotalmass"]:
        from oracle_bindings import oracle_constants
        assert abs(oracle_constants(kind)["body_mass"].sum() - t["settotalmass"]) < 1e-12
