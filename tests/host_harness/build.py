"""Builds and binds the host instantiation of the kernel math (TEST HARNESS ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "_build_planar_host.so")
SRC = os.path.join(HERE, "planar_host.cpp")
DEPS = [SRC] + [os.path.join(ROOT, "random-envs_amd", "csrc", f) for f in
                ("planar_spec.hpp", "planar_engine.hpp", "planar_model.hpp")]
_D = ctypes.POINTER(ctypes.c_double)
_lib = None
KINDS = {"hopper": 1, "halfcheetah": 2, "walker2d": 3}


def lib():
    global _lib
    if _lib is None:
        stale = (not os.path.exists(SO)) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS)
        if stale:
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO, SRC])
        try:
            _lib = ctypes.CDLL(SO)
        except OSError:
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO, SRC])
            _lib = ctypes.CDLL(SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(_D) if a is not None else None


def host_step(kind, f32, qpos, qvel, action, xi, nsub):
    L = lib()
    q = np.ascontiguousarray(np.asarray(qpos, dtype=np.float64).T); v = np.ascontiguousarray(np.asarray(qvel, dtype=np.float64).T)
    a = np.ascontiguousarray(np.asarray(action, dtype=np.float64).T); x = np.ascontiguousarray(np.asarray(xi, dtype=np.float64).T)
    n = q.shape[1]
    qo = np.zeros_like(q); vo = np.zeros_like(v); cap = np.zeros(n, dtype=np.int32)
    rc = L.ph_step(KINDS[kind], int(f32), n, nsub, _p(q), _p(v), _p(a), _p(x), None, _p(qo), _p(vo),
                   cap.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return qo.T.copy(), vo.T.copy(), cap


def set_fast(flag):
    """SolParams.fast of the following calls: 1 = the feet-only straight-line solver instantiation is allowed (default),
    0 = general instantiation only."""
    lib().ph_set_fast(int(flag))


def set_rolled(flag):
    """1 = the general path as the rolled row-list solver (planar_engine.hpp::solve_newton_rolled: what the 256-register hopper kernel of
    handles with >= 65 536 envs runs), 0 = the unrolled per-slot instantiation"""
    lib().ph_set_rolled(int(flag))


def set_line_search(ls_max=-1, ls_free=-1):
    """override SolParams.ls_max / ls_free of the following calls (-1: the model's defaults)"""
    lib().ph_set_ls(int(ls_max), int(ls_free))


def last_mode():
    """solver instantiation the last host_forward entered: 0 no rows, 1 general, 2 general + self rows, 3 feet-only"""
    return lib().ph_last_mode()


def host_forward(kind, f32, qpos, qvel, action, xi):
    L = lib(); nv = len(qvel)
    q = np.ascontiguousarray(qpos, dtype=np.float64); v = np.ascontiguousarray(qvel, dtype=np.float64)
    a = np.ascontiguousarray(action, dtype=np.float64); x = np.ascontiguousarray(xi, dtype=np.float64)
    qacc = np.zeros(nv); M = np.zeros((nv, nv)); it = ctypes.c_int()
    rc = L.ph_forward(KINDS[kind], int(f32), _p(q), _p(v), _p(a), _p(x), None, _p(qacc), _p(M), ctypes.byref(it))
    assert rc == 0
    return qacc, M, it.value


def host_constants(kind, f32=False, size=None):
    L = lib(); nb = {"hopper": 4, "halfcheetah": 7, "walker2d": 7}[kind]
    mass = np.zeros(nb); iyy = np.zeros(nb); tran = np.zeros(nb); dofw = np.zeros(nb); sol = np.zeros(5)
    s = None if size is None else np.ascontiguousarray(size, dtype=np.float64)
    rc = L.ph_constants(KINDS[kind], int(f32), _p(s), _p(mass), _p(iyy), _p(tran), _p(dofw), _p(sol))
    assert rc == 0
    return dict(mass=mass, iyy=iyy, tran_invw=tran, dof_invw=dofw, con_K=sol[0], con_B=sol[1], lim_K=sol[2],
                lim_B=sol[3], meaninertia=sol[4])
