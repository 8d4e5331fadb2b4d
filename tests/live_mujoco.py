"""Optional LIVE cross-check against a stock MuJoCo (third party, NOT reference code), the one route from "parity
unpinned" to a numeric pin of the MuJoCo-backed rows (SURVEY.md section 8c "GPU box").

Nothing here reads the reference: the MJCF below is emitted from the build's OWN model description --
tests/golden/mjcf_tables.json (what the four templates say, as data; the oracle's compiled models are checked against it
element by element in tests/test_model_vs_mjcf.py) plus the oracle's compiled inertial constants
(oracle_bindings.oracle_constants: MuJoCo 2.1.0's capsule volume pi r^2 (L + r)), written as explicit
<inertial> elements so that any MuJoCo version integrates the SAME rigid bodies.  Body frames are world-aligned at qpos0
(the tables are in world coordinates), which changes no dynamics.

Importable without MuJoCo (the emitter is plain string building and is unit-tested on the CPU); everything that needs
`mujoco` / `mujoco_py` imports it lazily and the tests skip when neither is installed."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND_TABLE = {"hopper": "hopper", "walker2d": "walker2d", "halfcheetah": "halfcheetah", "humanoid": "humanoid"}


def tables(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", "mjcf_tables.json")))[name]


def _f(v):
    return " ".join(repr(float(x)) for x in np.atleast_1d(v))


def emit_mjcf(kind, consts, table=None):
    """MJCF (local coordinates, radians) of `kind` from the tables + the oracle's inertial constants `consts`."""
    t = tables(KIND_TABLE[kind]) if table is None else table
    bodies = {b["name"]: b for b in t["bodies"]}
    order = [b["name"] for b in t["bodies"]]
    wpos = {"world": np.zeros(3)}
    for b in t["bodies"]:
        wpos[b["name"]] = np.array(b["pos"], dtype=float)
    children = {}
    for b in t["bodies"]:
        children.setdefault(b["parent"], []).append(b["name"])
    opt = dict(t["option"])
    flags = dict(t["flags"])
    out = ['<mujoco model="rex_%s">' % kind,
           '  <compiler angle="radian" inertiafromgeom="false"/>',
           '  <option %s>' % " ".join('%s="%s"' % kv for kv in sorted(opt.items())),
           '    <flag %s/>' % " ".join('%s="%s"' % kv for kv in sorted(flags.items())) if flags else '',
           '  </option>',
           '  <worldbody>']

    def geom_xml(g, origin, indent):
        a = ['name="%s"' % g["name"], 'type="%s"' % g["type"], 'contype="%d"' % g["contype"], 'conaffinity="%d"' % g["conaffinity"],
             'condim="%d"' % g["condim"], 'margin="%s"' % repr(float(g["margin"]))]
        if g.get("friction"):
            fr = list(g["friction"]) + [0.005, 0.0001][len(g["friction"]) - 1:] if len(g["friction"]) < 3 else g["friction"]
            a.append('friction="%s"' % _f(fr))
        if g.get("solimp"):
            a.append('solimp="%s"' % _f(g["solimp"]))
        if g.get("solref"):
            a.append('solref="%s"' % _f(g["solref"]))
        if g["type"] == "plane":
            a += ['pos="0 0 0"', 'size="%s"' % _f(g["size"])]
        elif g["type"] == "sphere":
            a += ['pos="%s"' % _f(np.array(g["center"]) - origin), 'size="%s"' % repr(float(g["radius"]))]
        else:
            a += ['fromto="%s %s"' % (_f(np.array(g["p0"]) - origin), _f(np.array(g["p1"]) - origin)), 'size="%s"' % repr(float(g["radius"]))]
        return indent + "<geom %s/>" % " ".join(a)

    def joint_xml(j, origin, indent):
        if j["type"] == "free":
            return indent + '<joint name="%s" type="free" armature="%s" damping="%s"/>' % (j["name"], repr(float(j["armature"])), repr(float(j["damping"])))
        a = ['name="%s"' % j["name"], 'type="%s"' % j["type"], 'pos="%s"' % _f(np.array(j["pos"]) - origin), 'axis="%s"' % _f(j["axis"]),
             'armature="%s"' % repr(float(j["armature"])), 'damping="%s"' % repr(float(j["damping"])), 'stiffness="%s"' % repr(float(j["stiffness"])),
             'ref="%s"' % repr(float(j["ref"])), 'limited="%s"' % ("true" if j["limited"] else "false")]
        if j["limited"]:
            a.append('range="%s"' % _f(j["range"]))
        if j.get("solimplimit"):
            a.append('solimplimit="%s"' % _f(j["solimplimit"]))
        return indent + "<joint %s/>" % " ".join(a)

    for g in t["geoms"]:
        if g["body"] == "world":
            out.append(geom_xml(g, np.zeros(3), "    "))

    def body_xml(name, indent):
        b = bodies[name]; i = 1 + order.index(name); origin = wpos[name]
        out.append('%s<body name="%s" pos="%s">' % (indent, name, _f(origin - wpos[b["parent"]])))
        I = np.array(consts["body_inertia"][i]).reshape(3, 3)
        out.append('%s  <inertial pos="%s" mass="%s" fullinertia="%s"/>' % (
            indent, _f(consts["body_ipos"][i]), repr(float(consts["body_mass"][i])), _f([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])))
        for j in t["joints"]:
            if j["body"] == name:
                out.append(joint_xml(j, origin, indent + "  "))
        for g in t["geoms"]:
            if g["body"] == name:
                out.append(geom_xml(g, origin, indent + "  "))
        for c in children.get(name, []):
            body_xml(c, indent + "  ")
        out.append("%s</body>" % indent)

    for c in children.get("world", []):
        body_xml(c, "    ")
    out.append("  </worldbody>")
    if t["pairs"]:
        out.append("  <contact>")
        for pr in t["pairs"]:
            a = ['geom1="%s"' % pr["geom1"], 'geom2="%s"' % pr["geom2"], 'condim="%d"' % pr["condim"], 'friction="%s"' % _f(pr["friction"])]
            if pr.get("solimp"):
                a.append('solimp="%s"' % _f(pr["solimp"]))
            out.append("    <pair %s/>" % " ".join(a))
        out.append("  </contact>")
    out.append("  <actuator>")
    for m in t["motors"]:
        out.append('    <motor joint="%s" gear="%s" ctrllimited="true" ctrlrange="%s"/>' % (m["joint"], repr(float(m["gear"])), _f(m["ctrlrange"])))
    out.append("  </actuator>")
    out.append("</mujoco>")
    return "\n".join(x for x in out if x)


class LiveSim:
    """A stock-MuJoCo sim of the build's own model: `mujoco` (>= 2.2 bindings) or `mujoco_py` (2.1)."""

    def __init__(self, kind, consts):
        self.kind = kind
        xml = emit_mjcf(kind, consts)
        try:
            import mujoco
            self.api = "mujoco"; self.mj = mujoco
            self.model = mujoco.MjModel.from_xml_string(xml); self.data = mujoco.MjData(self.model)
            self.version = mujoco.__version__
        except ImportError:
            import mujoco_py
            self.api = "mujoco_py"; self.mj = mujoco_py
            self.model = mujoco_py.load_model_from_xml(xml); self.sim = mujoco_py.MjSim(self.model); self.data = self.sim.data
            self.version = "mujoco_py"
        # regularisation scales the solver reads at run time: the oracle's restatement of mj_setConst must reproduce what
        # MuJoCo itself derives from the same inertials (checked by the caller), after which they are left as compiled
        self.compiled = dict(body_mass=np.array(self.model.body_mass), body_invweight0=np.array(self.model.body_invweight0),
                             dof_invweight0=np.array(self.model.dof_invweight0), body_subtreemass=np.array(self.model.body_subtreemass))

    def set_task(self, kind, xi):
        """The reference's set_task writes (body_mass / pair_friction / dof_damping only: SURVEY Q4, Q13)."""
        m = self.model
        if kind == "hopper":
            m.body_mass[1:] = xi
        elif kind == "halfcheetah":
            m.body_mass[1:] = xi[:7]; m.pair_friction[0:2, 0:2] = xi[7]
        elif kind == "walker2d":
            m.body_mass[1:] = xi[:7]; m.pair_friction[0, 0:2] = xi[11]; m.pair_friction[1, 0:2] = xi[12]
        elif kind == "humanoid":
            m.body_mass[1:] = xi[:13]; m.dof_damping[6:] = xi[13:]

    def step(self, qpos, qvel, ctrl, frame_skip):
        d = self.data
        if self.api == "mujoco":
            self.mj.mj_resetData(self.model, d)
            d.qpos[:] = qpos; d.qvel[:] = qvel; d.ctrl[:] = ctrl
            self.mj.mj_forward(self.model, d)              # set_state -> sim.forward()
            for _ in range(frame_skip):
                self.mj.mj_step(self.model, d)
        else:
            self.sim.reset()
            st = self.sim.get_state()
            self.sim.set_state(self.mj.MjSimState(st.time, np.asarray(qpos), np.asarray(qvel), st.act, st.udd_state))
            self.sim.forward(); d.ctrl[:] = ctrl
            for _ in range(frame_skip):
                self.sim.step()
        return np.array(d.qpos), np.array(d.qvel)


def have_mujoco():
    for name in ("mujoco", "mujoco_py"):
        try:
            __import__(name)
            return name
        except Exception:
            continue
    return None
