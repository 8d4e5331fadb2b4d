#!/usr/bin/env python3
"""Render the reference's MJCF templates (build container only) and commit what they SAY as data:
tests/golden/mjcf_tables.json.  The templates are rendered with plain jinja2 exactly as
random_envs/jinja/template_renderer.py:16-19 does (size=..., sin, cos, pi) and parsed with ElementTree;
defaults classes are resolved for joints / geoms / motors; everything is converted to WORLD coordinates at
qpos0 so the oracle's and the kernels' model builders can be checked against it element by element."""
import json, math, os, sys
import xml.etree.ElementTree as ET

import jinja2
import numpy as np

REF = os.environ.get("REX_REFERENCE", "/root/reference")
ASSETS = os.path.join(REF, "random_envs", "jinja", "assets")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mjcf_tables.json")


def fl(s, n=None):
    v = [float(eval(x, {"__builtins__": {}})) if "/" in x else float(x) for x in s.split()]   # "0.2/2" literal in walker2d.xml:37
    return v


def qmul(a, b):
    return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                     a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])


def qrot(q, v):
    w, x, y, z = q
    R = np.array([[w*w+x*x-y*y-z*z, 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), w*w-x*x+y*y-z*z, 2*(y*z-w*x)],
                  [2*(x*z-w*y), 2*(y*z+w*x), w*w-x*x-y*y+z*z]])
    return R @ np.asarray(v)


def parse(xml, name):
    root = ET.fromstring(xml)
    comp = root.find("compiler").attrib
    glob = comp.get("coordinate", "local") == "global"
    deg = comp.get("angle", "degree") == "degree"
    ang = (math.pi / 180.0) if deg else 1.0
    dflt = root.find("default")
    dj = dict(dflt.find("joint").attrib) if dflt is not None and dflt.find("joint") is not None else {}
    dg = dict(dflt.find("geom").attrib) if dflt is not None and dflt.find("geom") is not None else {}
    dm = dict(dflt.find("motor").attrib) if dflt is not None and dflt.find("motor") is not None else {}
    opt = root.find("option")
    out = {"name": name, "coordinate": "global" if glob else "local", "option": dict(opt.attrib) if opt is not None else {},
           "flags": dict(opt.find("flag").attrib) if opt is not None and opt.find("flag") is not None else {},
           "settotalmass": float(comp.get("settotalmass", 0)), "bodies": [], "joints": [], "geoms": [], "motors": [], "pairs": []}

    def geom_rec(g, bname, bpos, bquat):
        a = dict(dg); a.update(g.attrib)
        rec = {"name": a.get("name"), "body": bname, "type": a.get("type", "sphere"), "size": fl(a["size"]),
               "friction": fl(a["friction"]) if "friction" in a else None, "contype": int(a.get("contype", 1)),
               "conaffinity": int(a.get("conaffinity", 1)), "condim": int(a.get("condim", 3)), "margin": float(a.get("margin", 0)),
               "solimp": fl(a["solimp"]) if "solimp" in a else None, "solref": fl(a["solref"]) if "solref" in a else None,
               "density": float(a.get("density", 1000))}
        if rec["type"] == "capsule":
            if "fromto" in a:
                ft = np.array(fl(a["fromto"])); p0, p1 = ft[:3], ft[3:]
                if not glob:
                    p0 = bpos + qrot(bquat, p0); p1 = bpos + qrot(bquat, p1)
            else:   # pos + axisangle + size = (radius, half length)
                pos = np.array(fl(a.get("pos", "0 0 0"))); aa = fl(a.get("axisangle", "0 0 1 0"))
                axis = np.array(aa[:3]) / np.linalg.norm(aa[:3]); th = aa[3] * ang
                gq = np.concatenate([[math.cos(th / 2)], axis * math.sin(th / 2)])
                d = qrot(gq, [0, 0, rec["size"][1]])
                p0, p1 = pos + d, pos - d
                if not glob:
                    p0 = bpos + qrot(bquat, p0); p1 = bpos + qrot(bquat, p1)
            rec["p0"], rec["p1"], rec["radius"] = list(map(float, p0)), list(map(float, p1)), rec["size"][0]
        elif rec["type"] == "sphere":
            pos = np.array(fl(a.get("pos", "0 0 0")))
            rec["center"] = list(map(float, pos if glob else bpos + qrot(bquat, pos))); rec["radius"] = rec["size"][0]
        return rec

    def walk(body, ppos, pquat, pname):
        pos = np.array(fl(body.attrib.get("pos", "0 0 0")))
        quat = np.array(fl(body.attrib["quat"])) if "quat" in body.attrib else np.array([1.0, 0, 0, 0])
        quat = quat / np.linalg.norm(quat)
        if glob:
            wpos, wquat = pos, quat
        else:
            wpos, wquat = ppos + qrot(pquat, pos), qmul(pquat, quat)
        bname = body.attrib["name"]
        out["bodies"].append({"name": bname, "parent": pname, "pos": list(map(float, wpos))})
        for j in body.findall("joint"):
            a = dict(dj); a.update(j.attrib)
            jpos = np.array(fl(a.get("pos", "0 0 0"))); axis = np.array(fl(a.get("axis", "0 0 1")))
            if not glob:
                jpos = wpos + qrot(wquat, jpos); axis = qrot(wquat, axis)
            lim = a.get("limited", "false") == "true"
            rng = [x * (ang if a.get("type", "hinge") == "hinge" else 1.0) for x in fl(a["range"])] if "range" in a else [0.0, 0.0]
            out["joints"].append({"name": a["name"], "body": bname, "type": a.get("type", "hinge"), "pos": list(map(float, jpos)),
                                  "axis": list(map(float, axis / max(np.linalg.norm(axis), 1e-15))), "limited": lim, "range": rng,
                                  "armature": float(a.get("armature", 0)), "damping": float(a.get("damping", 0)),
                                  "stiffness": float(a.get("stiffness", 0)), "ref": float(a.get("ref", 0)),
                                  "solimplimit": fl(a["solimplimit"]) if "solimplimit" in a else None})
        for g in body.findall("geom"):
            out["geoms"].append(geom_rec(g, bname, wpos, wquat))
        for b in body.findall("body"):
            walk(b, wpos, wquat, bname)

    wb = root.find("worldbody")
    for g in wb.findall("geom"):
        out["geoms"].append(geom_rec(g, "world", np.zeros(3), np.array([1.0, 0, 0, 0])))
    for b in wb.findall("body"):
        walk(b, np.zeros(3), np.array([1.0, 0, 0, 0]), "world")
    for m in root.find("actuator").findall("motor"):
        a = dict(dm); a.update(m.attrib)
        out["motors"].append({"joint": a["joint"], "gear": float(a.get("gear", 1)), "ctrlrange": fl(a["ctrlrange"])})
    c = root.find("contact")
    if c is not None:
        pd = {}
        for d2 in dflt.findall("default"):
            if d2.find("pair") is not None:
                pd[d2.attrib["class"]] = dict(d2.find("pair").attrib)
        for p in c.findall("pair"):
            a = dict(pd.get(p.attrib.get("class"), {})); a.update(p.attrib)
            out["pairs"].append({"geom1": a["geom1"], "geom2": a["geom2"], "condim": int(a.get("condim", 3)),
                                 "friction": fl(a["friction"]) if "friction" in a else None,
                                 "solimp": fl(a["solimp"]) if "solimp" in a else None})
    return out


def main():
    env = jinja2.Environment(loader=jinja2.FileSystemLoader(ASSETS))
    render = lambda f, size: env.get_template(f).render(size=size, sin=math.sin, cos=math.cos, pi=np.pi)
    out = {
        "hopper": parse(render("hopper.xml", [.4, .45, .5, .39]), "hopper"),
        "walker2d": parse(render("walker2d.xml", [.4, .45, .6, .2]), "walker2d"),
        "walker2d_alt": parse(render("walker2d.xml", [.5, .3, .7, .25]), "walker2d_alt"),
        "halfcheetah": parse(render("half_cheetah.xml", [1., .15, .145, .15, .094, .133, .106, .07]), "halfcheetah"),
        "humanoid": parse(render("humanoid.xml", []), "humanoid"),
    }
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, {k: (len(v["bodies"]), len(v["joints"]), len(v["geoms"]), len(v["motors"]), len(v["pairs"])) for k, v in out.items()})


if __name__ == "__main__":
    main()
