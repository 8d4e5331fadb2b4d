#!/usr/bin/env python3
"""Extract the per-env task TABLES from the reference's task-definition files (build container only) and
commit them as data: tests/golden/task_tables.json.

The MuJoCo-backed task files cannot be imported (mujoco_py absent), but the tables the build needs are plain
literals in their source: dyn_ind_to_name, the dict literals inside get_search_bounds_mean /
get_task_lower_bound, noise_level, preferred_lr, reward_threshold, frame_skip (MujocoEnv.__init__ call), and
the gym.envs.register(...) calls.  They are read with `ast` (no execution of reference code)."""
import ast, json, os, sys

REF = os.environ.get("REX_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "task_tables.json")
FILES = ["random_envs/random_cartpole.py"] + ["random_envs/jinja/" + f for f in (
    "random_hopper.py", "random_half_cheetah.py", "random_walker2d.py", "random_humanoid.py",
    "random_hopper_unmodeled.py", "random_half_cheetah_unmodeled.py", "random_walker2d_unmodeled.py",
    "random_humanoid_unmodeled.py")]


def lit(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return None


def main():
    out = {}
    for rel in FILES:
        src = open(os.path.join(REF, rel)).read()
        tree = ast.parse(src)
        rec = {"registered": []}
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and len(node.targets) == 1:
                t = node.targets[0]
                name = t.attr if isinstance(t, ast.Attribute) else (t.id if isinstance(t, ast.Name) else None)
                if name in ("dyn_ind_to_name", "noise_level", "preferred_lr", "reward_threshold", "search_bounds_mean",
                            "lowest_value"):
                    v = lit(node.value)
                    if v is not None:
                        rec[name] = v if not isinstance(v, dict) else {str(k): val for k, val in v.items()}
            if isinstance(node, ast.Call):
                f = node.func
                fname = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", "")
                if fname == "register":
                    kw = {k.arg: lit(k.value) for k in node.keywords}
                    rec["registered"].append({"id": kw.get("id"), "max_episode_steps": kw.get("max_episode_steps"),
                                              "kwargs": kw.get("kwargs")})
                if fname == "__init__" and isinstance(f, ast.Attribute) and getattr(f.value, "id", "") == "MujocoEnv":
                    args = [lit(a) for a in node.args[1:]]
                    rec["xml"], rec["frame_skip"] = args[0], args[1]
        out[os.path.basename(rel)] = rec
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, {k: (len(v.get("dyn_ind_to_name", {})), [r["id"] for r in v["registered"]]) for k, v in out.items()})


if __name__ == "__main__":
    main()
