#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own result artifacts.

Runs ONLY in the build container (needs /root/reference).  The fixtures are data: inputs and
expected outputs decoded from LotkaVolterra/results/*.jld2 (written by scenario_1.jl:210-213,
scenario_2.jl:250-253, scenario_3.jl:211-214, hudson_bay.jl:231-235): saved states, time grids,
DEStats, the last-step integrator cache, the Tsit5/Vern7 coefficient tableaux that upstream
OrdinaryDiffEq stored inside the ODESolution, training data, NN parameters and loss histories.

usage: python tools/make_golden.py [/root/reference]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jld2_reader import JLD2File  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
RES = os.path.join(REF, "LotkaVolterra", "results")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def tolist(x):
    if isinstance(x, np.ndarray):
        return [float(v) for v in x.ravel()] if x.dtype.kind == "f" else [int(v) for v in x.ravel()]
    return x


def flat_params(p):
    """Lux ComponentVector / NamedTuple / plain vector -> flat list in Julia memory order.
    Per layer [vec(W) column-major (out x in); b]  (scenario_1.jl:62-66, Lux.setup)."""
    if isinstance(p, np.ndarray):
        return tolist(p), str(p.dtype)
    if isinstance(p, dict) and "data" in p:
        return tolist(p["data"]), str(p["data"].dtype)
    if isinstance(p, dict):
        out = []
        dt = None
        for lname in sorted(p.keys(), key=lambda s: int(s.split("_")[1])):
            out += tolist(p[lname]["weight"]) + tolist(p[lname]["bias"])
            dt = str(p[lname]["weight"].dtype)
        return out, dt
    raise TypeError(type(p))


def flatten_tab(tab, prefix=""):
    out = {}
    for k, v in tab.items():
        if isinstance(v, dict):
            out.update(flatten_tab(v, prefix))
        else:
            out[prefix + k] = float(v)
    return out


def solution(s):
    u = s["u"]
    dt = str(u[0].dtype)
    d = {
        "dtype": dt,
        "t": tolist(np.asarray(s["t"])),
        "u": [tolist(x) for x in u],
        "destats": {k: int(s["destats"][k]) for k in ("nf", "naccept", "nreject")},
        "u0": tolist(s["prob"]["u0"]),
        "tspan": [float(s["prob"]["tspan"]["1"]), float(s["prob"]["tspan"]["2"])],
    }
    p = s["prob"].get("p")
    if isinstance(p, np.ndarray):
        d["p"] = tolist(p)
    cache = s["interp"].get("cache")
    tab = None
    if isinstance(cache, dict) and "tab" in cache:        # in-place cache: k-stages + tableau
        d["last_step_cache"] = {k: tolist(v) for k, v in cache.items() if isinstance(v, np.ndarray)}
        d["alg"] = "Vern7" if "k10" in cache else "Tsit5"
        tab = flatten_tab(cache["tab"])
    elif isinstance(cache, dict) and "c1" in cache:      # out-of-place: the cache IS the tableau
        d["alg"] = "Vern7" if "a021" in cache else "Tsit5"
        tab = flatten_tab(cache)
    return d, tab


def trange(t):
    if isinstance(t, np.ndarray):
        return tolist(t)
    # StepRangeLen{Float64,TwicePrecision,...}: value_i = ref + (i-offset)*step
    ref, step = t["ref"], t["step"]
    if isinstance(ref, dict):
        ref = ref["hi"] + ref["lo"]
        step = step["hi"] + step["lo"]
    return [float(ref + (i + 1 - t["offset"]) * step) for i in range(t["len"])]


def main():
    os.makedirs(OUT, exist_ok=True)
    tabs = {}
    for fn in sorted(os.listdir(RES)):
        if not fn.endswith(".jld2"):
            continue
        f = JLD2File(os.path.join(RES, fn))
        doc = {"source": "LotkaVolterra/results/" + fn}
        for k in f.keys():
            if k in ("neural_network", "result", "model"):
                continue
            v = f[k]
            if isinstance(v, dict) and "destats" in v:
                doc[k], tab = solution(v)
                if tab is not None:
                    key = doc[k]["alg"].lower() + "_" + doc[k]["dtype"]
                    tabs.setdefault(key, tab)
            elif k in ("initial_parameters", "trained_parameters"):
                doc[k], doc[k + "_dtype"] = flat_params(v)
            elif k == "t":
                doc[k] = trange(v)
            elif isinstance(v, np.ndarray):
                doc[k] = {"shape_julia": list(v.shape[::-1]), "dtype": str(v.dtype),
                          "data_colmajor": tolist(v)}
        name = fn.replace(".jld2", ".json")
        with open(os.path.join(OUT, name), "w") as fh:
            json.dump(doc, fh)
        print(name, os.path.getsize(os.path.join(OUT, name)), "bytes; keys:", list(doc.keys()))
    with open(os.path.join(OUT, "tableaux.json"), "w") as fh:
        json.dump(tabs, fh, indent=0)
    print("tableaux:", {k: len(v) for k, v in tabs.items()})


if __name__ == "__main__":
    main()
