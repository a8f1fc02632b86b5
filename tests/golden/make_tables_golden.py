#!/usr/bin/env python3
"""Fixtures for every table-like restatement on the path, derived from the reference's TEXT by evaluation.

Runs in the build container only (reads /root/reference, which does not travel to the GPU box):

    python tests/golden/make_tables_golden.py

What it writes (data: inputs and expected outputs, no source text):

* coeff_symmetry.json -- the image of arange(N) under the four stream-relabelling routines of src/optprop.F90, for
  every (lswitch_east, lswitch_north).  The assignment lines `coeff(lo:hi) = newcoeff([..] + off)` of each routine are
  *interpreted* (a tiny evaluator of that one statement form), so a commented-out line is a block that stays as it is,
  exactly as the compiled reference behaves.
* solver_tables.json -- is_inward / area_divider / dof / streams of t_solver_3_10 and t_solver_8_16
  (src/pprts.F90 allocate_pprts_solver_from_commandline's select type), inv_dof evaluated from its body (:5739-5752).
* lut_presets.json -- every `preset_*` parameter array of src/optprop_parameters.F90 as float32 values, and the
  dimension lists the LUT configs of 3_10 / 8_16 are built from (src/optprop_base.F90).
"""
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TENSTREAM_REFERENCE", "/root/reference")


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read().split("\n")


def find_sub(lines, name):
    """(first, last) 0-based line indices of `subroutine name ... end subroutine`."""
    start = next(i for i, l in enumerate(lines) if re.match(rf"\s*subroutine\s+{name}\b", l))
    end = next(i for i in range(start, len(lines)) if re.match(r"\s*end subroutine", lines[i]))
    return start, end


# ---------------------------------------------------------------- expression evaluation (integers only)
def ev(expr, env):
    expr = expr.strip()
    if not re.fullmatch(r"[\w\s+\-*/()]+", expr):
        raise ValueError(f"unexpected expression {expr!r}")
    return int(eval(expr.replace("/", "//"), {"__builtins__": {}}, dict(env)))


ASSIGN = re.compile(r"^\s*coeff\(([^:]+):([^)]+)\)\s*=\s*newcoeff\((.*)\)\s*$")


def run_symmetry(lines, first, last, n, east, north):
    """Interprets the body: `if (lswitch_x) then / newcoeff = coeff / coeff(a:b) = newcoeff(idx) / end if`."""
    env = {}
    coeff = np.arange(n, dtype=np.int64)
    newcoeff = coeff.copy()
    active = [True]
    for ln in range(first + 1, last):
        raw = lines[ln]
        code = raw.split("!")[0].strip()
        if not code:
            continue
        m = re.match(r"integer\(iintegers\),\s*parameter\s*::\s*(\w+)\s*=\s*(\d+)", code)
        if m:
            env[m.group(1)] = int(m.group(2))
            continue
        m = re.match(r"if\s*\((.*)\)\s*then", code)
        if m:
            cond = m.group(1).strip()
            val = {"lswitch_east": east, "lswitch_north": north, ".false.": False}.get(cond)
            if val is None:
                raise ValueError(f"line {ln + 1}: condition {cond!r}")
            active.append(active[-1] and val)
            continue
        if code == "end if":
            active.pop()
            continue
        if not active[-1]:
            continue
        if re.fullmatch(r"newcoeff(\(:\))?\s*=\s*coeff", code):
            newcoeff = coeff.copy()
            continue
        m = ASSIGN.match(code)
        if m:
            lo, hi = ev(m.group(1), env), ev(m.group(2), env)
            rhs = m.group(3).strip()
            mm = re.fullmatch(r"\[([\d,\s]+)\](.*)", rhs)
            if not mm:
                raise ValueError(f"line {ln + 1}: rhs {rhs!r}")
            base = np.array([int(t) for t in mm.group(1).split(",")], dtype=np.int64)
            rest = mm.group(2).strip()
            off = ev(rest[1:], env) if rest.startswith("+") else (0 if not rest else None)
            if off is None:
                raise ValueError(f"line {ln + 1}: offset {rest!r}")
            idx = base + off
            assert hi - lo + 1 == idx.size, f"line {ln + 1}"
            coeff[lo - 1:hi] = newcoeff[idx - 1]
            continue
        if re.match(r"(class|logical|real|select type|end select|return)\b", code):
            continue
        raise ValueError(f"line {ln + 1}: cannot interpret {code!r}")
    return coeff.tolist()


def make_symmetry():
    lines = read("src/optprop.F90")
    routines = {
        # name: (number of coefficients = dst * src, solver, coefficient family)
        "dir3_to_diff10_coeff_symmetry": (30, "3_10", "dir2diff"),
        "dir8_to_diff16_coeff_symmetry": (128, "8_16", "dir2diff"),
        "dir2dir8_coeff_symmetry": (64, "8_16", "dir2dir"),
        "dir2dir_coeff_symmetry_none": (9, "3_10", "dir2dir"),
    }
    out = {"source": "src/optprop.F90, interpreted by tests/golden/make_tables_golden.py",
           "meaning": "image[q] = index (0-based) of the table coefficient that ends up at position q (flat dst*S + src)",
           "routines": {}}
    for name, (n, solver, fam) in routines.items():
        a, b = find_sub(lines, name)
        ent = {"lines": f"{a + 1}-{b + 1}", "n": n, "solver": solver, "family": fam, "image": {}}
        for east in (0, 1):
            for north in (0, 1):
                ent["image"][f"e{east}n{north}"] = run_symmetry(lines, a, b, n, bool(east), bool(north))
        assert ent["image"]["e0n0"] == list(range(n))
        out["routines"][name] = ent
    dump("coeff_symmetry.json", out)


# ---------------------------------------------------------------- solver tables
def logical_list(txt):
    return [t.strip() == ".true." for t in txt.split(",") if t.strip()]


def make_solver_tables():
    lines = read("src/pprts.F90")
    out = {"source": "src/pprts.F90 (select type of the solver class), parsed by tests/golden/make_tables_golden.py",
           "solvers": {}}
    for cls in ("t_solver_3_10", "t_solver_8_16"):
        start = next(i for i, l in enumerate(lines) if re.match(rf"\s*class is \({cls}\)", l))
        end = next(i for i in range(start + 1, len(lines)) if re.match(r"\s*class (is|default)", lines[i]))
        body = " ".join(l.split("!")[0].strip().rstrip("&").lstrip("&") for l in lines[start:end])
        ent = {"lines": f"{start + 1}-{end}"}
        for grp in ("difftop", "diffside", "dirtop", "dirside"):
            m = re.search(rf"allocate\s*\(solver%{grp}%is_inward\((\d+)\)(?:,\s*source=\s*([^)]*?\]|\.true\.|\.false\.))?\s*\)",
                          body)
            n = int(m.group(1))
            src = m.group(2)
            if src is None:
                m2 = re.search(rf"solver%{grp}%is_inward\s*=\s*(\[[^\]]*\]|\.true\.|\.false\.)", body)
                src = m2.group(1)
            vals = logical_list(src.strip("[]")) if src.startswith("[") else [src == ".true."] * n
            assert len(vals) == n
            m3 = re.search(rf"solver%{grp}%area_divider\s*=\s*(\d+)", body)
            ent[grp] = {"dof": n, "is_inward": [int(v) for v in vals], "area_divider": int(m3.group(1)) if m3 else 1,
                        "streams": n // 2 if grp.startswith("diff") else n}
        # inv_dof (:5739-5752): partner stream in the opposite vertical direction
        a = next(i for i, l in enumerate(lines) if re.match(r"\s*pure function inv_dof\(dof\)", l))
        body = [l.split("!")[0].strip() for l in lines[a:a + 14]]
        assert "if (solver%difftop%is_inward(1)) then" in body and "inc = 1" in body and "inc = -1" in body
        assert "if (solver%difftop%is_inward(i1 + dof)) then" in body and "inv_dof = dof + inc" in body
        top = ent["difftop"]["is_inward"]
        inc = 1 if top[0] else -1
        ent["inv_dof"] = [d + inc if top[d] else d - inc for d in range(len(top))]
        ent["inv_dof_lines"] = f"{a + 1}-{a + 14}"
        out["solvers"][cls] = ent
    dump("solver_tables.json", out)


# ---------------------------------------------------------------- LUT presets
def make_presets():
    lines = read("src/optprop_parameters.F90")
    text = []
    for l in lines:
        l = l.split("!")[0].rstrip()
        text.append(l)
    joined = "\n".join(text)
    joined = re.sub(r"&\s*\n\s*&?", " ", joined)
    out = {"source": "src/optprop_parameters.F90 preset_* parameter arrays (float32 = irealLUT), parsed by "
                     "tests/golden/make_tables_golden.py", "presets": {}}
    consts = {}
    m = re.search(r"real\(irealLUT\),\s*parameter\s*::\s*peps_r\s*=\s*([^\n]+)", joined)
    if m:
        rhs = m.group(1).strip()
        # e.g. epsilon(peps_r) * N or a literal
        mm = re.fullmatch(r"([\d.eE+\-]+)(_irealLUT)?", rhs)
        if mm:
            consts["peps_r"] = float(mm.group(1))
    for m in re.finditer(r"real\(irealLUT\),\s*parameter\s*::\s*(preset_\w+)\((\d+)\)\s*=\s*\[([^\]]*)\]", joined):
        name, n, body = m.group(1), int(m.group(2)), m.group(3)
        body = re.sub(r"real\(irealLUT\)\s*::", "", body)
        toks = [t.strip() for t in body.split(",") if t.strip()]
        try:
            vals = []
            for t in toks:
                tt = t.replace("_irealLUT", "")
                if re.fullmatch(r"[+\-]?peps_r", tt):
                    if "peps_r" not in consts:
                        raise KeyError("peps_r")
                    vals.append((-1.0 if tt.startswith("-") else 1.0) * consts["peps_r"])
                else:
                    vals.append(float(tt))
        except (ValueError, KeyError):
            continue  # arrays built from expressions the path does not use
        assert len(vals) == n, name
        f32 = np.asarray(vals, dtype=np.float32)
        out["presets"][name] = {"n": n, "f32_hex": [v.tobytes().hex() for v in f32], "values": [float(v) for v in f32]}
    # the dimension lists of the configs on the path (src/optprop_base.F90 set_parameter_space: case ('LUT_3_10') ...)
    base = read("src/optprop_base.F90")
    out["configs"] = {}
    for want in ("LUT_3_10", "LUT_8_16"):
        a = next(i for i, l in enumerate(base) if re.match(rf"\s*case \('{want}'\)", l))
        b = next(i for i in range(a + 1, len(base)) if re.match(r"\s*case \(", base[i]))
        ent = {"lines": f"{a + 1}-{b}", "dirconfig": [], "diffconfig": []}
        for l in base[a:b]:
            code = l.split("!")[0].strip()
            m = re.match(r"call populate_op_dim\('(\w+)',\s*(.*)\)$", code)
            if not m:
                continue
            name, rest = m.group(1), m.group(2)
            which = "dirconfig" if "dirconfig" in rest else "diffconfig"
            mp = re.search(r"preset=(\w+)", rest)
            if mp:
                ent[which].append({"dim": name, "preset": mp.group(1), "n": out["presets"][mp.group(1)]["n"]})
            else:
                mn = re.match(r"(\d+)_iintegers", rest)
                mv = re.search(r"vrange=real\(\[([\d,\s.]+)\],\s*irealLUT\)", rest)
                lo, hi = [float(t) for t in mv.group(1).split(",")]
                ent[which].append({"dim": name, "n": int(mn.group(1)), "vrange": [lo, hi]})
        out["configs"][want] = ent
    dump("lut_presets.json", out)


if __name__ == "__main__":
    make_symmetry()
    make_solver_tables()
    make_presets()
