#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ from the known-answer numbers of the reference's own
pFUnit suites (transcribed data: inputs and expected outputs, each with its reference file:line).

The reference cannot be built or imported here (Fortran + MPI + PETSc), so nothing is
executed from /root/reference; this script only serialises the vectors.  Run:  python tests/golden/make_golden.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


# ---- tests/eddington/test_delta_eddington.F90:20-140 (targets of eddington_coeff_ec, tol 1e-4) -------------
# inp = [tau, omega0, g, mu0]; out = [a11, a12, a13, a23, a33]
eddington = {
    "source": "tests/eddington/test_delta_eddington.F90:12-140",
    "tol": 1e-4,
    "cases": [
        {"line": 20, "inp": [0.4824516550e-01, 0.5542391539, 0.4550637007, 0.5],
         "targ": [0.947543502, 1.03599848e-02, 1.65330153e-02, 3.33561152e-02, 0.908018649]},
        {"line": 30, "inp": [0.4824516550e-01, 0.5542391539, 0.4550637007, 1.0],
         "targ": [0.947543502, 1.03599848e-02, 4.14878177e-03, 2.13973932e-02, 0.952900112]},
        {"line": 40, "inp": [0.2018013448, 0.3797843754, 0.4556422830, 1.0],
         "targ": [0.754876196, 2.39296965e-02, 1.03387358e-02, 5.08127585e-02, 0.817257285]},
        {"line": 50, "inp": [0.1731484532, 0.6180083156, 0.4121485054, 1.0],
         "targ": [0.836549520, 3.96644436e-02, 1.88425109e-02, 7.29937181e-02, 0.841012776]},
        {"line": 60, "inp": [0.1931012775e-04, 0.4384377003, -0.0, 0.7070999742],
         "targ": [0.999971986, 6.34953813e-06, 6.10753023e-06, 4.93132757e-06, 0.999972701]},
        {"line": 70, "inp": [4.895462513, 0.3104626103e-05, -0.0, 0.49999997019767761],
         "targ": [5.59581749e-05, 5.82118503e-07, 7.76157776e-07, 8.66550609e-10, 5.59570581e-05]},
        {"line": 80, "inp": [3.2662250689525390e-011, 0.99999171495417127, 0.0, 0.17364817766693041],
         "targ": [1.00000000, 2.45013107e-11, 9.40557285e-11, 9.40355432e-11, 1.00000000]},
        {"line": 90, "inp": [2.9317851124478626e-012, 1.0, 0.0, 0.17364817766693041],
         "targ": [0.99999999999779732, 2.2026824808563106e-012, 8.4443208651779663e-012, 8.4384068595938508e-012,
                  0.99999999998311651]},
        {"line": 99, "inp": [1.93321303e-10, 0.999984443, 2.22044605e-16, 0.17364817766693041],
         "targ": [0.99999999985500665, 1.4499335065920604e-010, 5.5664139018583103e-010, 5.5664738411040951e-010,
                  0.99999999888670699]},
        {"line": 108, "inp": [7.89528581e-11, 0.999988437, 2.22044605e-16, 0.17364817766693041],
         "targ": [0.99999999994077626, 5.9209526170889148e-011, 2.2734076817292897e-010, 2.2733741157516502e-010,
                  0.99999999954532859]},
        {"line": 117, "inp": [1.3865453490508738e-011, 0.99999499320987351, 0.0, 0.17364817766693041],
         "targ": [0.99999999998959765, 1.0402345651527867e-011, 3.9931125946960987e-011, 3.9926650483275707e-011,
                  0.99999999992015198]},
        {"line": 126, "inp": [113.59224626216431, 2.7005225550306174e-008, 0.0, 0.17364817766693041],
         "targ": [0.0, 0.0, 9.6352077372196252e-009, 4.1335735289420085e-024, 0.0]},
        {"line": 135, "inp": [1.4503301e-02, 1.5233955e-12, 1.1920928e-07, 1.0],
         "targ": [0.971410036, 1.60991978e-14, 1.08102478e-14, 1.08094940e-14, 0.985601366]},
    ],
}
dump("eddington_ec.json", eddington)

# ---- tests/test_search/test_search.F90:13-50 (search_sorted_bisection, exact) -----------------------------------
search = {
    "source": "tests/test_search/test_search.F90:13-50",
    "cases": [
        {"arr": [-10, 0, 2], "pairs": [[-20, 1.0], [-10, 1.0], [-7.5, 1.25], [-5, 1.5], [-0.0, 2.0], [0.0, 2.0], [1, 2.5],
                                       [2, 3.0], [3, 3.0]]},
        {"arr": [1], "pairs": [[-1, 1.0], [1, 1.0], [2, 1.0]]},
        {"arr": [0, 1], "pairs": [[-1, 1.0], [1, 2.0], [2, 2.0]]},
        {"arr": [10, 0, -2], "pairs": [[20, 1.0], [10, 1.0], [7.5, 1.25], [5, 1.5], [0.0, 2.0], [-0.0, 2.0], [-1, 2.5],
                                       [-2, 3.0], [-3, 3.0]]},
    ],
}
dump("search.json", search)

# ---- tests/interpolation/test_interp.F90 (interp_vec_simplex_nd known answers; tol sqrt(eps32)*10) ---------------
# db given as Fortran-ordered nested index lists flattened column-major (first unravelled dim fastest)
interp = {"source": "tests/interpolation/test_interp.F90:14-225,395-464", "tol": 10 * (2.0 ** -23) ** 0.5, "cases": []}
# 2-D table [0,1;2,4]: db_2d(1,1,1)=0,(1,2,1)=1,(1,1,2)=2,(1,2,2)=4   (:36-41)
db = [0.0, 1.0, 2.0, 4.0]
c = {"line": 36, "shape": [2, 2], "db": db, "queries": []}
for i in (1, 2):
    for j in (1, 2):
        c["queries"].append([[float(i), float(j)], db[(i - 1) + 2 * (j - 1)]])
c["queries"] += [[[1.5, 1.0], 0.5], [[1.75, 1.0], 0.75], [[1.5, 2.0], 3.0], [[1.75, 2.0], 3.5], [[1.0, 1.5], 1.0],
                 [[1.0, 1.75], 1.5], [[2.0, 1.5], 2.5], [[2.0, 1.75], 3.25]]  # :55-72
interp["cases"].append(c)
interp["cases"].append({"line": 81, "shape": [2, 2], "db": [0.0, 2.0, 2.0, 4.0], "queries": [[[1.5, 1.5], 2.0]]})  # :81-88
c = {"line": 97, "shape": [2, 2], "db": [0.0, 0.0, 2.0, 2.0], "queries": []}  # :97-108
for j in range(10):
    for i in range(10):
        c["queries"].append([[1.0 + i * 0.1, 1.0 + j * 0.1], j * 0.1 * 2])
interp["cases"].append(c)
# 3-D (:132-160): corners + centre == mean
db3 = [0.0, 2.0, 2.0, 4.0, 10.0, 12.0, 12.0, 14.0]
c = {"line": 132, "shape": [2, 2, 2], "db": db3, "queries": []}
for i in (1, 2):
    for j in (1, 2):
        for k in (1, 2):
            c["queries"].append([[float(i), float(j), float(k)], db3[(i - 1) + 2 * (j - 1) + 4 * (k - 1)]])
c["queries"].append([[1.5, 1.5, 1.5], sum(db3) / 8])
interp["cases"].append(c)
# 4-D (:184-224)
db4 = db3 + db3
c = {"line": 184, "shape": [2, 2, 2, 2], "db": db4, "queries": []}
for i in (1, 2):
    for j in (1, 2):
        for k in (1, 2):
            for l in (1, 2):
                c["queries"].append([[float(i), float(j), float(k), float(l)],
                                     db4[(i - 1) + 2 * (j - 1) + 4 * (k - 1) + 8 * (l - 1)]])
c["queries"].append([[1.5, 1.5, 1.5, 1.5], sum(db4) / 16])
interp["cases"].append(c)
# 6-D, 2 vectors (:396-464): x = i + 2j + 8k + 16l + 32m + 64n, second vector uses doubled arguments
interp["six_d"] = {"line": 396, "Nv": 4, "weights": [1, 2, 8, 16, 32, 64]}
dump("interp.json", interp)

# ---- tests/test_boxmc_3_10/test_boxmc_3_10.F90:151,176,188-235: one full 10x10 3_10 diffuse block ----------------
# S_target(dst) per src (1-based src), bg = [kabs 1e-3, ksca 0, g 0], dx = dy = 100, dz = 50 (:42-44), atol 1e-3 rtol 1e-2 (:22)
top, a, b = 0.56173, 0.104806, 0.1424402
block = {
    "source": "tests/test_boxmc_3_10/test_boxmc_3_10.F90:151,176,188-235",
    "kabs": 1e-3, "ksca": 0.0, "g": 0.0, "dx": 100.0, "dy": 100.0, "dz": 50.0, "atol": 1e-3, "rtol": 1e-2,
    "S_by_src": [
        [0.390156, 0.0, 0.0, 0.0, 0.1404375, 0.1404375, 0.0, 0.0, 0.1404375, 0.1404375],  # src 1  (:176)
        [0.0, 0.390156, 0.1404375, 0.1404375, 0.0, 0.0, 0.1404375, 0.1404375, 0.0, 0.0],  # src 2  (:151)
        [0.0, top, a, 0.0, 0.0, 0.0, b, b, 0.0, 0.0],   # src 3  (:199)
        [0.0, top, 0.0, a, 0.0, 0.0, b, b, 0.0, 0.0],   # src 4  (:204)
        [top, 0.0, 0.0, 0.0, a, 0.0, 0.0, 0.0, b, b],   # src 5  (:209)
        [top, 0.0, 0.0, 0.0, 0.0, a, 0.0, 0.0, b, b],   # src 6  (:214)
        [0.0, top, b, b, 0.0, 0.0, a, 0.0, 0.0, 0.0],   # src 7  (:219)
        [0.0, top, b, b, 0.0, 0.0, 0.0, a, 0.0, 0.0],   # src 8  (:224)
        [top, 0.0, 0.0, 0.0, b, b, 0.0, 0.0, a, 0.0],   # src 9  (:229)
        [top, 0.0, 0.0, 0.0, b, b, 0.0, 0.0, 0.0, a],   # src 10 (:234)
    ],
}
dump("boxmc_3_10_block.json", block)

# ---- tests/test_pprts_coord_native/test_pprts_coord_native.F90:13-77 (1 rank, exact) and :79-176 (4 ranks) --------
coord = {
    "source": "tests/test_pprts_coord_native/test_pprts_coord_native.F90:13-176",
    "one_rank": {"Nz": 5, "Nx": 8, "Ny": 6, "xs": 0, "xe": 7, "xm": 8, "ys": 0, "ye": 5, "ym": 6, "gxs": -1, "gxe": 8,
                 "gxm": 10, "gys": -1, "gye": 6, "gym": 8, "neighbors": [0, 0, 0, 0]},
    "four_rank": {"Nz": 3, "Nx": 10, "Ny": 8, "nproc": 4},
}
dump("coord_native.json", coord)
