"""CPU tests of the LUT plumbing: `.mmap4` container round trip (tests/test_mmap in the reference) and the oracle's
coefficient lookup on the synthetic table."""
import numpy as np

from oracle import oracle as O
from tenstream_amd import lut, synthetic


def test_mmap4_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    t = rng.random((37, 100), dtype=np.float32)
    p = tmp_path / lut.diffuse_lut_filename("LUT")
    lut.write_mmap4(p, t)
    assert p.name == "LUT_diffuse_10.tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4"
    hdr = np.fromfile(p, dtype=np.uint64, count=6)
    assert list(hdr) == [4, 3700, 14800, 100, 37, 0]
    assert p.stat().st_size == lut.PAGESIZE + 14800
    back = lut.read_mmap4(p)
    assert back.shape == (37, 100) and np.array_equal(back, t)


def test_oracle_lookup_on_nodes_and_between():
    axes = lut.diffuse_axes("3_10")
    table = lut.synthetic_diffuse_table("3_10")
    assert table.shape == (31 * 20 * 23 * 6, 100)
    L = O.make_lut(axes, table)
    tau, w0, asp, g = axes
    dz, dx = 50.0, 100.0
    kext = float(tau[20]) / dz
    ksca = kext * float(w0[5])
    c = O.get_coeff_diff2diff(L, kext - ksca, ksca, float(g[2]), dz, dx)
    ia = 10  # aspect 0.5 lies between the nodes 0.422 and 0.562
    lo = table[20 + 31 * (5 + 20 * (ia + 23 * 2))]
    hi = table[20 + 31 * (5 + 20 * (ia + 1 + 23 * 2))]
    wgt = (np.float32(0.5) - asp[ia]) / (asp[ia + 1] - asp[ia])
    # tau/w0 computed in double then cast may sit 1 ulp off the node: still inside the 1e-3 snapping band
    np.testing.assert_allclose(c, (1 - wgt) * lo + wgt * hi, rtol=2e-6)
    C = c.reshape(10, 10)  # [dst, src]
    assert np.all(C.sum(axis=0) <= 1 + 1e-6)  # energy conservation survives interpolation
    # clamping: tau beyond the table -> last node
    c2 = O.get_coeff_diff2diff(L, 0.0, 1e3, 0.0, dz, dx)
    np.testing.assert_array_equal(c2, O.get_coeff_diff2diff(L, 0.0, 1e5, 0.0, dz, dx))


def test_alloc_coeff_field_skips_1d_layers():
    axes = lut.diffuse_axes("3_10")
    L = O.make_lut(axes, lut.synthetic_diffuse_table("3_10"))
    kabs, ksca, g = synthetic.cloud_field(5, 4, 6)
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    dz = np.full_like(kabs, 50.0)
    l1d = np.array([1, 0, 0, 0, 0, 1], dtype=np.uint8)
    c = O.alloc_coeff_diff2diff(L, kabs, ksca, g, dz, 100.0, l1d)
    assert np.all(c[:, :, 0] == 0) and np.all(c[:, :, 5] == 0) and np.all(c[:, :, 1:5].sum(axis=-1) > 0)
    assert np.array_equal(c.astype(np.float32).astype(np.float64), c)  # real(v, ireals): fp32-exact


def test_mmap4_written_in_pieces_equals_the_single_write(tmp_path):
    """write_mmap4_generated (tables of the reference's preset size are written chunk by chunk): same bytes as one write,
    header as src/mmap.F90:63-127 lays it out."""
    n, nc = 1000, 9
    full = lut.hashed_table_values(0, n, nc, salt=1)
    assert full.shape == (n, nc) and full.dtype == np.float32 and 0 <= full.min() and full.max() < 1.0 / nc
    lut.write_mmap4(tmp_path / "a.mmap4", full)
    lut.write_mmap4_generated(tmp_path / "b.mmap4", n, nc, lambda lo, hi: lut.hashed_table_values(lo, hi, nc, salt=1), chunk=300)
    assert (tmp_path / "a.mmap4").read_bytes() == (tmp_path / "b.mmap4").read_bytes()


def _mmap4_fixture():
    import importlib.util
    import json
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_mmap4_fixture", os.path.join(here, "make_mmap4_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, json.load(open(os.path.join(here, "mmap4_fixture.json"))), here


def test_mmap4_container_against_the_byte_fixture(tmp_path):
    """The committed fixture (tests/golden/mmap4_*: header page, first payload bytes, SHA-256 of the whole file) was built from
    the description of arr_to_binary_datafile_2d (src/mmap.F90:63-127) by a script that imports nothing from tenstream_amd.
    The product's own writer must produce those bytes, and its reader must see the array of that file."""
    import hashlib
    import os

    gen, doc, here = _mmap4_fixture()
    header = open(os.path.join(here, "mmap4_diffuse_3_10_header.bin"), "rb").read()
    assert len(header) == 4096 == lut.PAGESIZE
    words = np.frombuffer(header, dtype="<u8")
    assert list(words[:6]) == [4, 8556000, 34224000, 100, 85560, 0] and not words[6:].any()
    assert gen.header_page(4, 100, 85560) == header   # the generator is the description, executable
    path = tmp_path / doc["name"]
    assert gen.write_file(str(path)) == doc["sha256"] and path.stat().st_size == doc["file_bytes"] == 4096 + 34224000
    raw = open(path, "rb").read(4096 + 1024)
    assert raw[4096:] == open(os.path.join(here, "mmap4_diffuse_3_10_first_payload.bin"), "rb").read()
    # reader: (nentries, ncoeff) view of the Fortran (ncoeff, nentries) array, coefficient index fastest
    t = lut.read_mmap4(path)
    assert t.shape == (85560, 100)
    e = np.array([0, 1, 30, 31, 85559, 4242])[:, None]
    c = np.arange(100)[None, :]
    assert np.array_equal(t[e[:, 0]], (((7 * c + 13 * e) % 1009) / 131072.0).astype(np.float32))
    # writer: the same table through lut.write_mmap4 gives the fixture's bytes
    E, Cc = np.meshgrid(np.arange(85560), np.arange(100), indexing="ij")
    own = tmp_path / "own.mmap4"
    lut.write_mmap4(own, (((7 * Cc + 13 * E) % 1009) / 131072.0).astype(np.float32))
    assert hashlib.sha256(open(own, "rb").read()).hexdigest() == doc["sha256"]
    assert lut.diffuse_lut_filename("LUT") == doc["name"]
