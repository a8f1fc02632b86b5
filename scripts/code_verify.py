"""Compare libtsx's device code AS IT SITS IN DEVICE MEMORY with the bytes in the file (diagnostics, round 6).

libtsx.so carries eight gfx950 code objects (one per translation unit, clang offload bundles in .hip_fatbin).  Every unit has a
probe kernel (TSX_CODE_PROBE, tenstream_amd/csrc/tsx_host.hpp) that reports its own program counter and copies words from
pc + delta; this module knows each unit's ELF (section and symbol tables, the s_getpc_b64 inside the probe) and so can read back
the unit's whole .text through `tsx_debug_code_read` and compare it byte for byte with the file.

    python scripts/code_verify.py            # on a GPU box: verify all units in this process, print a summary
    from scripts import code_verify; code_verify.verify(lib)   # -> list of mismatch records (empty = intact)
"""
import ctypes as C
import os
import re
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ["api", "spmv310", "spmv816", "pc", "pcs", "pcsflow", "dedup", "peer"]


def _code_objects(blob):
    """(offset, size) of every gfx950 code object bundled in the shared library."""
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        o = m.start() + 24
        (n,) = struct.unpack_from("<Q", blob, o)
        o += 8
        for _ in range(n):
            off, sz, tl = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tl]
            o += tl
            if sz and b"gfx950" in triple:
                out.append((m.start() + off, sz))
    return out


def _elf(blob, base):
    """Minimal ELF64 reader: sections by name, symbols by name."""
    assert blob[base:base + 4] == b"\x7fELF"
    shoff, = struct.unpack_from("<Q", blob, base + 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, base + 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, base + shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    strtab = secs[shstrndx]

    def sname(o, tab):
        s = base + tab["off"] + o
        return blob[s:blob.index(b"\0", s)].decode()

    by = {}
    for s in secs:
        s["sname"] = sname(s["name"], strtab)
        by[s["sname"]] = s
    syms = {}
    st = by.get(".symtab")
    if st:
        stt = secs[st["link"]]
        for i in range(st["size"] // 24):
            nm, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", blob, base + st["off"] + i * 24)
            syms[sname(nm, stt)] = (value, size, shndx)
    return by, syms


def units(path=None):
    """{unit: dict(text bytes, text addr, delta = text addr - value of the probe's pc)} for every code object with a probe."""
    path = path or os.environ.get("TSX_LIB") or os.path.join(ROOT, "tenstream_amd", "lib", "libtsx.so")
    blob = open(path, "rb").read()
    out = {}
    for off, sz in _code_objects(blob):
        secs, syms = _elf(blob, off)
        text = secs[".text"]
        for u in UNITS:
            sym = syms.get("tsx_k_code_probe_" + u)
            if not sym:
                continue
            value, size, _ = sym
            fo = off + text["off"] + (value - text["addr"])
            body = blob[fo:fo + size]
            hit = None
            for w in range(0, len(body) - 3, 4):   # s_getpc_b64 sdst: SOP1 0xBE80_1Cxx with sdst in bits 16..22
                (word,) = struct.unpack_from("<I", body, w)
                if word & 0xFF80FF00 == 0xBE801C00:
                    hit = w
                    break
            assert hit is not None, f"no s_getpc_b64 in the probe of unit {u}"
            pc_value = value + hit + 4
            out[u] = dict(text=blob[off + text["off"]:off + text["off"] + text["size"]], addr=text["addr"], delta=text["addr"] - pc_value,
                          syms={k: v for k, v in syms.items() if v[2] != 0 and text["addr"] <= v[0] < text["addr"] + text["size"]})
    return out


def verify(lib, device=-1, which=None, info=None):
    """Read back every unit's .text and compare with the file.  -> list of (unit, byte offset in .text, run length, nearest symbol,
    first device words, file words) per run of differing words."""
    info = info or units()
    lib.tsx_debug_code_read.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.tsx_debug_code_read.restype = C.c_int
    bad = []
    for u in (which or UNITS):
        d = info[u]
        n = len(d["text"]) // 4
        buf = (C.c_uint32 * n)()
        pc = C.c_ulonglong(0)
        rc = lib.tsx_debug_code_read(device, UNITS.index(u), d["delta"], n, buf, C.byref(pc))
        if rc:
            bad.append((u, -1, 0, "tsx_debug_code_read failed rc=%d" % rc, [], []))
            continue
        dev = bytes(buf)
        if dev == d["text"][:4 * n]:
            continue
        ws = struct.unpack("<%dI" % n, dev)
        fs = struct.unpack("<%dI" % n, d["text"][:4 * n])
        i = 0
        by_addr = sorted((v[0], k) for k, v in d["syms"].items())
        while i < n:
            if ws[i] == fs[i]:
                i += 1
                continue
            j = i
            while j < n and (ws[j] != fs[j] or (j + 1 < n and ws[j + 1] != fs[j + 1])):
                j += 1
            a = d["addr"] + 4 * i
            near = max((s for s in by_addr if s[0] <= a), default=(0, "?"))
            bad.append((u, 4 * i, 4 * (j - i), "%s+%d" % (near[1], a - near[0]), ["%08x" % w for w in ws[i:min(j, i + 16)]],
                        ["%08x" % w for w in fs[i:min(j, i + 16)]]))
            i = j
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--list":
        for u, d in units().items():
            print(u, "text bytes", len(d["text"]), "addr 0x%x" % d["addr"], "delta", d["delta"], "kernels", sum(1 for k in d["syms"]))
        sys.exit(0)
    sys.path.insert(0, ROOT)
    from tenstream_amd import _lib

    lib = _lib.load()
    bad = verify(lib)
    print("code_verify: %d differing runs" % len(bad))
    for b in bad[:40]:
        print(b)
    sys.exit(1 if bad else 0)
