"""A minimal writer of uncompressed 64-bit little-endian .blend files, for tests of the built-in loader.

It writes its own small SDNA -- only the structs and fields the loader reads (Scene.base, Base, Object{id.name, type,
parent, data, obmat}, Mesh{totvert, totpoly, totloop, totcol, mvert, mpoly, mloop, mat}, MVert, MPoly, MLoop, Lamp,
Camera, Material) -- so the files are not what Blender would write, but they are valid in the sense of the format:
header, blocks with old-pointer addresses, a DNA1 block that describes every struct used, ENDB.  TEST
INFRASTRUCTURE: nothing in the product imports this."""
import struct

import numpy as np

# struct name -> [(type, field name as DNA spells it)]
STRUCTS = [
    ("ID", [("char", "name[64]")]),
    ("ListBase", [("void", "*first"), ("void", "*last")]),
    ("Scene", [("ID", "id"), ("ListBase", "base"), ("Object", "*camera")]),
    ("Base", [("Base", "*next"), ("Base", "*prev"), ("Object", "*object")]),
    ("Object", [("ID", "id"), ("short", "type"), ("short", "pad[3]"), ("Object", "*parent"), ("void", "*data"), ("float", "obmat[4][4]")]),
    ("Mesh", [("ID", "id"), ("int", "totvert"), ("int", "totpoly"), ("int", "totloop"), ("short", "totcol"), ("short", "pad"),
              ("MVert", "*mvert"), ("MPoly", "*mpoly"), ("MLoop", "*mloop"), ("Material", "**mat")]),
    ("MVert", [("float", "co[3]"), ("short", "no[3]"), ("char", "flag"), ("char", "bweight")]),
    ("MPoly", [("int", "loopstart"), ("int", "totloop"), ("short", "mat_nr"), ("char", "flag"), ("char", "pad")]),
    ("MLoop", [("int", "v"), ("int", "e")]),
    ("Lamp", [("ID", "id"), ("short", "type"), ("short", "pad"), ("float", "r"), ("float", "g"), ("float", "b"), ("float", "energy"),
              ("float", "spotsize")]),
    ("Camera", [("ID", "id"), ("float", "lens")]),
    ("Material", [("ID", "id")] + [("float", n) for n in ("r", "g", "b", "specr", "specg", "specb", "ambr", "ambg", "ambb")]),
]
BASIC = {"char": 1, "short": 2, "int": 4, "float": 4, "void": 0}
OB_MESH, OB_LAMP, OB_CAMERA, OB_EMPTY = 1, 10, 11, 0
LA_LOCAL, LA_SUN, LA_SPOT = 0, 1, 2


def _field_size(ftype, fname, lengths):
    n = 1
    for part in fname.split("[")[1:]:
        n *= int(part.rstrip("]"))
    return (8 if fname.startswith("*") else lengths[ftype]) * n


def _sdna():
    lengths = dict(BASIC)
    for name, fields in STRUCTS:  # every struct only uses structs listed before it by value
        lengths[name] = sum(_field_size(t, f, lengths) for t, f in fields)
    types = list(BASIC) + [n for n, _ in STRUCTS]
    names = []
    for _, fields in STRUCTS:
        for _, f in fields:
            if f not in names:
                names.append(f)

    def pad4(b):
        return b + b"\0" * (-len(b) % 4)
    out = b"SDNA" + b"NAME" + struct.pack("<I", len(names)) + b"".join(n.encode() + b"\0" for n in names)
    out = pad4(out) + b"TYPE" + struct.pack("<I", len(types)) + b"".join(t.encode() + b"\0" for t in types)
    out = pad4(out) + b"TLEN" + b"".join(struct.pack("<H", lengths[t]) for t in types)
    out = pad4(out) + b"STRC" + struct.pack("<I", len(STRUCTS))
    for name, fields in STRUCTS:
        out += struct.pack("<HH", types.index(name), len(fields))
        for t, f in fields:
            out += struct.pack("<HH", types.index(t), names.index(f))
    return out, lengths


class BlendFile:
    def __init__(self):
        self.blocks = []
        self.next_ptr = 0x10000
        self.sdna, self.lengths = _sdna()
        self.struct_index = {n: i for i, (n, _) in enumerate(STRUCTS)}
        self.bases = []

    def _add(self, code, payload, struct_name=None, count=1):
        ptr = self.next_ptr
        self.next_ptr += 0x1000
        self.blocks.append((code, payload, ptr, self.struct_index.get(struct_name, 0), count))
        return ptr

    @staticmethod
    def _id(prefix, name):
        return (prefix + name).encode().ljust(64, b"\0")

    def material(self, name, rgb, spec=(0.5, 0.5, 0.5), amb=(0.0, 0.0, 0.0)):
        return self._add(b"MA\0\0", self._id("MA", name) + struct.pack("<9f", *rgb, *spec, *amb), "Material")

    def mesh(self, name, verts, normals, polys, materials=()):
        """polys: list of (vertex numbers, material slot); materials: Material block addresses per slot (0 = empty slot)"""
        mv = b"".join(struct.pack("<3f3hbb", *[float(x) for x in v], *[int(round(c * 32767)) for c in n], 0, 0) for v, n in zip(verts, normals))
        loops, mp = [], b""
        for vs, slot in polys:
            mp += struct.pack("<iihbb", len(loops), len(vs), slot, 0, 0)
            loops.extend(vs)
        ml = b"".join(struct.pack("<ii", v, 0) for v in loops)
        pv = self._add(b"DATA", mv, "MVert", len(verts))
        pp = self._add(b"DATA", mp, "MPoly", len(polys))
        pl = self._add(b"DATA", ml, "MLoop", len(loops))
        pm = self._add(b"DATA", b"".join(struct.pack("<Q", m) for m in materials), None, len(materials)) if materials else 0
        body = self._id("ME", name) + struct.pack("<iiihh", len(verts), len(polys), len(loops), len(materials), 0) + struct.pack("<4Q", pv, pp, pl, pm)
        return self._add(b"ME\0\0", body, "Mesh")

    def lamp(self, name, kind, rgb, energy, spotsize=0.0):
        return self._add(b"LA\0\0", self._id("LA", name) + struct.pack("<hh5f", kind, 0, *rgb, energy, spotsize), "Lamp")

    def camera(self, name):
        return self._add(b"CA\0\0", self._id("CA", name) + struct.pack("<f", 35.0), "Camera")

    def object(self, name, kind, data, obmat, parent=0):
        """obmat: 4x4 numpy matrix in the usual row/column sense (translation in the last column); stored as Blender
        does, column after column"""
        m = np.asarray(obmat, np.float32).T.reshape(-1)
        body = self._id("OB", name) + struct.pack("<h3h", kind, 0, 0, 0) + struct.pack("<QQ", parent, data) + m.tobytes()
        ptr = self._add(b"OB\0\0", body, "Object")
        self.bases.append(ptr)
        return ptr

    def write(self, path):
        base_ptrs = []
        for _ in self.bases:
            base_ptrs.append(self.next_ptr)
            self.next_ptr += 0x1000
        for k, ob in enumerate(self.bases):
            nxt = base_ptrs[k + 1] if k + 1 < len(base_ptrs) else 0
            prv = base_ptrs[k - 1] if k else 0
            self.blocks.append((b"DATA", struct.pack("<QQQ", nxt, prv, ob), base_ptrs[k], self.struct_index["Base"], 1))
        first = base_ptrs[0] if base_ptrs else 0
        last = base_ptrs[-1] if base_ptrs else 0
        scene = self._id("SC", "Scene") + struct.pack("<QQQ", first, last, 0)
        self.blocks.insert(0, (b"SC\0\0", scene, self.next_ptr, self.struct_index["Scene"], 1))
        with open(path, "wb") as f:
            f.write(b"BLENDER-v274")
            for code, payload, ptr, sidx, count in self.blocks:
                f.write(code + struct.pack("<IQii", len(payload), ptr, sidx, count) + payload)
            f.write(b"DNA1" + struct.pack("<IQii", len(self.sdna), 0x7000, 0, 1) + self.sdna)
            f.write(b"ENDB" + struct.pack("<IQii", 0, 0, 0, 0))
        return path


def trs(t=(0, 0, 0), axis=(0, 0, 1), angle=0.0, s=(1, 1, 1)):
    a = np.asarray(axis, float)
    x, y, z = a / np.linalg.norm(a)
    c, si = np.cos(angle), np.sin(angle)
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * si, x * z * (1 - c) + y * si],
                  [y * x * (1 - c) + z * si, c + y * y * (1 - c), y * z * (1 - c) - x * si],
                  [z * x * (1 - c) - y * si, z * y * (1 - c) + x * si, c + z * z * (1 - c)]])
    m = np.eye(4)
    m[:3, :3] = r * np.asarray(s, float)[None, :]
    m[:3, 3] = t
    return m.astype(np.float32)
