"""Import-level scenes on the Python side: an owning container that packs into
the C `acr_import_scene` struct, and a JSON form used for the committed scene
fixtures (floats are written as the shortest decimal that reads back to the
same binary32 value, so a fixture round-trips bit-exactly)."""
import ctypes as C
import json

import numpy as np

from . import _capi


def _name(s):
    return s.encode()[:63]


class ImportData:
    def __init__(self):
        self.nodes = []      # dict(name, parent, transform[16], meshes[list of int])
        self.meshes = []     # dict(positions (n,3) f32, normals (n,3) f32, indices (f,3) u32, material int)
        self.materials = []  # dict(diffuse[3], ambient[3], specular[3], refracti)
        self.lights = []     # dict(name, type, position, direction, ambient, diffuse, specular, att[3], cones[2])
        self.camera = dict(name="", aspect=0.0, hfov=0.0, position=[0, 0, 0], up=[0, 0, 0], look_at=[0, 0, 0])

    # ---- C struct <-> container ------------------------------------------------
    @staticmethod
    def from_view(view):
        """Copies an `acr_import_scene` (ctypes struct) into a container."""
        d = ImportData()
        ids = np.ctypeslib.as_array(view.node_mesh_ids, (max(view.num_node_mesh_ids, 1),))[:view.num_node_mesh_ids]
        for i in range(view.num_nodes):
            n = view.nodes[i]
            d.nodes.append(dict(name=n.name.decode(), parent=int(n.parent), transform=[float(x) for x in n.transform],
                                meshes=[int(x) for x in ids[n.mesh_begin:n.mesh_begin + n.mesh_count]]))
        nv, nf = view.num_vertices, view.num_faces
        pos = np.ctypeslib.as_array(view.positions, (max(nv, 1) * 3,))[:nv * 3].reshape(-1, 3)
        nrm = np.ctypeslib.as_array(view.normals, (max(nv, 1) * 3,))[:nv * 3].reshape(-1, 3)
        idx = np.ctypeslib.as_array(view.indices, (max(nf, 1) * 3,))[:nf * 3].reshape(-1, 3)
        for i in range(view.num_meshes):
            m = view.meshes[i]
            d.meshes.append(dict(positions=pos[m.vertex_begin:m.vertex_begin + m.vertex_count].copy(),
                                 normals=nrm[m.vertex_begin:m.vertex_begin + m.vertex_count].copy(),
                                 indices=idx[m.face_begin:m.face_begin + m.face_count].copy(),
                                 material=int(m.material_index)))
        for i in range(view.num_materials):
            m = view.materials[i]
            d.materials.append(dict(diffuse=list(m.diffuse), ambient=list(m.ambient), specular=list(m.specular),
                                    refracti=float(m.refracti)))
        for i in range(view.num_lights):
            l = view.lights[i]
            d.lights.append(dict(name=l.name.decode(), type=int(l.type), position=list(l.position),
                                 direction=list(l.direction), ambient=list(l.ambient), diffuse=list(l.diffuse),
                                 specular=list(l.specular), att=[l.att_constant, l.att_linear, l.att_quadratic],
                                 cones=[l.inner_cone, l.outer_cone]))
        c = view.camera
        d.camera = dict(name=c.name.decode(), aspect=float(c.aspect), hfov=float(c.hfov), position=list(c.position),
                        up=list(c.up), look_at=list(c.look_at))
        return d

    def pack(self):
        """Returns (acr_import_scene struct, keepalive list)."""
        keep = []
        v = _capi.ImportScene()
        nodes = (_capi.ImportNode * max(len(self.nodes), 1))()
        ids = []
        for i, n in enumerate(self.nodes):
            nodes[i].parent = n["parent"]
            nodes[i].transform = (C.c_float * 16)(*[np.float32(x) for x in n["transform"]])
            nodes[i].mesh_begin = len(ids)
            nodes[i].mesh_count = len(n["meshes"])
            nodes[i].name = _name(n["name"])
            ids.extend(n["meshes"])
        ids_a = np.asarray(ids if ids else [0], dtype=np.uint32)
        meshes = (_capi.ImportMesh * max(len(self.meshes), 1))()
        pos, nrm, idx = [], [], []
        vb = fb = 0
        for i, m in enumerate(self.meshes):
            p = np.ascontiguousarray(m["positions"], dtype=np.float32).reshape(-1, 3)
            q = np.ascontiguousarray(m["normals"], dtype=np.float32).reshape(-1, 3)
            f = np.ascontiguousarray(m["indices"], dtype=np.uint32).reshape(-1, 3)
            meshes[i].vertex_begin, meshes[i].vertex_count = vb, len(p)
            meshes[i].face_begin, meshes[i].face_count = fb, len(f)
            meshes[i].material_index = m["material"]
            vb += len(p)
            fb += len(f)
            pos.append(p); nrm.append(q); idx.append(f)
        pos_a = np.ascontiguousarray(np.concatenate(pos) if pos else np.zeros((1, 3)), dtype=np.float32)
        nrm_a = np.ascontiguousarray(np.concatenate(nrm) if nrm else np.zeros((1, 3)), dtype=np.float32)
        idx_a = np.ascontiguousarray(np.concatenate(idx) if idx else np.zeros((1, 3)), dtype=np.uint32)
        mats = (_capi.ImportMaterial * max(len(self.materials), 1))()
        for i, m in enumerate(self.materials):
            mats[i].diffuse = (C.c_float * 3)(*m["diffuse"])
            mats[i].ambient = (C.c_float * 3)(*m["ambient"])
            mats[i].specular = (C.c_float * 3)(*m["specular"])
            mats[i].refracti = m["refracti"]
        lights = (_capi.ImportLight * max(len(self.lights), 1))()
        for i, l in enumerate(self.lights):
            lights[i].name = _name(l["name"])
            lights[i].type = l["type"]
            for k in ("position", "direction", "ambient", "diffuse", "specular"):
                setattr(lights[i], k, (C.c_float * 3)(*l[k]))
            lights[i].att_constant, lights[i].att_linear, lights[i].att_quadratic = l["att"]
            lights[i].inner_cone, lights[i].outer_cone = l["cones"]
        v.num_nodes, v.nodes = len(self.nodes), nodes
        v.num_node_mesh_ids, v.node_mesh_ids = len(ids), ids_a.ctypes.data_as(_capi.c_u32_p)
        v.num_meshes, v.meshes = len(self.meshes), meshes
        v.num_vertices = vb
        v.positions = pos_a.ctypes.data_as(_capi.c_float_p)
        v.normals = nrm_a.ctypes.data_as(_capi.c_float_p)
        v.colors = None
        v.num_faces, v.indices = fb, idx_a.ctypes.data_as(_capi.c_u32_p)
        v.num_materials, v.materials = len(self.materials), mats
        v.num_lights, v.lights = len(self.lights), lights
        c = self.camera
        v.camera.name = _name(c["name"])
        v.camera.aspect, v.camera.hfov = c["aspect"], c["hfov"]
        v.camera.position = (C.c_float * 3)(*c["position"])
        v.camera.up = (C.c_float * 3)(*c["up"])
        v.camera.look_at = (C.c_float * 3)(*c["look_at"])
        keep.extend([nodes, ids_a, meshes, pos_a, nrm_a, idx_a, mats, lights])
        return v, keep

    # ---- JSON fixtures ----------------------------------------------------------
    def to_json(self):
        def fl(a):
            return [float(np.float32(x)) for x in np.asarray(a, dtype=np.float32).ravel()]
        out = dict(
            format="acr-import-scene/1",
            nodes=[dict(name=n["name"], parent=n["parent"], transform=fl(n["transform"]), meshes=list(n["meshes"]))
                   for n in self.nodes],
            meshes=[dict(material=m["material"], positions=fl(m["positions"]), normals=fl(m["normals"]),
                         indices=[int(x) for x in np.asarray(m["indices"]).ravel()]) for m in self.meshes],
            materials=[dict(diffuse=fl(m["diffuse"]), ambient=fl(m["ambient"]), specular=fl(m["specular"]),
                            refracti=float(np.float32(m["refracti"]))) for m in self.materials],
            lights=[dict(name=l["name"], type=l["type"], position=fl(l["position"]), direction=fl(l["direction"]),
                         ambient=fl(l["ambient"]), diffuse=fl(l["diffuse"]), specular=fl(l["specular"]),
                         att=fl(l["att"]), cones=fl(l["cones"])) for l in self.lights],
            camera=dict(name=self.camera["name"], aspect=float(np.float32(self.camera["aspect"])),
                        hfov=float(np.float32(self.camera["hfov"])), position=fl(self.camera["position"]),
                        up=fl(self.camera["up"]), look_at=fl(self.camera["look_at"])),
        )
        return json.dumps(out, indent=1)

    @staticmethod
    def from_json(text):
        j = json.loads(text)
        if j.get("format") != "acr-import-scene/1":
            raise ValueError("not an acr import-scene fixture")
        d = ImportData()
        d.nodes = j["nodes"]
        for m in j["meshes"]:
            d.meshes.append(dict(material=m["material"],
                                 positions=np.asarray(m["positions"], dtype=np.float32).reshape(-1, 3),
                                 normals=np.asarray(m["normals"], dtype=np.float32).reshape(-1, 3),
                                 indices=np.asarray(m["indices"], dtype=np.uint32).reshape(-1, 3)))
        d.materials = j["materials"]
        d.lights = j["lights"]
        d.camera = j["camera"]
        return d

    @staticmethod
    def load(path):
        with open(path) as f:
            return ImportData.from_json(f.read())


def load_blend(path):
    """Reads a .blend file through the C++ loader and returns an ImportData."""
    lib = _capi.host_lib()
    err = C.create_string_buffer(512)
    h = lib.acr_host_import_blend(str(path).encode(), err, len(err))
    if not h:
        raise RuntimeError(f"cannot import {path}: {err.value.decode()}")
    try:
        return ImportData.from_view(lib.acr_host_import_view(h).contents)
    finally:
        lib.acr_host_import_destroy(h)
