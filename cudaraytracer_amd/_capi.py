"""ctypes bindings of the two product libraries.

libacr_hip.so  -- the C ABI of include/acr_hip.h (HIP kernels; the drop-in boundary)
libacr_host.so -- the C view of the C++ host side, include/acr_host.h

Both are built in-tree by cudaraytracer_amd/csrc/Makefile.  There is no
fallback: if a library is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_i32_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


class ImportNode(C.Structure):
    _fields_ = [("parent", C.c_int32), ("transform", C.c_float * 16), ("mesh_begin", C.c_uint32),
                ("mesh_count", C.c_uint32), ("name", C.c_char * 64)]


class ImportMesh(C.Structure):
    _fields_ = [("vertex_begin", C.c_uint32), ("vertex_count", C.c_uint32), ("face_begin", C.c_uint32),
                ("face_count", C.c_uint32), ("material_index", C.c_uint32), ("reserved", C.c_uint32)]


class ImportMaterial(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("ambient", C.c_float * 3), ("specular", C.c_float * 3),
                ("refracti", C.c_float)]


class ImportLight(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("type", C.c_int32), ("position", C.c_float * 3),
                ("direction", C.c_float * 3), ("ambient", C.c_float * 3), ("diffuse", C.c_float * 3),
                ("specular", C.c_float * 3), ("att_constant", C.c_float), ("att_linear", C.c_float),
                ("att_quadratic", C.c_float), ("inner_cone", C.c_float), ("outer_cone", C.c_float)]


class ImportCamera(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("aspect", C.c_float), ("hfov", C.c_float),
                ("position", C.c_float * 3), ("up", C.c_float * 3), ("look_at", C.c_float * 3)]


class ImportScene(C.Structure):
    _fields_ = [("num_nodes", C.c_uint32), ("nodes", C.POINTER(ImportNode)),
                ("num_node_mesh_ids", C.c_uint32), ("node_mesh_ids", c_u32_p),
                ("num_meshes", C.c_uint32), ("meshes", C.POINTER(ImportMesh)),
                ("num_vertices", C.c_uint32), ("positions", c_float_p), ("normals", c_float_p), ("colors", c_float_p),
                ("num_faces", C.c_uint32), ("indices", c_u32_p),
                ("num_materials", C.c_uint32), ("materials", C.POINTER(ImportMaterial)),
                ("num_lights", C.c_uint32), ("lights", C.POINTER(ImportLight)),
                ("camera", ImportCamera)]


class SceneSoA(C.Structure):
    _fields_ = [
        ("scene_depth", C.c_uint32), ("scene_min", C.c_float * 3), ("scene_max", C.c_float * 3),
        ("snode_start", c_u32_p), ("snode_end", c_u32_p), ("snode_left", c_float_p), ("snode_right", c_float_p),
        ("snode_flags", c_u8_p),
        ("num_objects", C.c_uint32),
        ("obj_transform", c_float_p), ("obj_inv_transform", c_float_p), ("obj_normal_transform", c_float_p),
        ("obj_inv_normal_transform", c_float_p), ("obj_world_min", c_float_p), ("obj_world_max", c_float_p),
        ("obj_mesh_begin", c_u32_p), ("obj_mesh_count", c_u32_p),
        ("num_obj_mesh_ids", C.c_uint32), ("obj_mesh_ids", c_u32_p),
        ("num_meshes", C.c_uint32), ("mesh_min", c_float_p), ("mesh_max", c_float_p), ("mesh_material", c_u32_p),
        ("mesh_depth", c_u32_p), ("mesh_node_begin", c_u32_p), ("mesh_tri_begin", c_u32_p),
        ("num_mesh_nodes", C.c_uint32), ("mnode_start", c_u32_p), ("mnode_end", c_u32_p), ("mnode_left", c_float_p),
        ("mnode_right", c_float_p), ("mnode_flags", c_u8_p),
        ("num_triangles", C.c_uint32), ("tri_p0", c_float_p), ("tri_p1", c_float_p), ("tri_p2", c_float_p),
        ("tri_n0", c_float_p), ("tri_n1", c_float_p), ("tri_n2", c_float_p), ("tri_prim_id", c_u32_p),
        ("num_materials", C.c_uint32), ("mat_ambient", c_float_p), ("mat_diffuse", c_float_p),
        ("mat_specular", c_float_p), ("mat_refractive_index", c_float_p),
        ("num_lights", C.c_uint32), ("light_type", c_i32_p), ("light_diffuse", c_float_p),
        ("light_position", c_float_p), ("light_direction", c_float_p), ("light_attenuation", c_float_p),
        ("light_cone", c_float_p),
    ]


class Camera(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("horizontal_fov", C.c_float), ("position", C.c_float * 3),
                ("forward", C.c_float * 3), ("up", C.c_float * 3)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("seed", C.c_uint64), ("frames", C.c_uint64),
                ("max_bounces", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("num_tiles", C.c_uint32), ("tile_ids", c_u32_p), ("pixel_values_dev", C.c_void_p),
                ("collect_stats", C.c_uint32), ("tile_rays_dev", C.c_void_p), ("tile_cycles_dev", C.c_void_p),
                ("path_keys_dev", C.c_void_p), ("block_order_dev", C.c_void_p), ("tiles_into_frame", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_bounce", C.c_uint64),
                ("primary_hits", C.c_uint64), ("kernel_ms", C.c_float), ("lds_bytes", C.c_uint32),
                ("scene_in_lds", C.c_uint32), ("grid_blocks", C.c_uint32), ("interval_traversal", C.c_uint32),
                ("six_plane_pixels", C.c_uint32), ("block_order", C.c_uint32), ("near_first", C.c_uint32), ("wave_tasks", C.c_uint32),
                ("shared_walk", C.c_uint32)]


# every symbol include/acr_hip.h declares: name -> (restype, argtypes)
HIP_SYMBOLS = {
    "acr_hip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "acr_hip_destroy": (None, [C.c_void_p]),
    "acr_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "acr_hip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "acr_hip_use_own_stream": (C.c_int, [C.c_void_p]),
    "acr_hip_sync": (C.c_int, [C.c_void_p]),
    "acr_hip_selftest_reciprocal": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "acr_hip_set_traversal": (C.c_int, [C.c_void_p, C.c_int]),
    "acr_hip_set_block_order": (C.c_int, [C.c_void_p, C.c_uint32]),
    "acr_hip_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneSoA)]),
    "acr_hip_set_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_hip_get_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_hip_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "acr_hip_render": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_void_p]),
    "acr_hip_render_static": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_void_p]),
    "acr_hip_trace_rays": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p, c_float_p, c_float_p, c_u8_p, c_float_p,
                                     c_u32_p, c_float_p, c_float_p]),
    "acr_hip_read_framebuffer": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t]),
    "acr_hip_read_primid": (C.c_int, [C.c_void_p, c_u32_p, C.c_size_t]),
    "acr_hip_read_hit_t": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t]),
    "acr_hip_read_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "acr_hip_framebuffer": (C.c_void_p, [C.c_void_p]),
    "acr_hip_pixel_map": (C.c_void_p, [C.c_void_p]),
    "acr_hip_path_keys": (C.c_void_p, [C.c_void_p]),
    "acr_hip_path_votes": (C.c_void_p, [C.c_void_p]),
    "acr_hip_vote_paths": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "acr_hip_vote_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "acr_hip_sort_pixel_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "acr_hip_scatter_tiles": (C.c_int, [C.c_void_p, C.c_void_p, c_u32_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.c_void_p]),
    "acr_hip_group_create": (C.c_int, [C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "acr_hip_group_destroy": (None, [C.c_void_p]),
    "acr_hip_group_last_error": (C.c_char_p, [C.c_void_p]),
    "acr_hip_group_size": (C.c_uint32, [C.c_void_p]),
    "acr_hip_group_context": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "acr_hip_group_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneSoA)]),
    "acr_hip_group_set_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_hip_group_get_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_hip_group_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "acr_hip_group_set_balance": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p]),
    "acr_hip_group_split": (C.c_int, [C.c_void_p, c_u32_p, c_u32_p]),
    "acr_hip_group_render": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_int]),
    "acr_hip_group_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "acr_hip_group_read_framebuffer": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t]),
    "acr_hip_group_read_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "acr_hip_group_sync": (C.c_int, [C.c_void_p]),
}

HOST_SYMBOLS = {
    "acr_host_import_blend": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "acr_host_import_soup": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    "acr_host_import_view": (C.POINTER(ImportScene), [C.c_void_p]),
    "acr_host_import_destroy": (None, [C.c_void_p]),
    "acr_host_scene_create": (C.c_void_p, [C.POINTER(ImportScene), C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "acr_host_scene_load": (C.c_void_p, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "acr_host_scene_destroy": (None, [C.c_void_p]),
    "acr_host_scene_soa": (C.c_int, [C.c_void_p, C.POINTER(SceneSoA)]),
    "acr_host_scene_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_host_scene_object_order": (C.c_int, [C.c_void_p, c_i32_p, C.c_uint32]),
    "acr_host_scene_face_order": (C.c_int, [C.c_void_p, C.c_uint32, c_u32_p, C.c_uint32]),
    "acr_host_scene_bih_text": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]),
    "acr_host_scene_bih_defined": (C.c_int, [C.c_void_p, C.c_uint32, c_u8_p, C.c_uint32]),
    "acr_host_renderer_create": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]),
    "acr_host_renderer_create_multi": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "acr_host_renderer_destroy": (None, [C.c_void_p]),
    "acr_host_renderer_error": (C.c_char_p, [C.c_void_p]),
    "acr_host_renderer_load_scene": (C.c_int, [C.c_void_p, C.c_void_p]),
    "acr_host_renderer_move_camera": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "acr_host_move_camera": (C.c_int, [C.POINTER(Camera), c_float_p, c_float_p]),
    "acr_host_renderer_set_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_host_renderer_get_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "acr_host_renderer_configure": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int64, C.c_uint32]),
    "acr_host_renderer_set_adaptive": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "acr_host_renderer_render": (C.c_int, [C.c_void_p]),
    "acr_host_renderer_frames_no_move": (C.c_uint64, [C.c_void_p]),
    "acr_host_renderer_read": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, c_u32_p, c_float_p]),
    "acr_host_renderer_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "acr_host_renderer_ctx": (C.c_void_p, [C.c_void_p]),
    "acr_host_renderer_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "acr_host_partition_tiles": (C.c_int, [c_u32_p, c_float_p, C.c_uint32, C.c_uint32, c_u32_p]),
    "acr_host_partition_tiles_weighted": (C.c_int, [c_u32_p, c_float_p, C.c_uint32, C.c_uint32, c_float_p, c_u32_p]),
    "acr_host_tile_order": (C.c_int, [C.c_uint32, C.c_uint32, c_u32_p]),
    "acr_host_kat": (C.c_int, [C.c_char_p, c_float_p, c_float_p]),
    "acr_host_fastersin": (C.c_float, [C.c_float]),
    "acr_host_fastercos": (C.c_float, [C.c_float]),
    "acr_host_fastertanfull": (C.c_float, [C.c_float]),
    "acr_host_rng_uniforms": (None, [C.c_uint64, C.c_uint32, c_float_p]),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None


def hip_lib():
    """libacr_hip.so with prototypes set.  Raises OSError if it was not built."""
    global _hip
    if _hip is None:
        path = os.path.join(_HERE, "libacr_hip.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make -C cudaraytracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no fallback path")
        _hip = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), HIP_SYMBOLS)
    return _hip


def host_lib():
    """libacr_host.so with prototypes set (loads libacr_hip.so first)."""
    global _host
    if _host is None:
        hip_lib()
        path = os.path.join(_HERE, "libacr_host.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make -C cudaraytracer_amd/csrc`")
        _host = _bind(C.CDLL(path), HOST_SYMBOLS)
    return _host
