"""TEST INFRASTRUCTURE: binding of oracle/_ref - the reference's own sources run on
the CPU (shader GLSL compiled as C++, polygonal_light.c / camera.c compiled as
they are).  Built by oracle/Makefile.ref when /root/reference exists; on the GPU
box only the prebuilt shared objects are present."""
import ctypes as C
import glob
import os

import numpy as np

from . import Frame, PSA_STATE_FLOATS, _f32, _fp

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def available():
    return os.path.exists(os.path.join(_REF, "libref_host.so")) and bool(glob.glob(os.path.join(_REF, "libref_shader_*.so")))


def variant_name(strategy=0, heuristic=0, technique="projected_solid_angle", lights=1, min_light_vertices=None,
                 max_light_vertices=3, samples=1, rays=False, show_lights=False, output_linear_rgb=True, error_display=0):
    lo = max_light_vertices if min_light_vertices is None else min_light_vertices
    name = "s%d_h%d_%s_L%d_V%d-%d_S%d_r%d_l%d_o%d" % (strategy, heuristic, technique, lights, lo, max_light_vertices,
                                                      samples, int(rays), int(show_lights), int(output_linear_rgb))
    return name + ("_e%d" % error_display if error_display else "")


def variants():
    return sorted(os.path.basename(p)[len("libref_shader_"):-3] for p in glob.glob(os.path.join(_REF, "libref_shader_*.so")))


_cache = {}


def shader(name):
    if name in _cache:
        return _cache[name]
    path = os.path.join(_REF, "libref_shader_%s.so" % name)
    if not os.path.exists(path):
        raise FileNotFoundError("reference shader variant %s has not been built (oracle/ref_stubs/build_ref_shaders.py)" % name)
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    L.ref_variant_matches.argtypes = [C.POINTER(Frame)]
    L.ref_shade_rows.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_uint32, C.c_uint32]
    L.ref_last_ray_count.restype = C.c_ulonglong
    L.ref_clip_polygon.restype = C.c_uint32
    L.ref_clip_polygon.argtypes = [C.c_uint32, fp]
    L.ref_psa_prepare.argtypes = [C.c_uint32, fp, fp]
    L.ref_psa_sample.argtypes = [fp, C.c_float, C.c_float, fp]
    L.ref_psa_error.argtypes = [fp, C.c_float, C.c_float, fp, fp]
    L.ref_solid_angle_sample.restype = C.c_float
    L.ref_solid_angle_sample.argtypes = [C.c_uint32, fp, fp, C.c_float, C.c_float, fp]
    L.ref_fast_positive_atan.restype = C.c_float
    L.ref_fast_positive_atan.argtypes = [C.c_float]
    L.ref_kahan.restype = C.c_float
    L.ref_kahan.argtypes = [C.c_float] * 4
    L.ref_decode_position.argtypes = [C.c_uint32, C.c_uint32, fp, fp, fp]
    L.ref_decode_normal.argtypes = [C.c_uint16, C.c_uint16, fp]
    L.ref_evaluate_brdf.argtypes = [fp, fp, C.c_int, C.c_int, fp]
    L.ref_srgb.argtypes = [C.c_float, fp, fp]
    _cache[name] = L
    return L


def shade(name, frame, y0=0, y1=None):
    L = shader(name)
    if not L.ref_variant_matches(C.byref(frame)):
        raise ValueError("frame settings do not match the compiled reference variant %s" % name)
    out = np.zeros((frame.height, frame.width, 4), np.float32)
    L.ref_shade_rows(C.byref(frame), out.ctypes.data, y0, frame.height if y1 is None else y1)
    return out


def capacity_of(name):
    """MAX_POLYGON_VERTEX_COUNT of a variant (max light vertices, +1 when the technique clips)."""
    parts = name.split("_")
    vmax = int([p for p in parts if p.startswith("V")][0].split("-")[1])
    clipped = not ("_solid_angle_" in name and "projected" not in name and "clipped" not in name)
    return vmax + (1 if clipped else 0)


def clip_polygon(name, vertices, count):
    cap = capacity_of(name)
    buf = np.zeros((cap, 3), np.float32)
    v = _f32(vertices)
    buf[:len(v)] = v
    if count < cap:
        buf[count] = v[0]
    n = shader(name).ref_clip_polygon(count, _fp(buf))
    return n, buf


def psa_prepare(name, vertices, count):
    cap = capacity_of(name)
    buf = np.zeros((cap, 3), np.float32)
    v = _f32(vertices)
    buf[:min(len(v), cap)] = v[:cap]
    state = np.zeros(PSA_STATE_FLOATS, np.float32)
    shader(name).ref_psa_prepare(count, _fp(buf), _fp(state))
    return state


def psa_sample(name, state, u0, u1):
    out = np.zeros(3, np.float32)
    shader(name).ref_psa_sample(_fp(state), u0, u1, _fp(out))
    return out


def psa_error(name, state, u0, u1, direction):
    out = np.zeros(3, np.float32)
    d = _f32(direction)
    shader(name).ref_psa_error(_fp(state), u0, u1, _fp(d), _fp(out))
    return out


# ---- host boundary (reference C sources compiled unmodified) ------------------------
_host = None


def host():
    global _host
    if _host is None:
        _host = C.CDLL(os.path.join(_REF, "libref_host.so"))
        _host.ref_wang_random_number.restype = C.c_uint32
        _host.ref_wang_random_number.argtypes = [C.c_uint32]
        _host.ref_half_to_float.restype = C.c_float
        _host.ref_half_to_float.argtypes = [C.c_uint16]
    return _host


# ---- experiment table and screenshot writers of the reference ------------------------
class _ExperimentView(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("scene_index", C.c_uint32), ("use_hdr", C.c_uint32),
                ("quick_save_path", C.c_char_p), ("screenshot_path", C.c_char_p),
                ("exposure_factor", C.c_float), ("roughness_factor", C.c_float),
                ("sample_count", C.c_uint32), ("sampling_strategies", C.c_uint32), ("mis_heuristic", C.c_uint32),
                ("mis_visibility_estimate", C.c_float), ("polygon_sampling_technique", C.c_uint32),
                ("error_display", C.c_uint32), ("error_min_exponent", C.c_float), ("noise_type", C.c_uint32),
                ("animate_noise", C.c_uint32), ("trace_shadow_rays", C.c_uint32), ("show_polygonal_lights", C.c_uint32),
                ("show_gui", C.c_uint32), ("v_sync", C.c_uint32)]


def experiments():
    """The reference's create_experiment_list() (src/experiment_list.c) as a list of dicts."""
    lib = host()
    lib.ref_experiment_count.restype = C.c_uint32
    lib.ref_experiment_get.argtypes = [C.c_uint32, C.POINTER(_ExperimentView)]
    out = []
    for i in range(lib.ref_experiment_count()):
        view = _ExperimentView()
        assert lib.ref_experiment_get(i, C.byref(view)) == 0
        entry = {}
        for name, _ in _ExperimentView._fields_:
            value = getattr(view, name)
            entry[name] = value.decode() if isinstance(value, bytes) else value
        out.append(entry)
    return out


def write_png(path, rgb8):
    """stbi_write_png as called at src/main.c:1731"""
    import numpy as np
    a = np.ascontiguousarray(rgb8, np.uint8)
    lib = host()
    lib.ref_write_png_rgb8.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    assert lib.ref_write_png_rgb8(path.encode(), a.shape[1], a.shape[0], a.ctypes.data) == 0


def write_hdr(path, rgb32f):
    """stbi_write_hdr as called at src/main.c:1752"""
    import numpy as np
    a = np.ascontiguousarray(rgb32f, np.float32)
    lib = host()
    lib.ref_write_hdr_rgb32f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    assert lib.ref_write_hdr_rgb32f(path.encode(), a.shape[1], a.shape[0], a.ctypes.data) == 0


def half_to_float_bits(half):
    import numpy as np
    return int(np.float32(host().ref_half_to_float(half)).view(np.uint32))
