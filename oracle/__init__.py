"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package.  The product (vulkan_renderer_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

STRATEGY = {"diffuse_only": 0, "diffuse_ggx_mis": 1, "diffuse_specular_separately": 2,
            "diffuse_specular_mis": 3, "diffuse_specular_random": 4}
MIS = {"balance": 0, "power": 1, "weighted": 2, "optimal_clamped": 3, "optimal": 4}
TECHNIQUE = {"baseline": 0, "area_turk": 1, "rectangle_solid_angle_urena": 2, "solid_angle_arvo": 3,
             "bilinear_cosine_warp_hart": 6, "bilinear_cosine_warp_clipping_hart": 7,
             "biquadratic_cosine_warp_hart": 8, "biquadratic_cosine_warp_clipping_hart": 9,
             "projected_solid_angle_arvo": 10, "solid_angle": 4, "clipped_solid_angle": 5, "projected_solid_angle": 11,
             "projected_solid_angle_biased": 12}
PSA_STATE_FLOATS = 49


def build(force=False):
    """Compiles liboracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("oracle_shading.c", "oracle_bvh.c", "oracle_libm.c", "oracle.h", "oracle_math.h", os.path.join("..", "vulkan_renderer_amd", "csrc", "glibc_math.h"))):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Frame(C.Structure):
    _fields_ = [
        ("constants", C.c_void_p),
        ("light_count", C.c_uint32), ("max_light_vertex_count", C.c_uint32),
        ("quantized_positions", C.c_void_p), ("normals_and_tex_coords", C.c_void_p),
        ("material_indices", C.c_void_p), ("triangle_count", C.c_uint64),
        ("material_constants", C.c_void_p), ("material_count", C.c_uint32),
        ("visibility", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
        ("ltc_rgba", C.c_void_p), ("ltc_rg", C.c_void_p),
        ("ltc_resolution", C.c_uint32), ("ltc_fresnel_count", C.c_uint32),
        ("noise", C.c_void_p), ("noise_width", C.c_uint32), ("noise_height", C.c_uint32), ("noise_depth", C.c_uint32),
        ("sampling_strategies", C.c_int32), ("mis_heuristic", C.c_int32), ("polygon_technique", C.c_int32),
        ("sample_count", C.c_uint32), ("trace_shadow_rays", C.c_int32), ("show_polygonal_lights", C.c_int32),
        ("bvh", C.c_void_p), ("brute_force_rays", C.c_int32),
        ("error_display", C.c_int32), ("error_index", C.c_int32),
        ("material_textures", C.c_void_p),
        ("light_textures", C.c_void_p), ("light_texture_count", C.c_uint32),
    ]


class Texture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("mip_count", C.c_uint32), ("srgb", C.c_uint32)]


class LightTexture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.oracle_shade_rows.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_last_ray_count.restype = C.c_uint64
        L.oracle_set_math_mode.argtypes = [C.c_int]
        L.oracle_encode_srgb8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.oracle_encode_half_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]
        L.oracle_bvh_build.restype = C.c_void_p
        L.oracle_bvh_build.argtypes = [C.c_void_p, C.c_uint64, fp, fp]
        L.oracle_bvh_destroy.argtypes = [C.c_void_p]
        L.oracle_bvh_any_hit.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, C.c_int]
        L.oracle_primary_visibility.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.oracle_clip_polygon.restype = C.c_uint32
        L.oracle_clip_polygon.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, fp]
        L.oracle_psa_prepare.argtypes = [C.c_uint32, C.c_uint32, fp, fp]
        L.oracle_psa_sample.argtypes = [fp, C.c_uint32, C.c_float, C.c_float, C.c_int, fp]
        L.oracle_psa_error.argtypes = [fp, C.c_uint32, C.c_float, C.c_float, fp, fp]
        L.oracle_solid_angle_sample.restype = C.c_float
        L.oracle_solid_angle_sample.argtypes = [C.c_uint32, C.c_uint32, fp, fp, C.c_float, C.c_float, fp]
        for name in ("oracle_atan", "oracle_acos_unit", "oracle_fast_positive_atan", "oracle_rsqrt", "oracle_log2"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.oracle_sincos.argtypes = [C.c_float, fp, fp]
        L.oracle_kahan.restype = C.c_float
        L.oracle_kahan.argtypes = [C.c_float] * 4
        L.oracle_decode_position.argtypes = [C.c_uint32, C.c_uint32, fp, fp, fp]
        L.oracle_decode_normal.argtypes = [C.c_uint16, C.c_uint16, fp]
        L.oracle_ltc_coefficients.argtypes = [C.POINTER(Frame), C.c_float, C.c_float, fp, fp, fp, fp]
        L.oracle_shading_data.argtypes = [C.POINTER(Frame), C.c_uint32, C.c_uint32, fp]
        L.oracle_evaluate_brdf.argtypes = [fp, fp, C.c_int, C.c_int, fp]
        L.oracle_noise_stream.argtypes = [C.POINTER(Frame), C.c_uint32, C.c_uint32, C.c_uint32, fp]
        L.oracle_libm_evaluate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_libm_count_mismatches.restype = C.c_size_t
        L.oracle_libm_count_mismatches.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, C.c_float, C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_math_mode(mode):
    lib().oracle_set_math_mode(int(mode))


def set_libm_source(source):
    """"port" (default): the oracle's transcendentals are the restatement of glibc 2.35 in
    vulkan_renderer_amd/csrc/glibc_math.h - identical frames on every machine; "system": this machine's
    C library (what the compiled reference shader under oracle/_ref calls)."""
    lib().oracle_set_libm_source({"port": 0, "system": 1}[source])


def libm_description():
    if lib().oracle_get_libm_source():
        return "transcendentals by the C library of this machine"
    return ("transcendentals by the restatement of glibc 2.35's float functions (x86-64 FMA / AVX2 IFUNC variants; csrc/glibc_math.h), "
            "equal to that library for all 2^32 arguments of every one-argument function (profiles/r03a/glibc_math_exhaustive.txt, tests/test_glibc_math.py)")


# operation codes of oracle_libm.c == evaluate_device_arithmetic (include/vkr_shading_pass.h)
LIBM_OPERATIONS = {"atan": 5, "acos": 6, "sin": 7, "cos": 8, "log2": 9, "pow": 10, "atan2": 11, "inverse_sqrt": 12, "atan_rows": 13}


def libm_evaluate(operation, a, b=None, port=False):
    """f(a, b) elementwise by this machine's C library (or, port=True, by the restatement of glibc
    2.35 in vulkan_renderer_amd/csrc/glibc_math.h compiled for the host)"""
    a = np.ascontiguousarray(a, np.float32)
    b = None if b is None else np.ascontiguousarray(b, np.float32)
    out = np.zeros_like(a)
    lib().oracle_libm_evaluate(LIBM_OPERATIONS[operation], int(port), a.ctypes.data, None if b is None else b.ctypes.data, out.ctypes.data, a.size)
    return out


def libm_count_mismatches(operation, first_bits, stride, count, second_argument=0.0):
    """-> (arguments for which the restatement and the C library differ, bits of one of them)"""
    bad = C.c_uint32(0)
    n = lib().oracle_libm_count_mismatches(LIBM_OPERATIONS[operation], first_bits, stride, count, second_argument, C.byref(bad))
    return int(n), int(bad.value)


def clip_polygon(vertices, max_count=None, min_count=3):
    """vertices: (n,3). Returns the clipped polygon (count,3) incl. nothing else."""
    v = _f32(vertices)
    n = v.shape[0]
    cap = max_count or n + 1
    buf = np.zeros((cap, 3), np.float32)
    buf[:n] = v
    if n < cap:
        buf[n] = v[0]
    count = lib().oracle_clip_polygon(n, min_count, cap, _fp(buf))
    return count, buf


def psa_prepare(vertices, count=None, max_count=None):
    v = _f32(vertices)
    n = count if count is not None else v.shape[0]
    cap = max_count or max(n + 1, 4)
    buf = np.zeros((9, 3), np.float32)
    buf[:v.shape[0]] = v
    if n < cap and v.shape[0] <= n:
        buf[n] = v[0]
    state = np.zeros(PSA_STATE_FLOATS, np.float32)
    lib().oracle_psa_prepare(n, cap, _fp(buf), _fp(state))
    return state, cap


def psa_sample(state, cap, u0, u1, biased=False):
    out = np.zeros(3, np.float32)
    lib().oracle_psa_sample(_fp(state), cap, float(u0), float(u1), int(biased), _fp(out))
    return out


def psa_error(state, cap, u0, u1, direction):
    out = np.zeros(3, np.float32)
    d = _f32(direction)
    lib().oracle_psa_error(_fp(state), cap, float(u0), float(u1), _fp(d), _fp(out))
    return out


def solid_angle_sample(vertices, shading_position, u0, u1, max_count=None):
    v = _f32(vertices)
    n = v.shape[0]
    cap = max_count or n
    buf = np.zeros((9, 3), np.float32)
    buf[:n] = v
    if n < cap:
        buf[n] = v[0]
    out = np.zeros(3, np.float32)
    sp = _f32(shading_position)
    sa = lib().oracle_solid_angle_sample(n, cap, _fp(buf), _fp(sp), float(u0), float(u1), _fp(out))
    return sa, out


class Bvh:
    def __init__(self, quantized_positions, factor, summand):
        self._q = np.ascontiguousarray(quantized_positions, dtype=np.uint32)
        f, s = _f32(factor), _f32(summand)
        self.handle = lib().oracle_bvh_build(self._q.ctypes.data, self._q.size // 6, _fp(f), _fp(s))

    def any_hit(self, origin, direction, t_min, t_max, brute_force=False):
        o, d = _f32(origin), _f32(direction)
        return bool(lib().oracle_bvh_any_hit(self.handle, _fp(o), _fp(d), t_min, t_max, int(brute_force)))

    def __del__(self):
        if getattr(self, "handle", None):
            lib().oracle_bvh_destroy(self.handle)
            self.handle = None


def make_frame(inputs, settings, bvh=None):
    """inputs: dict of numpy arrays as produced by vulkan_renderer_amd.synthetic /
    the product's host library (byte-identical to the GPU uploads).
    settings: dict with sampling_strategies, mis_heuristic, polygon_technique (names
    or ints), sample_count, trace_shadow_rays, show_polygonal_lights."""
    def enum(table, v):
        return table[v] if isinstance(v, str) else int(v)
    keep = {}
    f = Frame()
    def ptr(name, dtype):
        a = np.ascontiguousarray(inputs[name], dtype=dtype)
        keep[name] = a
        return a.ctypes.data
    f.constants = ptr("constants", np.uint8)
    f.light_count = int(inputs["light_count"])
    f.max_light_vertex_count = int(inputs["max_light_vertex_count"])
    f.quantized_positions = ptr("quantized_positions", np.uint32)
    f.normals_and_tex_coords = ptr("normals_and_tex_coords", np.uint16)
    f.material_indices = ptr("material_indices", np.uint8)
    f.triangle_count = keep["material_indices"].size
    f.material_constants = ptr("material_constants", np.float32)
    f.material_count = keep["material_constants"].size // 8
    f.visibility = ptr("visibility", np.uint32)
    f.height, f.width = inputs["visibility"].shape
    f.ltc_rgba = ptr("ltc_rgba", np.uint16)
    f.ltc_rg = ptr("ltc_rg", np.uint16)
    f.ltc_fresnel_count, f.ltc_resolution = inputs["ltc_rgba"].shape[0], inputs["ltc_rgba"].shape[1]
    f.noise = ptr("noise", np.uint16)
    f.noise_depth, f.noise_height, f.noise_width = inputs["noise"].shape[:3]
    f.sampling_strategies = enum(STRATEGY, settings.get("sampling_strategies", "diffuse_specular_mis"))
    f.mis_heuristic = enum(MIS, settings.get("mis_heuristic", "optimal_clamped"))
    f.polygon_technique = enum(TECHNIQUE, settings.get("polygon_technique", "projected_solid_angle"))
    f.sample_count = int(settings.get("sample_count", 1))
    f.trace_shadow_rays = int(bool(settings.get("trace_shadow_rays", False)))
    f.show_polygonal_lights = int(bool(settings.get("show_polygonal_lights", False)))
    f.bvh = bvh.handle if bvh is not None else None
    f.brute_force_rays = int(bool(settings.get("brute_force_rays", False)))
    # error_display_t of the reference (main.h:93-118): 1..3 diffuse, 4..6 specular
    display = int(settings.get("error_display", 0))
    f.error_display = 0 if display == 0 else (1 if display <= 3 else 2)
    f.error_index = (display - 1) % 3 if display else 0
    # material textures: list of dicts {"texels": uint8 array of all mips (RGBA8), "width", "height", "mip_count", "srgb"},
    # three per material (base colour, specular, normal)
    textures = inputs.get("material_textures")
    if textures:
        array = (Texture * len(textures))()
        for index, (t, entry) in enumerate(zip(array, textures)):
            texels = np.ascontiguousarray(entry["texels"], np.uint8)
            keep["texture_texels_%d" % index] = texels
            t.texels = texels.ctypes.data
            t.width, t.height, t.mip_count, t.srgb = int(entry["width"]), int(entry["height"]), int(entry["mip_count"]), int(entry["srgb"])
        keep["texture_array"] = array
        f.material_textures = C.cast(array, C.c_void_p)
    # light textures: list of float32 arrays (height, width, 4), level 0 only; None = white
    textures = inputs.get("light_textures")
    if textures:
        array = (LightTexture * len(textures))()
        for index, (t, entry) in enumerate(zip(array, textures)):
            if entry is None:
                continue
            texels = np.ascontiguousarray(entry, np.float32)
            keep["light_texels_%d" % index] = texels
            t.texels = texels.ctypes.data
            t.height, t.width = texels.shape[:2]
        keep["light_texture_array"] = array
        f.light_textures = C.cast(array, C.c_void_p)
        f.light_texture_count = len(textures)
    f._keep = keep
    f._bvh = bvh
    return f


def shade(frame, y0=0, y1=None, threads=0):
    """Returns (height, width, 4) float32; rows outside [y0, y1) stay zero."""
    out = np.zeros((frame.height, frame.width, 4), np.float32)
    lib().oracle_shade_rows(C.byref(frame), out.ctypes.data, y0, frame.height if y1 is None else y1, threads)
    return out


def last_ray_count():
    return int(lib().oracle_last_ray_count())


def encode_srgb8(rgba):
    a = _f32(rgba)
    out = np.zeros(a.shape[:-1] + (4,), np.uint8)
    lib().oracle_encode_srgb8(a.ctypes.data, out.ctypes.data, a.size // 4)
    return out


def encode_half_bits(rgba, frame_bits, output_linear_rgb=False):
    a = _f32(rgba)
    out = np.zeros(a.shape[:-1] + (4,), np.uint8)
    lib().oracle_encode_half_bits(a.ctypes.data, out.ctypes.data, a.size // 4, frame_bits, int(output_linear_rgb))
    return out


def primary_visibility(constants, bvh, width, height, near, far):
    c = np.ascontiguousarray(constants, dtype=np.uint8)
    out = np.zeros((height, width), np.uint32)
    lib().oracle_primary_visibility(c.ctypes.data, bvh.handle, width, height, near, far, out.ctypes.data)
    return out
