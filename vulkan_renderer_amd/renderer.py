"""Host-side driver of the shading pass, written against the C-ABI only.

It plays the role of the reference's startup_application / render_frame
(src/main.c:1896, :2197) for tests and bench.py: load the scene, LTC table and
noise table, specify lights and settings, create the pass, render, read back.
All arithmetic happens inside libvkr_shading.so; if the library or a GPU is
missing, construction fails loudly - there is no CPU fallback."""
import ctypes as C
import math

import numpy as np

from . import capi

STRATEGY = {"diffuse_only": 0, "diffuse_ggx_mis": 1, "diffuse_specular_separately": 2,
            "diffuse_specular_mis": 3, "diffuse_specular_random": 4}
MIS = {"balance": 0, "power": 1, "weighted": 2, "optimal_clamped": 3, "optimal": 4}
TECHNIQUE = {"baseline": 0, "area_turk": 1, "rectangle_solid_angle_urena": 2, "solid_angle_arvo": 3,
             "bilinear_cosine_warp_hart": 6, "bilinear_cosine_warp_clipping_hart": 7,
             "biquadratic_cosine_warp_hart": 8, "biquadratic_cosine_warp_clipping_hart": 9,
             "projected_solid_angle_arvo": 10, "solid_angle": 4, "clipped_solid_angle": 5, "projected_solid_angle": 11,
             "projected_solid_angle_biased": 12}
NOISE = {"white": 0, "blue": 1, "ahmed": 2}
# arithmetic_mode_t; and the math mode of the CPU oracle that evaluates the same operations
ARITHMETIC_MODES = {"libm": 0, "fast": 1, "exact": 2}
ORACLE_MATH_MODE = {"libm": 0, "fast": 0, "exact": 1}
# acceleration_structure_builder_t (include/vkr_scene.h); True selects the default (HIP kernels, binned SAH)
BVH_BUILDER = {False: 0, None: 0, True: 1, "sah_device": 1, "lbvh_device": 2, "sah_host": 3}
BVH_BUILDER_NAME = {0: "none", 1: "binned SAH built by HIP kernels", 2: "Morton-code LBVH built by HIP kernels", 3: "binned SAH built on the host"}


def _enum(table, value):
    return table[value] if isinstance(value, str) else int(value)


class HostScene:
    """Device-less use of the loader surface (parsing, quantisation, constants)."""

    def __init__(self):
        self.lib = capi.load()
        self.app = capi.Application()
        self._device = None
        self._lights_keepalive = None

    # -- loaders -------------------------------------------------------------------
    def _dev(self):
        return C.byref(self.app.device) if self._device else None

    def load_scene(self, path, texture_path=None, acceleration_structure=False):
        rc = self.lib.load_scene(C.byref(self.app.scene), self._dev(), path.encode(),
                                 texture_path.encode() if texture_path else None, BVH_BUILDER[acceleration_structure])
        if rc:
            raise RuntimeError("load_scene failed for %s" % path)

    def load_ltc_table(self, directory, fresnel_count=51):
        if self.lib.load_ltc_table(C.byref(self.app.ltc_table), self._dev(), directory.encode(), fresnel_count):
            raise RuntimeError("load_ltc_table failed for %s" % directory)

    def load_noise_table(self, noise_type="white", resolution=None):
        t = _enum(NOISE, noise_type)
        res = self.lib.get_default_noise_resolution(t)
        if resolution is not None:
            res = capi.Extent3D(*resolution)
        if self.lib.load_noise_table(C.byref(self.app.noise_table), self._dev(), res, t):
            raise RuntimeError("load_noise_table failed")

    # -- scene specification ---------------------------------------------------------
    def set_camera(self, position, rotation_x, rotation_z, vertical_fov, near=0.05, far=1.0e3):
        cam = self.app.scene_specification.camera
        cam.position_world_space[:] = position
        cam.rotation_x, cam.rotation_z, cam.vertical_fov = rotation_x, rotation_z, vertical_fov
        cam.near, cam.far, cam.speed = near, far, 2.0

    def set_lights(self, lights):
        """lights: list of dicts from synthetic.light_spec()."""
        spec = self.app.scene_specification
        for i in range(spec.polygonal_light_count):
            self.lib.destroy_polygonal_light(C.byref(spec.polygonal_lights[i]))
        array = (capi.PolygonalLight * max(len(lights), 1))()
        for i, l in enumerate(lights):
            light = array[i]
            light.rotation_angles[:] = l["rotation_angles"]
            light.translation[:] = l["translation"]
            light.radiant_flux[:] = l["radiant_flux"]
            light.scaling_x, light.scaling_y = l["scaling"]
            v = np.asarray(l["vertices_plane_space"], np.float32)
            self.lib.set_polygonal_light_vertex_count(C.byref(light), len(v))
            for j in range(len(v)):
                light.vertices_plane_space[4 * j + 0] = float(v[j, 0])
                light.vertices_plane_space[4 * j + 1] = float(v[j, 1])
            light.texturing_technique = {"none": 0, "area": 1, "portal": 2, "ies_profile": 3}.get(l.get("texturing_technique", 0), l.get("texturing_technique", 0))
            if l.get("texture_file_path"):
                # the library frees the path with free(): hand it a malloc'ed copy
                encoded = l["texture_file_path"].encode() + b"\0"
                libc = C.CDLL(None)
                libc.malloc.restype = C.c_void_p
                copy = libc.malloc(len(encoded))
                C.memmove(copy, encoded, len(encoded))
                light.texture_file_path = copy
            self.lib.update_polygonal_light(C.byref(light))
        self._lights_keepalive = array
        spec.polygonal_lights = C.cast(array, C.POINTER(capi.PolygonalLight))
        spec.polygonal_light_count = len(lights)
        # like the reference's update rules (main.c:1831, 1854, 1879): new lights, new light textures
        if self.app.light_textures.texture_count:
            self.lib.destroy_light_textures(C.byref(self.app.light_textures), self._dev())
        if any(l.get("texturing_technique") for l in lights):
            self.create_light_textures()

    def create_light_textures(self):
        if self.lib.create_and_assign_light_textures(C.byref(self.app.light_textures), self._dev(), C.byref(self.app.scene_specification)):
            raise RuntimeError("create_and_assign_light_textures failed")

    def set_settings(self, **kw):
        s = self.app.render_settings
        if "exposure_factor" not in kw and s.exposure_factor == 0.0:
            self.lib.specify_default_render_settings(C.byref(s))
            s.show_polygonal_lights = 0
            s.animate_noise = 0
            s.noise_type = 0
        for key, value in kw.items():
            if key == "sampling_strategies":
                s.sampling_strategies = _enum(STRATEGY, value)
            elif key == "mis_heuristic":
                s.mis_heuristic = _enum(MIS, value)
            elif key in ("polygon_technique", "polygon_sampling_technique"):
                s.polygon_sampling_technique = _enum(TECHNIQUE, value)
            elif key in ("width", "height"):
                pass
            elif key in ("trace_shadow_rays", "show_polygonal_lights", "animate_noise"):
                setattr(s, key, int(bool(value)))
            else:
                setattr(s, key, value)
        if "width" in kw:
            self.app.swapchain.extent.width = kw["width"]
        if "height" in kw:
            self.app.swapchain.extent.height = kw["height"]

    # -- constants --------------------------------------------------------------------
    def constants(self):
        """Byte image of write_constants() as a numpy array."""
        size = self.lib.get_constant_buffer_size(C.byref(self.app))
        buf = np.zeros(size, np.uint8)
        self.lib.write_constants(buf.ctypes.data, C.byref(self.app))
        return buf

    def max_light_vertex_count(self):
        return self.lib.get_max_polygonal_light_vertex_count(C.byref(self.app.scene_specification))

    # -- host views of the loaded data (inputs for the CPU oracle in tests) -------------
    def host_inputs(self, visibility=None):
        app = self.app
        T = app.scene.mesh.triangle_count
        ltc = app.ltc_table
        n = app.noise_table.resolution
        layers, res = ltc.fresnel_count, ltc.roughness_count
        inputs = {
            "constants": self.constants(),
            "light_count": app.scene_specification.polygonal_light_count,
            "max_light_vertex_count": self.max_light_vertex_count(),
            "quantized_positions": np.ctypeslib.as_array(app.scene.mesh.host_positions, (T * 3, 2)).copy(),
            "normals_and_tex_coords": np.ctypeslib.as_array(app.scene.mesh.host_normals_and_tex_coords, (T * 3, 4)).copy(),
            "material_indices": np.ctypeslib.as_array(app.scene.mesh.host_material_indices, (T,)).copy(),
            "material_constants": np.ctypeslib.as_array(app.scene.materials.host_constants, (app.scene.materials.material_count * 8,)).copy(),
            "ltc_rgba": np.ctypeslib.as_array(ltc.host_rgba, (layers, res, res, 4)).copy(),
            "ltc_rg": np.ctypeslib.as_array(ltc.host_rg, (layers, res, res, 2)).copy(),
            "noise": np.ctypeslib.as_array(app.noise_table.host_data, (n.depth, n.height, n.width, 4)).copy(),
            "dequantization_factor": np.array(app.scene.mesh.dequantization_factor[:], np.float32),
            "dequantization_summand": np.array(app.scene.mesh.dequantization_summand[:], np.float32),
        }
        materials = app.scene.materials
        if materials.textured:
            descriptors = np.ctypeslib.as_array(materials.host_texture_descriptors, (materials.material_count * 3, 4)).copy()
            texels = np.ctypeslib.as_array(materials.host_texels, (materials.texel_count, 4)).copy()
            textures = []
            for first, width, height, packed in descriptors:
                count = 0
                w, h = int(width), int(height)
                for _ in range(int(packed) & 0xFFFF):
                    count += w * h
                    w, h = max(w // 2, 1), max(h // 2, 1)
                textures.append({"texels": texels[int(first):int(first) + count] if width else np.zeros((1, 4), np.uint8),
                                 "width": int(width), "height": int(height), "mip_count": int(packed) & 0xFFFF, "srgb": int(packed) >> 16})
            inputs["material_textures"] = textures
        if app.light_textures.texture_count:
            inputs["light_textures"] = self.light_texture_arrays()
        if visibility is not None:
            inputs["visibility"] = np.ascontiguousarray(visibility, np.uint32)
        return inputs

    def light_texture_arrays(self):
        """The loaded light textures as float32 arrays (height, width, 4); None for white."""
        lights = self.app.light_textures
        textures = []
        for i in range(lights.texture_count):
            first, width, height, _ = (int(v) for v in lights.host_descriptors[i])
            textures.append(np.ctypeslib.as_array(lights.host_texels, (lights.texel_count * 4,))[4 * first:4 * (first + width * height)].reshape(height, width, 4).copy() if width else None)
        return textures

    def oracle_settings(self):
        s = self.app.render_settings
        return {"sampling_strategies": s.sampling_strategies, "mis_heuristic": s.mis_heuristic,
                "polygon_technique": s.polygon_sampling_technique, "sample_count": s.sample_count,
                "trace_shadow_rays": bool(s.trace_shadow_rays), "show_polygonal_lights": bool(s.show_polygonal_lights),
                "error_display": int(s.error_display)}

    def close(self):
        app = self.app
        dev = self._dev()
        if getattr(self, "exchange", None) is not None:
            self.destroy_exchange()
        if app.shading_pass.constants_device:
            self.lib.destroy_shading_pass(C.byref(app.shading_pass), dev)
        if app.render_targets.radiance:
            self.lib.destroy_render_targets(C.byref(app.render_targets), dev)
        self.lib.destroy_light_textures(C.byref(app.light_textures), dev)
        self.lib.destroy_scene(C.byref(app.scene), dev)
        self.lib.destroy_ltc_table(C.byref(app.ltc_table), dev)
        self.lib.destroy_noise_table(C.byref(app.noise_table), dev)
        spec = app.scene_specification
        for i in range(spec.polygonal_light_count):
            self.lib.destroy_polygonal_light(C.byref(spec.polygonal_lights[i]))
        spec.polygonal_light_count = 0
        self._lights_keepalive = None
        if self._device:
            # releases the frame streams that create_hip_device made
            self.lib.destroy_hip_device(C.byref(app.device))
            self._device = None


class Renderer(HostScene):
    """The shading pass on one MI355X."""

    def __init__(self, hip_device=0, stream=None, fast_math=False, inline_rays=False, timing_stride=1, frames_in_flight=1, binary_traversal=False, arithmetic=None, band_count=0):
        super().__init__()
        self.binary_traversal = binary_traversal
        self.exchange = None
        self.timing_stride = timing_stride
        self.frames_in_flight = frames_in_flight
        if self.lib.create_hip_device(C.byref(self.app.device), hip_device, stream):
            raise RuntimeError("no usable HIP device: the shading pass has no CPU fallback")
        self._device = True
        # arithmetic_mode_t of include/vkr_shading_pass.h: "libm" (default; equals the oracle's
        # math mode 0 bit for bit), "fast", "exact" (polynomial; equals the oracle's math mode 1)
        self.arithmetic = arithmetic if arithmetic is not None else ("fast" if fast_math else "libm")
        self.fast_math = self.arithmetic == "fast"
        # launches per frame with wavefront rays (0: automatic, include/vkr_shading_pass.h band_count)
        self.band_count = band_count
        self.inline_rays = inline_rays

    def create_targets(self):
        if self.app.render_targets.radiance:
            self.lib.destroy_render_targets(C.byref(self.app.render_targets), self._dev())
        if self.lib.create_render_targets(C.byref(self.app.render_targets), self._dev(), C.byref(self.app.swapchain)):
            raise RuntimeError("create_render_targets failed")

    def create_pass(self):
        if self.app.shading_pass.constants_device:
            self.lib.destroy_shading_pass(C.byref(self.app.shading_pass), self._dev())
        self.app.shading_pass.arithmetic_mode = ARITHMETIC_MODES[self.arithmetic]
        self.app.shading_pass.band_count = int(self.band_count)
        self.app.shading_pass.inline_rays = int(self.inline_rays)
        self.app.shading_pass.timing_stride = int(self.timing_stride)
        self.app.shading_pass.frames_in_flight = int(self.frames_in_flight)
        self.app.shading_pass.binary_traversal = int(self.binary_traversal)
        if self.lib.create_shading_pass(C.byref(self.app.shading_pass), C.byref(self.app)):
            raise RuntimeError("create_shading_pass failed")

    def set_tiles(self, tile_size=16, rank=0, rank_count=1, slab_layout=False):
        t = self.app.tile_schedule
        t.tile_size, t.rank, t.rank_count, t.slab_layout = tile_size, rank, rank_count, int(slab_layout)

    def upload_visibility(self, visibility):
        v = np.ascontiguousarray(visibility, np.uint32)
        if self.lib.upload_visibility(C.byref(self.app), v.ctypes.data):
            raise RuntimeError("upload_visibility failed")

    def render_visibility(self):
        if self.lib.render_visibility_pass(C.byref(self.app)):
            raise RuntimeError("render_visibility_pass failed")

    def render(self, out_pointer=None):
        if self.lib.render_shading_pass(C.byref(self.app), out_pointer):
            raise RuntimeError("render_shading_pass failed")

    def next_frame_stream(self):
        return int(self.lib.get_next_frame_stream(C.byref(self.app)) or 0)

    def render_encoded(self, out_pointer, rgb8_pointer):
        if self.lib.render_shading_pass_encoded(C.byref(self.app), out_pointer, rgb8_pointer):
            raise RuntimeError("render_shading_pass_encoded failed")

    def last_ms(self):
        return float(self.lib.get_last_dispatch_milliseconds(C.byref(self.app)))

    def dispatch_ms(self, count):
        out = (C.c_float * count)()
        n = self.lib.get_dispatch_milliseconds(C.byref(self.app), out, count)
        return [float(out[i]) for i in range(n)]

    def shading_kernel_ms(self, count):
        out = (C.c_float * count)()
        n = self.lib.get_shading_kernel_milliseconds(C.byref(self.app), out, count)
        return [float(out[i]) for i in range(n)]

    def light_shaft_ms(self, count):
        out = (C.c_float * count)()
        n = self.lib.get_light_shaft_milliseconds(C.byref(self.app), out, count)
        return [float(out[i]) for i in range(n)]

    def frame_period_ms(self, count):
        out = (C.c_float * count)()
        n = self.lib.get_frame_period_milliseconds(C.byref(self.app), out, count)
        return [float(out[i]) for i in range(n)]

    def finish_frames(self):
        if self.lib.finish_frames(C.byref(self.app)):
            raise RuntimeError("finish_frames failed")

    def last_ray_count(self):
        return int(self.lib.get_last_ray_count(C.byref(self.app)))

    def traversal_statistics(self, wide_tree=None):
        """Work of the BVH traversal for the rays of the last wavefront frame (diagnostics).
        wide_tree None: the tree the frame walked; True / False: the four-wide / the binary one."""
        out = (C.c_uint64 * 12)()
        if wide_tree is None:
            wide_tree = bool(self.app.scene.acceleration_structure.wide_nodes) and not self.app.shading_pass.binary_traversal
        if self.lib.get_traversal_statistics_of_tree(C.byref(self.app), int(wide_tree), out):
            raise RuntimeError("get_traversal_statistics_of_tree failed")
        keys = ("rays", "node_visits", "triangle_tests", "blocked_rays", "wave_steps", "longest_ray_visits", "boxes_tested", "deepest_stack", "rays_beyond_lds_stack", "node_visits_of_blocked_rays")
        stats = dict(zip(keys, (int(v) for v in out)))
        stats["tree"] = "wide" if wide_tree else "binary"
        if not wide_tree:
            stats["boxes_tested"] = stats["node_visits"]
            for key in ("deepest_stack", "rays_beyond_lds_stack", "node_visits_of_blocked_rays"):
                del stats[key]
        return stats

    def light_shaft_statistics(self):
        """(patch, light) pairs of the last launch and how many of them needed no shadow rays (csrc/light_shafts.h)"""
        out = (C.c_uint64 * 12)()
        if self.lib.get_light_shaft_statistics(C.byref(self.app), out):
            raise RuntimeError("get_light_shaft_statistics failed")
        work = (C.c_uint64 * 3)()
        self.lib.get_light_shaft_work(C.byref(self.app), work)
        return {"pairs": int(out[0]), "clear_pairs": int(out[1]), "patches": int(out[2]), "lights": int(out[3]), "work": {"steps": int(work[0]), "triangle_batches": int(work[1]), "walks": int(work[2])},
                "not_clear": {"no_shaded_pixel": int(out[4]), "no_shaft": int(out[5]), "walk_too_long": int(out[6]), "queue_full": int(out[7]), "triangle_in_the_way": int(out[8]), "other": int(out[9])},
                # pairs whose rays the shading kernel decides against a handful of triangles, and those triangles
                "list_pairs": int(out[10]), "listed_triangles": int(out[11])}

    # -- multi-GPU exchange (include/vkr_slab_exchange.h) ---------------------------------
    def exchange_id(self):
        """The rendezvous token of a new communicator as 128 bytes (call on one rank, broadcast)."""
        token = capi.SlabExchangeId()
        if self.lib.get_slab_exchange_id(C.byref(token)):
            raise RuntimeError("get_slab_exchange_id failed (RCCL missing?)")
        return bytes(C.string_at(C.byref(token), 128))

    def create_exchange(self, token_bytes, slab_format="rgba32f"):
        token = capi.SlabExchangeId()
        C.memmove(C.byref(token), token_bytes, 128)
        self.exchange = capi.SlabExchange()
        if self.lib.create_slab_exchange(C.byref(self.exchange), C.byref(self.app), C.byref(token), capi.SLAB_FORMAT[slab_format]):
            self.exchange = None
            raise RuntimeError("create_slab_exchange failed")

    def create_exchange_with_gather(self, gather, slab_format="rgba32f"):
        """An exchange whose collective is the Python callable gather(rank, set, send_pointer, gathered_pointer,
        send_bytes, stream) -> 0 on success (include/vkr_slab_exchange.h slab_gather_function_t): any transport
        the caller has, e.g. a host-staged all-gather over a CPU process group."""
        prototype = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)

        def trampoline(context, rank, buffer_set, send, gathered, send_bytes, stream):
            try:
                return int(gather(int(rank), int(buffer_set), int(send or 0), int(gathered or 0), int(send_bytes), int(stream or 0)))
            except Exception as error:  # an exception must not unwind through the C frames
                print("slab gather callback failed: %r" % (error,), flush=True)
                return 1
        self._gather_keepalive = prototype(trampoline)
        self.exchange = capi.SlabExchange()
        if self.lib.create_slab_exchange_with_gather(C.byref(self.exchange), C.byref(self.app), C.cast(self._gather_keepalive, C.c_void_p), None, capi.SLAB_FORMAT[slab_format]):
            self.exchange = None
            raise RuntimeError("create_slab_exchange_with_gather failed")

    def create_local_exchange(self, group, slab_format="rgba32f"):
        """Joins a group made by capi.load().create_local_slab_group(rank_count): ranks of this process that
        exchange their slabs with device-to-device copies (call from the rank's own thread)"""
        self.exchange = capi.SlabExchange()
        if self.lib.create_local_slab_exchange(C.byref(self.exchange), C.byref(self.app), group, capi.SLAB_FORMAT[slab_format]):
            self.exchange = None
            raise RuntimeError("create_local_slab_exchange failed")

    def assemble_on_demand(self, on=True):
        """The frame stays tile-major - the gathered slabs - until a reader asks for it (finish_exchange(), assemble_exchanged());
        include/vkr_slab_exchange.h slab_exchange_t.assemble_on_demand"""
        self.exchange.assemble_on_demand = int(bool(on))

    def assemble_exchanged(self, out_pointer=None):
        if self.lib.assemble_exchanged_frame(C.byref(self.app), C.byref(self.exchange), out_pointer):
            raise RuntimeError("assemble_exchanged_frame failed")

    def destroy_exchange(self):
        if self.exchange is not None:
            self.lib.destroy_slab_exchange(C.byref(self.exchange), C.byref(self.app))
            self.exchange = None

    def render_and_exchange(self, out_pointer=None):
        if self.lib.render_and_exchange_frame(C.byref(self.app), C.byref(self.exchange), out_pointer):
            raise RuntimeError("render_and_exchange_frame failed")

    def finish_exchange(self):
        if self.lib.finish_slab_exchange(C.byref(self.app), C.byref(self.exchange)):
            raise RuntimeError("finish_slab_exchange failed")

    def exchange_ms(self):
        """(shade, all-gather, scatter) of the most recent timed frame, or None"""
        out = (C.c_float * 3)()
        if not self.lib.get_slab_exchange_milliseconds(C.byref(self.exchange), out):
            return None
        return [float(v) for v in out]

    def sync(self):
        self.lib.wait_for_device(C.byref(self.app.device))

    def read_radiance(self):
        e = self.app.swapchain.extent
        out = np.zeros((e.height, e.width, 4), np.float32)
        if self.lib.read_back_radiance(C.byref(self.app), out.ctypes.data):
            raise RuntimeError("read_back_radiance failed")
        return out

    def begin_read_back(self, slot=0, pointer=None, nbytes=0):
        """Queues the copy of a device buffer (default: the radiance target) into the slot's pinned staging memory behind
        the most recent frame and returns at once (include/vkr_shading_pass.h begin_read_back)"""
        if self.lib.begin_read_back(C.byref(self.app), slot, pointer, nbytes):
            raise RuntimeError("begin_read_back failed")

    def end_read_back(self, slot=0, shape=None, dtype=np.float32):
        """Waits for the slot's copy; returns a numpy VIEW of the pinned staging memory (valid until the slot's next
        begin_read_back), shaped like the radiance target unless `shape` says otherwise"""
        address = self.lib.end_read_back(C.byref(self.app), slot)
        if not address:
            raise RuntimeError("end_read_back failed")
        if shape is None:
            e = self.app.swapchain.extent
            shape = (e.height, e.width, 4)
        count = int(np.prod(shape))
        buffer = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(address)
        return np.frombuffer(buffer, dtype=dtype, count=count).reshape(shape)

    def read_visibility(self):
        e = self.app.swapchain.extent
        out = np.zeros((e.height, e.width), np.uint32)
        if self.lib.read_back_visibility(C.byref(self.app), out.ctypes.data):
            raise RuntimeError("read_back_visibility failed")
        return out

    def read_encoded(self, output_linear_rgb=False, frame_bits=0):
        e = self.app.swapchain.extent
        self.app.screenshot.frame_bits = frame_bits
        if self.lib.encode_output(C.byref(self.app), int(output_linear_rgb)):
            raise RuntimeError("encode_output failed")
        out = np.zeros((e.height, e.width, 4), np.uint8)
        if self.lib.read_back_encoded(C.byref(self.app), out.ctypes.data):
            raise RuntimeError("read_back_encoded failed")
        self.app.screenshot.frame_bits = 0
        return out

    def slab_pixel_count(self, rank=0):
        return int(self.lib.get_slab_pixel_count(C.byref(self.app), rank))

    def encode_slab(self, slab_pointer, encoded_pointer, pixel_count, output_linear_rgb=False):
        if self.lib.encode_slab(C.byref(self.app), slab_pointer, encoded_pointer, pixel_count, int(output_linear_rgb)):
            raise RuntimeError("encode_slab failed")

    def encode_slab_rgb8(self, slab_pointer, packed_pointer, pixel_count, output_linear_rgb=False):
        if self.lib.encode_slab_rgb8(C.byref(self.app), slab_pointer, packed_pointer, pixel_count, int(output_linear_rgb)):
            raise RuntimeError("encode_slab_rgb8 failed")

    def assemble_rgb8(self, gathered_pointer, out_pointer=None):
        if self.lib.assemble_rgb8_frame_from_slabs(C.byref(self.app), gathered_pointer, out_pointer):
            raise RuntimeError("assemble_rgb8_frame_from_slabs failed")

    def assemble_encoded(self, gathered_pointer, out_pointer=None):
        if self.lib.assemble_encoded_frame_from_slabs(C.byref(self.app), gathered_pointer, out_pointer):
            raise RuntimeError("assemble_encoded_frame_from_slabs failed")

    def assemble(self, gathered_pointer, out_pointer=None):
        if self.lib.assemble_frame_from_slabs(C.byref(self.app), gathered_pointer, out_pointer):
            raise RuntimeError("assemble_frame_from_slabs failed")


def frames_in_flight_for(rank_count):
    """Depth of the frame pipeline that bench.py and profiles/tools/predict_scaling.py use when the frame is tiled over
    `rank_count` GPUs.  Three frames like the reference's frame queue (main.c:1498); four when a rank's slab is an eighth
    of the frame or less: its kernels then last about as long as their slowest wave, whatever the slab's size, and one more
    frame in flight fills what that leaves idle (profiles/r07c: config 3 at N = 8 0.200 -> 0.190 ms per slab, config 4
    2.42 -> 2.39; five and more lose again, and at N <= 4 and on one GPU three are best)."""
    return 4 if rank_count >= 8 else 3


def setup_config(scene, config, dataset, width=None, height=None, **overrides):
    """Applies one of the BASELINE.json configurations to a HostScene / Renderer."""
    from . import synthetic
    settings = dict(synthetic.CONFIG_SETTINGS[config])
    if width:
        settings["width"] = width
    if height:
        settings["height"] = height
    settings.update(overrides)
    wants_rays = bool(settings.get("trace_shadow_rays", False))
    # (a builder named by the caller wins; rays alone ask for the default builder)
    scene.load_scene(dataset["scene"], dataset["textures"], acceleration_structure=overrides.get("acceleration_structure") or wants_rays)
    scene.load_ltc_table(dataset["ltc"], dataset["fresnel_count"])
    scene.load_noise_table("white")
    cam = synthetic.DEFAULT_CAMERA
    scene.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
    scene.set_lights(synthetic.config_lights(config))
    settings.pop("acceleration_structure", None)
    scene.set_settings(**settings)
    return settings
