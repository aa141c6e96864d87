"""ctypes mirror of the C-ABI declared in include/*.h (libvkr_shading.so).

This is the binding a Python host would use; the tests and bench.py drive the
library exclusively through it.  Struct layouts are verified against
get_abi_struct_sizes() at load time."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKR_SHADING_LIBRARY: an alternative build of the same library for A/B measurements (profiles/tools/ab_build.sh)
LIB_PATH = os.environ.get("VKR_SHADING_LIBRARY") or os.path.join(_HERE, "libvkr_shading.so")

c_float_p = C.POINTER(C.c_float)


class Device(C.Structure):
    _fields_ = [("hip_device", C.c_int32), ("stream", C.c_void_p), ("ray_tracing_supported", C.c_uint32),
                ("compute_unit_count", C.c_int32), ("architecture", C.c_char * 64), ("frame_streams", C.c_void_p * 8)]  # VKR_MAX_FRAMES_IN_FLIGHT


class PolygonalLight(C.Structure):
    _fields_ = [("rotation_angles", C.c_float * 3), ("scaling_x", C.c_float),
                ("translation", C.c_float * 3), ("scaling_y", C.c_float),
                ("radiant_flux", C.c_float * 3), ("inv_scaling_x", C.c_float),
                ("surface_radiance", C.c_float * 3), ("inv_scaling_y", C.c_float),
                ("plane", C.c_float * 4), ("vertex_count", C.c_uint32), ("texturing_technique", C.c_int32),
                ("texture_index", C.c_uint32), ("padding_0", C.c_uint32),
                ("rotation", (C.c_float * 4) * 3), ("area", C.c_float), ("rcp_area", C.c_float),
                ("padding_1", C.c_float * 2), ("texture_file_path", C.c_void_p),
                ("vertices_plane_space", c_float_p), ("vertices_world_space", c_float_p), ("fan_areas", c_float_p)]


class Camera(C.Structure):
    _fields_ = [("position_world_space", C.c_float * 3), ("rotation_z", C.c_float), ("rotation_x", C.c_float),
                ("vertical_fov", C.c_float), ("near", C.c_float), ("far", C.c_float), ("speed", C.c_float),
                ("rotate_camera", C.c_int), ("rotation_x_0", C.c_float), ("rotation_z_0", C.c_float)]


class LtcConstants(C.Structure):
    _fields_ = [("fresnel_index_factor", C.c_float), ("fresnel_index_summand", C.c_float),
                ("roughness_factor", C.c_float), ("roughness_summand", C.c_float),
                ("inclination_factor", C.c_float), ("inclination_summand", C.c_float), ("padding", C.c_float * 2)]


class LtcTable(C.Structure):
    _fields_ = [("roughness_count", C.c_uint32), ("inclination_count", C.c_uint32), ("fresnel_count", C.c_uint32),
                ("host_rgba", C.POINTER(C.c_uint16)), ("host_rg", C.POINTER(C.c_uint16)),
                ("device_rgba", C.c_void_p), ("device_rg", C.c_void_p), ("constants", LtcConstants)]


class Extent2D(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32)]


class Extent3D(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32)]


class NoiseTable(C.Structure):
    _fields_ = [("resolution", Extent3D), ("host_data", C.POINTER(C.c_uint16)), ("device_data", C.c_void_p),
                ("random_seed", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("triangle_count", C.c_uint64), ("dequantization_factor", C.c_float * 3), ("dequantization_summand", C.c_float * 3),
                ("host_positions", C.POINTER(C.c_uint32)), ("host_normals_and_tex_coords", C.POINTER(C.c_uint16)),
                ("host_material_indices", C.POINTER(C.c_uint8)),
                ("positions", C.c_void_p), ("normals_and_tex_coords", C.c_void_p), ("material_indices", C.c_void_p)]


class Materials(C.Structure):
    _fields_ = [("material_count", C.c_uint64), ("material_names", C.POINTER(C.c_char_p)),
                ("host_constants", c_float_p), ("constants", C.c_void_p),
                ("textured", C.c_uint32), ("host_texture_descriptors", C.POINTER(C.c_uint32)), ("host_texels", C.POINTER(C.c_uint8)),
                ("texel_count", C.c_uint64), ("texture_descriptors", C.c_void_p), ("texels", C.c_void_p), ("srgb_table", C.c_void_p)]


class AccelerationStructure(C.Structure):
    _fields_ = [("triangle_vertices", C.c_void_p), ("triangle_indices", C.c_void_p), ("nodes", C.c_void_p),
                ("node_count", C.c_uint32), ("root", C.c_uint32),
                ("grid_origin", C.c_float * 3), ("grid_inverse_cell", C.c_float * 3),
                ("wide_nodes", C.c_void_p), ("wide_node_count", C.c_uint32), ("wide_stack_need", C.c_uint32),
                ("builder", C.c_uint32), ("build_milliseconds", C.c_float), ("leaf_count", C.c_uint32), ("reserved", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("mesh", Mesh), ("materials", Materials), ("acceleration_structure", AccelerationStructure)]


class SceneSpecification(C.Structure):
    _fields_ = [("file_path", C.c_void_p), ("texture_path", C.c_void_p), ("quick_save_path", C.c_void_p),
                ("camera", Camera), ("polygonal_light_count", C.c_uint32), ("polygonal_lights", C.POINTER(PolygonalLight))]


class RenderSettings(C.Structure):
    _fields_ = [("exposure_factor", C.c_float), ("roughness_factor", C.c_float), ("sample_count", C.c_uint32),
                ("sampling_strategies", C.c_int32), ("mis_heuristic", C.c_int32), ("mis_visibility_estimate", C.c_float),
                ("polygon_sampling_technique", C.c_int32), ("error_display", C.c_int32), ("error_min_exponent", C.c_float),
                ("noise_type", C.c_int32), ("animate_noise", C.c_uint32), ("trace_shadow_rays", C.c_uint32),
                ("show_polygonal_lights", C.c_uint32), ("show_gui", C.c_uint32), ("v_sync", C.c_uint32)]


class PerFrameConstants(C.Structure):
    _fields_ = [("mesh_dequantization_factor", C.c_float * 3), ("padding_0", C.c_float),
                ("mesh_dequantization_summand", C.c_float * 3), ("error_factor", C.c_float),
                ("world_to_projection_space", (C.c_float * 4) * 4),
                ("pixel_to_ray_direction_world_space", (C.c_float * 4) * 3),
                ("camera_position_world_space", C.c_float * 3), ("mis_visibility_estimate", C.c_float),
                ("viewport_size", Extent2D), ("cursor_position", C.c_int32 * 2),
                ("exposure_factor", C.c_float), ("roughness_factor", C.c_float),
                ("noise_resolution_mask", C.c_uint32 * 2), ("noise_texture_index_mask", C.c_uint32),
                ("frame_bits", C.c_uint32), ("padding_3", C.c_uint32 * 2), ("noise_random_numbers", C.c_uint32 * 4),
                ("ltc_constants", LtcConstants)]


class Swapchain(C.Structure):
    _fields_ = [("extent", Extent2D)]


class RenderTargets(C.Structure):
    _fields_ = [("visibility_buffer", C.c_void_p), ("radiance", C.c_void_p), ("encoded", C.c_void_p), ("extent", Extent2D)]


class Screenshot(C.Structure):
    _fields_ = [("frame_bits", C.c_uint32)]


class TileSchedule(C.Structure):
    _fields_ = [("tile_size", C.c_uint32), ("rank", C.c_uint32), ("rank_count", C.c_uint32), ("slab_layout", C.c_uint32)]


class LightTextures(C.Structure):
    _fields_ = [("texture_count", C.c_uint32), ("host_descriptors", C.POINTER(C.c_uint32 * 4)), ("host_texels", C.POINTER(C.c_float)),
                ("texel_count", C.c_uint64), ("descriptors", C.c_void_p), ("texels", C.c_void_p)]


class ShadingPass(C.Structure):
    _fields_ = [("use_ray_tracing", C.c_uint32), ("variant", C.c_int32), ("max_polygon_vertex_count", C.c_uint32),
                ("constants_device", C.c_void_p), ("constants_host", C.c_void_p), ("constants_size", C.c_size_t), ("constants_ring", C.c_void_p),
                ("arithmetic_mode", C.c_int32), ("inline_rays", C.c_int32), ("frames_in_flight", C.c_uint32), ("inputs_changed", C.c_uint32), ("last_frame_in_flight", C.c_uint32), ("wavefront", C.c_void_p), ("ray_counter", C.c_void_p), ("pixel_materials", C.c_void_p), ("pixel_materials_size", C.c_size_t), ("last_dispatch_ms", C.c_float), ("timing_ring", C.c_void_p), ("timing_ring_size", C.c_uint32), ("timing_cursor", C.c_uint32),
                ("timing_stride", C.c_uint32), ("frame_counter", C.c_uint32), ("last_frame_traced_rays", C.c_uint32), ("binary_traversal", C.c_int32),
                ("wait_before_next_frame", C.c_void_p), ("last_frame_stream", C.c_void_p), ("band_count", C.c_uint32), ("last_band_count", C.c_uint32),
                ("last_shaft_groups", C.c_uint32), ("reserved", C.c_uint32), ("readback", C.c_void_p)]


class Application(C.Structure):
    _fields_ = [("device", Device), ("swapchain", Swapchain), ("scene_specification", SceneSpecification),
                ("render_settings", RenderSettings), ("scene", Scene), ("noise_table", NoiseTable),
                ("ltc_table", LtcTable), ("render_targets", RenderTargets), ("screenshot", Screenshot),
                ("shading_pass", ShadingPass), ("tile_schedule", TileSchedule), ("light_textures", LightTextures)]


class Experiment(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("scene_index", C.c_int32), ("quick_save_path", C.c_char_p),
                ("use_hdr", C.c_uint32), ("screenshot_path", C.c_char_p), ("render_settings", RenderSettings)]


class ExperimentList(C.Structure):
    _fields_ = [("experiments", C.POINTER(Experiment)), ("experiment", C.POINTER(Experiment)), ("count", C.c_uint32),
                ("next", C.c_uint32), ("next_setup_time", C.c_double), ("next_setup_frame", C.c_uint32),
                ("frame_index", C.c_uint32), ("state", C.c_int32)]


# stack entries per lane that trace_shadow_rays_wide keeps in LDS (csrc/lbvh.h kWideStackLds)
WIDE_STACK_LDS = 16
MAX_FRAMES_IN_FLIGHT = 8  # VKR_MAX_FRAMES_IN_FLIGHT


class SlabExchangeId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


class SlabExchange(C.Structure):
    _fields_ = [("binding", C.c_void_p), ("gather", C.c_void_p), ("gather_context", C.c_void_p), ("rank", C.c_uint32), ("rank_count", C.c_uint32), ("format", C.c_int32),
                ("slab_pixel_count", C.c_uint64), ("send_bytes", C.c_uint64), ("set_count", C.c_uint32), ("next_set", C.c_uint32),
                ("slab_radiance", C.c_void_p * MAX_FRAMES_IN_FLIGHT), ("send", C.c_void_p * MAX_FRAMES_IN_FLIGHT),
                ("gathered", C.c_void_p * MAX_FRAMES_IN_FLIGHT), ("stream", C.c_void_p),
                ("rendered", C.c_void_p * MAX_FRAMES_IN_FLIGHT), ("assembled", C.c_void_p * MAX_FRAMES_IN_FLIGHT),
                ("timing", (C.c_void_p * 5) * MAX_FRAMES_IN_FLIGHT), ("timed", C.c_uint32 * MAX_FRAMES_IN_FLIGHT),
                ("frame_counter", C.c_uint64), ("timing_stride", C.c_uint32), ("last_frame", C.c_void_p),
                ("assemble_on_demand", C.c_uint32), ("last_set", C.c_uint32), ("last_frame_assembled", C.c_uint32)]


SLAB_FORMAT = {"rgba32f": 0, "rgb8": 1}

ABI_STRUCTS = [Device, PolygonalLight, Camera, LtcConstants, LtcTable, NoiseTable, Mesh, Materials,
               AccelerationStructure, Scene, SceneSpecification, RenderSettings, PerFrameConstants, Swapchain,
               RenderTargets, Screenshot, TileSchedule, LightTextures, ShadingPass, Application, Experiment, ExperimentList,
               SlabExchangeId, SlabExchange]

# every symbol include/*.h declares, with (restype, argtypes)
P = C.POINTER
SIGNATURES = {
    "create_hip_device": (C.c_int, [P(Device), C.c_int32, C.c_void_p]),
    "specify_default_scene": (None, [P(SceneSpecification)]),
    "create_and_assign_light_textures": (C.c_int, [P(LightTextures), P(Device), P(SceneSpecification)]),
    "destroy_light_textures": (None, [P(LightTextures), P(Device)]),
    "destroy_hip_device": (None, [P(Device)]),
    "wait_for_device": (C.c_int, [P(Device)]),
    "set_polygonal_light_vertex_count": (C.c_int, [P(PolygonalLight), C.c_uint32]),
    "update_polygonal_light": (None, [P(PolygonalLight)]),
    "duplicate_polygonal_light": (PolygonalLight, [P(PolygonalLight)]),
    "destroy_polygonal_light": (None, [P(PolygonalLight)]),
    "get_world_to_view_space": (None, [P((C.c_float * 4) * 4), P(Camera)]),
    "get_view_to_projection_space": (None, [P((C.c_float * 4) * 4), P(Camera), C.c_float]),
    "get_world_to_projection_space": (None, [P((C.c_float * 4) * 4), P(Camera), C.c_float]),
    "load_ltc_table": (C.c_int, [P(LtcTable), P(Device), C.c_char_p, C.c_uint32]),
    "destroy_ltc_table": (None, [P(LtcTable), P(Device)]),
    "get_default_noise_resolution": (Extent3D, [C.c_int32]),
    "load_noise_table": (C.c_int, [P(NoiseTable), P(Device), Extent3D, C.c_int32]),
    "destroy_noise_table": (None, [P(NoiseTable), P(Device)]),
    "set_noise_constants": (None, [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32), P(NoiseTable), C.c_uint32]),
    "get_material_texture_suffix": (C.c_char_p, [C.c_int32]),
    "load_scene": (C.c_int, [P(Scene), P(Device), C.c_char_p, C.c_char_p, C.c_uint32]),
    "destroy_scene": (None, [P(Scene), P(Device)]),
    "specify_default_render_settings": (None, [P(RenderSettings)]),
    "get_min_polygonal_light_vertex_count": (C.c_uint32, [P(SceneSpecification)]),
    "get_max_polygonal_light_vertex_count": (C.c_uint32, [P(SceneSpecification)]),
    "get_max_polygon_vertex_count": (C.c_uint32, [P(SceneSpecification), P(RenderSettings)]),
    "destroy_scene_specification": (None, [P(SceneSpecification)]),
    "quick_save": (None, [P(SceneSpecification)]),
    "quick_load": (None, [P(SceneSpecification), P(C.c_uint32)]),
    "create_render_targets": (C.c_int, [P(RenderTargets), P(Device), P(Swapchain)]),
    "destroy_render_targets": (None, [P(RenderTargets), P(Device)]),
    "get_constant_buffer_size": (C.c_size_t, [P(Application)]),
    "write_constants": (None, [C.c_void_p, P(Application)]),
    "create_shading_pass": (C.c_int, [P(ShadingPass), P(Application)]),
    "destroy_shading_pass": (None, [P(ShadingPass), P(Device)]),
    "render_visibility_pass": (C.c_int, [P(Application)]),
    "render_shading_pass": (C.c_int, [P(Application), C.c_void_p]),
    "get_slab_pixel_count": (C.c_uint64, [P(Application), C.c_uint32]),
    "assemble_frame_from_slabs": (C.c_int, [P(Application), C.c_void_p, C.c_void_p]),
    "encode_output": (C.c_int, [P(Application), C.c_uint32]),
    "read_back_radiance": (C.c_int, [P(Application), C.c_void_p]),
    "read_back_encoded": (C.c_int, [P(Application), C.c_void_p]),
    "read_back_visibility": (C.c_int, [P(Application), C.c_void_p]),
    "check_hardware_wave_slots": (C.c_int, [P(Device), C.c_uint32, C.c_uint32, P(C.c_uint64)]),
    "copy_with_workgroups": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "begin_read_back": (C.c_int, [P(Application), C.c_uint32, C.c_void_p, C.c_uint64]),
    "end_read_back": (C.c_void_p, [P(Application), C.c_uint32]),
    "upload_visibility": (C.c_int, [P(Application), C.c_void_p]),
    "get_last_dispatch_milliseconds": (C.c_float, [P(Application)]),
    "get_last_ray_count": (C.c_uint64, [P(Application)]),
    "assemble_encoded_frame_from_slabs": (C.c_int, [P(Application), C.c_void_p, C.c_void_p]),
    "encode_slab": (C.c_int, [P(Application), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]),
    "get_next_frame_stream": (C.c_void_p, [P(Application)]),
    "render_shading_pass_encoded": (C.c_int, [P(Application), C.c_void_p, C.c_void_p]),
    "encode_slab_rgb8": (C.c_int, [P(Application), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]),
    "assemble_rgb8_frame_from_slabs": (C.c_int, [P(Application), C.c_void_p, C.c_void_p]),
    "get_traversal_statistics": (C.c_int, [P(Application), P(C.c_uint64)]),
    "get_light_shaft_statistics": (C.c_int, [P(Application), P(C.c_uint64)]),
    "get_light_shaft_milliseconds": (C.c_uint32, [P(Application), P(C.c_float), C.c_uint32]),
    "read_back_light_shafts": (C.c_uint64, [P(Application), C.c_void_p, C.c_uint64]),
    "get_light_shaft_work": (C.c_int, [P(Application), P(C.c_uint64)]),
    "get_dispatch_milliseconds": (C.c_uint32, [P(Application), P(C.c_float), C.c_uint32]),
    "get_shading_kernel_milliseconds": (C.c_uint32, [P(Application), P(C.c_float), C.c_uint32]),
    "get_frame_period_milliseconds": (C.c_uint32, [P(Application), P(C.c_float), C.c_uint32]),
    "finish_frames": (C.c_int, [P(Application)]),
    "mark_inputs_changed": (None, [P(Application)]),
    "get_slab_pixel_coordinates": (C.c_uint64, [P(Application), C.c_uint32, C.c_void_p, C.c_uint64]),
    "evaluate_device_arithmetic": (C.c_int, [P(Device), C.c_uint32, P(C.c_float), P(C.c_float), P(C.c_float), C.c_uint32]),
    "compare_device_arithmetic": (C.c_int, [P(Device), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, P(C.c_uint64)]),
    "compare_device_division": (C.c_int, [P(Device), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_uint64)]),
    "get_traversal_statistics_of_tree": (C.c_int, [P(Application), C.c_uint32, P(C.c_uint64)]),
    "get_abi_struct_sizes": (C.c_uint32, [P(C.c_uint64), C.c_uint32]),
    "create_experiment_list": (None, [P(ExperimentList)]),
    "destroy_experiment_list": (None, [P(ExperimentList)]),
    "apply_experiment": (C.c_int, [P(Application), P(Experiment), C.c_char_p]),
    "half_to_float": (C.c_float, [C.c_uint16]),
    "format_screenshot_path": (C.c_void_p, [C.c_char_p, C.c_float]),
    "write_png_rgb8": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "write_hdr_rgb32f": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "take_screenshot": (C.c_int, [P(Application), C.c_char_p, C.c_char_p]),
    "get_slab_exchange_id": (C.c_int, [P(SlabExchangeId)]),
    "create_slab_exchange": (C.c_int, [P(SlabExchange), P(Application), P(SlabExchangeId), C.c_int]),
    "create_slab_exchange_with_gather": (C.c_int, [P(SlabExchange), P(Application), C.c_void_p, C.c_void_p, C.c_int]),
    "create_local_slab_group": (C.c_void_p, [C.c_uint32]),
    "destroy_local_slab_group": (None, [C.c_void_p]),
    "create_local_slab_exchange": (C.c_int, [P(SlabExchange), P(Application), C.c_void_p, C.c_int]),
    "destroy_slab_exchange": (None, [P(SlabExchange), P(Application)]),
    "render_and_exchange_frame": (C.c_int, [P(Application), P(SlabExchange), C.c_void_p]),
    "assemble_exchanged_frame": (C.c_int, [P(Application), P(SlabExchange), C.c_void_p]),
    "all_gather_slabs": (C.c_int, [P(SlabExchange), C.c_void_p, C.c_void_p, C.c_void_p]),
    "finish_slab_exchange": (C.c_int, [P(Application), P(SlabExchange)]),
    "get_slab_exchange_milliseconds": (C.c_uint32, [P(SlabExchange), P(C.c_float)]),
}

_lib = None


def load():
    """Loads libvkr_shading.so and checks struct layouts.  Raises if the library
    has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    # VKR_SHADING_LIBRARY: another build of the same library, e.g. libvkr_shading_ieee.so (the check build whose
    # divisions and square roots are the compiler's full-range IEEE expansions; tests/test_gpu_division_window.py)
    path = os.environ.get("VKR_SHADING_LIBRARY") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C vulkan_renderer_amd/csrc -j`." % os.path.basename(path))
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    sizes = (C.c_uint64 * 32)()
    count = lib.get_abi_struct_sizes(sizes, 32)
    if count != len(ABI_STRUCTS):
        raise RuntimeError("ABI mismatch: library reports %d structs, binding has %d" % (count, len(ABI_STRUCTS)))
    for cls, size in zip(ABI_STRUCTS, sizes):
        if C.sizeof(cls) != size:
            raise RuntimeError("ABI mismatch for %s: ctypes %d bytes, C %d bytes" % (cls.__name__, C.sizeof(cls), size))
    _lib = lib
    return lib
