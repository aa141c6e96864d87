/* The shading pass behind the reference's own entry points.
 *
 * Reference call path (src/main.c): create_shading_pass (:598) selects the shader
 * variant from render_settings_t + scene_specification_t, write_constants (:2114)
 * fills the uniform buffer every frame, record_render_frame_commands (:1428-1434)
 * runs the pass with vkCmdDraw(3) over the viewport, implement_screenshot (:1719)
 * reads the result back.  The functions below keep those names and argument
 * meanings; application_t is reduced to the members that path touches. */
#ifndef VKR_SHADING_PASS_H
#define VKR_SHADING_PASS_H
#include "vkr_device.h"
#include "vkr_camera.h"
#include "vkr_polygonal_light.h"
#include "vkr_ltc_table.h"
#include "vkr_noise_table.h"
#include "vkr_scene.h"

/*! reference main.h:28-42 */
typedef struct scene_specification_s {
	char* file_path;
	char* texture_path;
	char* quick_save_path;
	first_person_camera_t camera;
	uint32_t polygonal_light_count;
	polygonal_light_t* polygonal_lights;
} scene_specification_t;

/*! reference main.h:45-67 */
typedef enum sampling_strategies_e {
	sampling_strategies_diffuse_only,
	sampling_strategies_diffuse_ggx_mis,
	sampling_strategies_diffuse_specular_separately,
	sampling_strategies_diffuse_specular_mis,
	sampling_strategies_diffuse_specular_random,
	sampling_strategies_count
} sampling_strategies_t;

/*! reference main.h:71-89 */
typedef enum mis_heuristic_e {
	mis_heuristic_balance,
	mis_heuristic_power,
	mis_heuristic_weighted,
	mis_heuristic_optimal_clamped,
	mis_heuristic_optimal,
	mis_heuristic_count
} mis_heuristic_t;

/*! reference main.h:93-118.  All displays are implemented (the projected solid angle
	techniques define them, shading_pass.frag.glsl:80-114, polygon_sampling.glsl:823-883);
	a frame with an error display returns before it samples, so it traces no rays. */
typedef enum error_display_e {
	error_display_none,
	error_display_diffuse_backward,
	error_display_diffuse_backward_scaled,
	error_display_diffuse_forward,
	error_display_specular_backward,
	error_display_specular_backward_scaled,
	error_display_specular_forward,
	error_display_count
} error_display_t;

/*! reference main.h:128-159, same member order */
typedef struct render_settings_s {
	float exposure_factor, roughness_factor;
	uint32_t sample_count;
	sampling_strategies_t sampling_strategies;
	mis_heuristic_t mis_heuristic;
	float mis_visibility_estimate;
	sample_polygon_technique_t polygon_sampling_technique;
	error_display_t error_display;
	float error_min_exponent;
	noise_type_t noise_type;
	VkBool32 animate_noise;
	VkBool32 trace_shadow_rays;
	VkBool32 show_polygonal_lights;
	VkBool32 show_gui;
	VkBool32 v_sync;
} render_settings_t;

/*! reference main.h:488-505; byte image of the uniform block
	shared_constants.glsl:20-66 (std140, row_major) */
typedef struct per_frame_constants_s {
	float mesh_dequantization_factor[3], padding_0, mesh_dequantization_summand[3];
	float error_factor;
	float world_to_projection_space[4][4];
	float pixel_to_ray_direction_world_space[3][4];
	float camera_position_world_space[3];
	float mis_visibility_estimate;
	VkExtent2D viewport_size;
	int32_t cursor_position[2];
	float exposure_factor;
	float roughness_factor;
	uint32_t noise_resolution_mask[2];
	uint32_t noise_texture_index_mask;
	uint32_t frame_bits;
	uint32_t padding_3[2];
	uint32_t noise_random_numbers[4];
	ltc_constants_t ltc_constants;
} per_frame_constants_t;

/*! Only the extent of the reference swapchain_t matters to the pass */
typedef struct swapchain_s {
	VkExtent2D extent;
} swapchain_t;

/*! Replaces render_targets_t (reference main.h:233-247) plus the swapchain
	image the pass writes: linear device buffers, row-major. */
typedef struct render_targets_s {
	/*! R32_UINT primitive index per pixel, 0xFFFFFFFF = nothing visible
		(reference main.c:282-298, cleared at main.c:1409) */
	void* visibility_buffer;
	/*! RGBA32F, vec4(final_color * exposure, 1): the value the reference
		computes at shading_pass.frag.glsl:866 before any output encoding */
	void* radiance;
	/*! RGBA8 after the reference's output encoding (sRGB or half-bit split,
		shading_pass.frag.glsl:871-892); filled by encode_output() */
	void* encoded;
	VkExtent2D extent;
} render_targets_t;

/*! frame_bits of reference screenshot_t (main.h:405-428): 0 = LDR frame, 1 / 2 =
	low / high byte of the half-float HDR screenshot */
typedef struct screenshot_s {
	uint32_t frame_bits;
} screenshot_t;

/*! How pixels are distributed over GPUs: the image is cut into square tiles,
	tile t (row-major) belongs to rank t % rank_count, and a rank stores its tiles
	densely one after the other ("slab").  rank_count == 1 renders in place unless
	slab_layout is set (the dense layout with a single rank: how the exchange path of
	include/vkr_slab_exchange.h is exercised on one GPU).  The blocks of a tile are consecutive in the
	launch, so with one rank the tile size only decides which pixels are in flight together; tile_size 0
	then picks the fastest order (64, or 128 above three megapixels: 2 % at BASELINE config 3).  Below
	16, and 0 with a slab layout, means 16. */
typedef struct tile_schedule_s {
	uint32_t tile_size;
	uint32_t rank, rank_count;
	VkBool32 slab_layout;
} tile_schedule_t;

/*! How the kernels evaluate what GLSL leaves to the driver (division, sqrt, inversesqrt and the
	transcendental functions; the reference notes the freedom at polygon_sampling.glsl:79-82).
	All three run the same program.
	- libm (0, the default): IEEE division and square roots, no contraction, inversesqrt as
	  1 / sqrt, atan / acos / sin / cos / log2 / pow with the operations of glibc 2.35 - the
	  arithmetic of the CPU oracle in its libm mode, which is pinned bit for bit against the
	  reference's shader source compiled as C++.  Frames equal that oracle's in every bit.
	- fast (1): approximate reciprocals and roots (1 ulp), contraction.
	- polynomial (2; "exact" in bench.py and the Python driver): the libm mode with ONE function replaced, the
	  arctangent, by a polynomial behind a single division (the oracle's math mode 1 mirrors it bit for bit).
	  RMSE 2.3e-7 against libm at 1920x1080 config 3, no pixel across a discontinuity of the shader. */
typedef enum arithmetic_mode_e {
	arithmetic_mode_libm = 0,
	arithmetic_mode_fast = 1,
	arithmetic_mode_polynomial = 2,
	arithmetic_mode_count
} arithmetic_mode_t;

/*! Replaces shading_pass_t (reference main.h:278-285).  The "pipeline" is a
	pre-compiled kernel variant chosen on the same axes as the reference's
	preprocessor defines (main.c:752-792). */
typedef struct shading_pass_s {
	VkBool32 use_ray_tracing;
	/*! index into the kernel variant table, negative if none */
	int32_t variant;
	/*! MAX_POLYGON_VERTEX_COUNT of the selected variant (main.c:194-216) */
	uint32_t max_polygon_vertex_count;
	/*! device + pinned host staging copy of the constant buffer of the most recent
		frame (one slot of constants_ring) */
	void* constants_device;
	void* constants_host;
	size_t constants_size;
	/*! ring of constant buffers, one per set of constants in flight */
	void* constants_ring;
	/*! arithmetic mode, an arithmetic_mode_t; set before create_shading_pass */
	int32_t arithmetic_mode;
	/*! shadow rays: 0 = wavefront (shade -> compacted ray queue -> trace kernel ->
		ordered resolve; default), 1 = every lane walks the BVH inside the shading
		kernel.  Both give identical results. */
	int32_t inline_rays;
	/*! 0 or 1: render_shading_pass orders everything on device->stream.  n = 2 ...
		VKR_MAX_FRAMES_IN_FLIGHT: frames with wavefront shadow rays take turns on n of the
		device's frame streams and overlap (trace of frame k with shading of frames
		k + 1 ...); they still complete in order; their output is complete for work
		on device->stream only after finish_frames() or one of the entry points that read
		or post-process the targets (read_back_*, encode_*, assemble_*, take_screenshot,
		render_visibility_pass, wait_for_device).  Set before create_shading_pass. */
	uint32_t frames_in_flight;
	/*! set by the entry points that produce inputs of the pass on device->stream
		(render_visibility_pass, upload_visibility) and by mark_inputs_changed(): the
		next frame in flight waits for device->stream once */
	uint32_t inputs_changed;
	/*! pipeline depth (>= 2) if the last render_shading_pass ran on a frame stream, else 0 (it
		falls back to device->stream without wavefront rays, for textured scenes, and uses
		fewer frames when the sets of buffers would not fit) */
	uint32_t last_frame_in_flight;
	/*! per frame in flight: buffers of the wavefront ray path (ray queues, term
		streams, base colour), completion events */
	void* wavefront;
	/*! device counter of the rays traced inside the shading kernel (inline_rays) */
	void* ray_counter;
	/*! textured scenes: 8 floats per pixel sampled from the material textures before shading */
	void* pixel_materials;
	size_t pixel_materials_size;
	/*! timing of the last dispatch in milliseconds (HIP events on device->stream) */
	float last_dispatch_ms;
	/*! ring of HIP event pairs, one pair per timed render_shading_pass call */
	void* timing_ring;
	uint32_t timing_ring_size, timing_cursor;
	/*! time every timing_stride-th frame only (0 or 1: every frame; set before
		create_shading_pass like arithmetic_mode); frames rendered so far */
	uint32_t timing_stride, frame_counter;
	/*! 1 if the most recent render_shading_pass() traced shadow rays (an error display frame
		does not, whatever the settings say): get_last_ray_count() reports 0 otherwise */
	uint32_t last_frame_traced_rays;
	/*! 1: the wavefront kernel walks the binary tree even if the scene has the four-wide one
		(acceleration_structure_t.wide_nodes), for comparisons; results are identical.  Set before
		create_shading_pass like arithmetic_mode. */
	int32_t binary_traversal;
	/*! hipEvent_t or NULL: the next render_shading_pass() makes the stream it runs the frame on wait
		for this event before its first kernel, then clears the field.  For callers that hand the
		frame a target which earlier work on another stream still reads (the slab exchange): only the
		pass knows which stream the frame will take. */
	void* wait_before_next_frame;
	/*! hipStream_t the most recent render_shading_pass() queued its (last) kernels on
		(device->stream or one of the frame streams) */
	void* last_frame_stream;
	/*! Frames with wavefront shadow rays are rendered in `band_count` launches over consecutive parts of
		the frame ("bands"), each shaded, traced and resolved with wavefront buffers sized for the band;
		bands overlap on the frame streams like frames do.  0 (default): as few bands as keep all sets of
		buffers in flight within the budget (36 GiB, environment VKR_WAVEFRONT_BUDGET_MIB) - one for
		1920x1080 frames, three (12 GB each) for BASELINE config 4.  Set before create_shading_pass like
		arithmetic_mode.  last_band_count: what the most recent frame used. */
	uint32_t band_count, last_band_count;
	/*! Light shafts (csrc/light_shafts.h): before a launch with wavefront shadow rays one conservative walk of the
		BVH per 8x8 pixel patch and light finds the pairs whose shadow rays cannot be blocked by anything - their
		terms are final at once and their rays are never queued - and the pairs whose rays can only meet a handful of
		triangles (an occluder list, at most 12): the shading kernel decides those rays itself, with the tracing kernel's
		triangle test.  Results of ray queries - and frames - are unchanged.  Environment VKR_LIGHT_SHAFTS: 0 off, 1 on,
		default automatic - on when a pixel may queue 8 rays or more (samples x techniques x lights), where the walks
		repay themselves; VKR_SHAFT_LISTS=0: no occluder lists.  A verdict is a hint (a pair without one has its rays
		traced, with the same result), so a frame context keeps the verdicts of its previous frame: a pair whose walk met
		more triangles than a list holds is not walked again for seven of that context's frames (VKR_SHAFT_REST=0: every
		pair in every frame), and get_light_shaft_statistics() counts such pairs under "anything else".  Launches of less
		than 12 288 patches walk at most 12 steps (+ 2 per light) instead of 40 (+ 8), VKR_SHAFT_MAX_STEPS overrides.
		last_shaft_groups: patches of the most recent launch that were tested (0: the launch ran without the test). */
	uint32_t last_shaft_groups, reserved;
	/*! pinned staging, copy stream and events of begin_read_back() / end_read_back() (internal) */
	void* readback;
} shading_pass_t;

/*! The slice of reference application_t (main.h:440-476) that the pass uses */
/*! The textures of polygonal lights (reference: images_t light_textures, main.h:472, filled by
	create_and_assign_light_textures main.c:371-417).  One entry per distinct
	polygonal_light_t.texture_file_path; only the finest level is kept, as RGBA fp32, because the
	shader reads light textures at level 0 (shading_pass.frag.glsl:182). */
typedef struct light_textures_s {
	uint32_t texture_count;
	/*! per texture: index of its first texel, width, height, 0.  Width 0 means white (the
		reference's data/white.vkt fallback is built in). */
	uint32_t (*host_descriptors)[4];
	float* host_texels;
	uint64_t texel_count;
	/*! device copies */
	uint32_t* descriptors;
	float* texels;
} light_textures_t;

typedef struct application_s {
	device_t device;
	swapchain_t swapchain;
	scene_specification_t scene_specification;
	render_settings_t render_settings;
	scene_t scene;
	noise_table_t noise_table;
	ltc_table_t ltc_table;
	render_targets_t render_targets;
	screenshot_t screenshot;
	shading_pass_t shading_pass;
	tile_schedule_t tile_schedule;
	light_textures_t light_textures;
} application_t;

/*! reference main.c:232-249 */
VKR_API void specify_default_render_settings(render_settings_t* settings);
/*! reference main.c:134-168: default scene, camera and light, then quick_load() */
VKR_API void specify_default_scene(scene_specification_t* scene);
/*! reference main.c:173-216 */
VKR_API uint32_t get_min_polygonal_light_vertex_count(const scene_specification_t* scene_specification);
VKR_API uint32_t get_max_polygonal_light_vertex_count(const scene_specification_t* scene_specification);
VKR_API uint32_t get_max_polygon_vertex_count(const scene_specification_t* scene_specification, const render_settings_t* render_settings);
/*! reference main.c:219-229 */
VKR_API void destroy_scene_specification(scene_specification_t* scene);
/*! reference main.c:49-80 / :82-130.  `updates` of the reference is reduced to an
	optional flag that is set when light count or vertex counts changed. */
VKR_API void quick_save(scene_specification_t* scene);
VKR_API void quick_load(scene_specification_t* scene, VkBool32* light_count_changed);

/*! reference create_render_targets main.c:259-327 / destroy main.c:253-257 */
VKR_API int create_render_targets(render_targets_t* targets, const device_t* device, const swapchain_t* swapchain);
VKR_API void destroy_render_targets(render_targets_t* targets, const device_t* device);

/*! reference main.c:371-417 / :364-366.  Loads each distinct light texture once and writes
	polygonal_light_t.texture_index of every light; absent files and lights without a path get the
	built-in white texture.  light_textures may be NULL to update the indices only (that is what
	write_constants() does each frame, reference main.c:2167).  Create (or re-create) it before
	create_shading_pass() whenever a light uses a texturing technique other than none. */
VKR_API int create_and_assign_light_textures(light_textures_t* light_textures, const device_t* device, scene_specification_t* scene_specification);
VKR_API void destroy_light_textures(light_textures_t* light_textures, const device_t* device);

/*! Size in bytes of what write_constants() writes: sizeof(per_frame_constants_t)
	plus the packed light array (reference create_constant_buffers main.c:330-360) */
VKR_API size_t get_constant_buffer_size(const application_t* app);
/*! reference main.c:2114-2188: byte-identical output (the cursor position, which
	the reference reads from GLFW, is written as 0,0) */
VKR_API void write_constants(void* data, application_t* app);

/*! reference main.c:598-913: validates the settings the way the GUI does
	(user_interface.cpp:90-180), picks the kernel variant and allocates the constant
	buffer.  Returns 0 on success, 1 on failure (message printed, pass destroyed). */
VKR_API int create_shading_pass(shading_pass_t* pass, application_t* app);
/*! reference main.c:588-595 */
VKR_API void destroy_shading_pass(shading_pass_t* pass, const device_t* device);

/*! Replaces the visibility sub-pass (reference main.c:1421-1427,
	visibility_pass.*.glsl): closest hit through pixel centres with back-face
	culling against the LBVH.  Requires an acceleration structure. */
VKR_API int render_visibility_pass(application_t* app);
/*! Replaces sub-pass 1 of record_render_frame_commands (reference main.c:1428-1434):
	write_constants -> upload -> one launch over the tiles of app->tile_schedule.
	`out_radiance` NULL writes app->render_targets.radiance (full frame layout when
	rank_count == 1, slab layout otherwise). */
VKR_API int render_shading_pass(application_t* app, void* out_radiance);
/*! render_shading_pass() followed, on the same stream (the frame's own stream when frames are
	in flight), by the output encoding of the frame or slab as packed RGB8 (encode_slab_rgb8): the
	form in which ranks exchange their slabs.  Keeps the encoding out of device->stream, where
	it would sit behind the waits for earlier frames' collectives. */
VKR_API int render_shading_pass_encoded(application_t* app, void* out_radiance, void* out_rgb8);
/*! The hipStream_t on which the next render_shading_pass() will run if it is a frame in flight
	like the last one (else device->stream): for callers that chain their own work (a collective,
	a consumer) behind one frame without involving device->stream */
VKR_API void* get_next_frame_stream(const application_t* app);
/*! Makes device->stream wait (on the device, the host does not block) for the frames
	that are still in flight; a no-op without frames_in_flight >= 2 */
VKR_API int finish_frames(application_t* app);
/*! Tell a pass with frames in flight that work queued on device->stream by someone else
	(not through this library) has changed its inputs - visibility buffer, mesh, tables */
VKR_API void mark_inputs_changed(application_t* app);
/*! Number of pixels / floats of one rank's slab for the current schedule */
VKR_API uint64_t get_slab_pixel_count(const application_t* app, uint32_t rank);
/*! Pixel (x, y) of every slot of rank `rank`'s slab (0xFFFFFFFF for padding), so a
	host can scatter gathered slabs itself.  Returns the slot count; writes at most
	`capacity` slots. */
VKR_API uint64_t get_slab_pixel_coordinates(const application_t* app, uint32_t rank, uint32_t* out_xy, uint64_t capacity);
/*! Scatters the all-gathered slabs (rank-major, each padded to
	get_slab_pixel_count(app, 0) pixels) back into a row-major frame */
VKR_API int assemble_frame_from_slabs(application_t* app, const void* gathered_slabs, void* out_radiance);
/*! The same for slabs of the encoded frame (RGBA8, 4 bytes per pixel): what the
	reference's pass actually outputs, and a quarter of the bytes to exchange */
VKR_API int assemble_encoded_frame_from_slabs(application_t* app, const void* gathered_slabs, void* out_encoded);
/*! Output encoding (as encode_output) of `pixel_count` pixels of a slab or any other
	RGBA32F device buffer into an RGBA8 device buffer */
VKR_API int encode_slab(application_t* app, const void* slab_radiance, void* slab_encoded, uint64_t pixel_count, VkBool32 output_linear_rgb);
/*! The same as packed RGB8, three bytes per pixel (the alpha of the encoded output is always
	255): the smallest lossless form of the pass's output, for the exchange between GPUs.
	pixel_count must be a multiple of 4 (slabs are multiples of 256). */
VKR_API int encode_slab_rgb8(application_t* app, const void* slab_radiance, void* slab_rgb8, uint64_t pixel_count, VkBool32 output_linear_rgb);
/*! Scatters all-gathered RGB8 slabs into the RGBA8 frame (out_encoded NULL: render_targets.encoded) */
VKR_API int assemble_rgb8_frame_from_slabs(application_t* app, const void* gathered_slabs, void* out_encoded);
/*! Output encoding of shading_pass.frag.glsl:871-892 into render_targets.encoded */
VKR_API int encode_output(application_t* app, VkBool32 output_linear_rgb);
/*! Synchronous copies to host memory (implement_screenshot, main.c:1601-1631) */
VKR_API int read_back_radiance(application_t* app, float* host_rgba);
VKR_API int read_back_encoded(application_t* app, uint8_t* host_rgba8);
VKR_API int read_back_visibility(application_t* app, uint32_t* host_primitives);
/*! Asynchronous read-back through pinned staging (round 6).  The reference's screenshot path maps a host-visible image
	and waits for the device (implement_screenshot, main.c:1719-1770); a host that wants EVERY frame cannot afford
	that - read_back_radiance() into pageable memory runs at 12 GB/s and stops the frame pipeline, 2.8 ms for a
	1920x1080 RGBA32F frame of 1.15 ms.  begin_read_back() queues the copy of `bytes` bytes at `device_source` (NULL: the
	whole radiance target) into the pinned staging buffer of `slot` (0 ... VKR_MAX_FRAMES_IN_FLIGHT) on a copy stream
	of the pass, ordered behind the most recent render_shading_pass() / render_and_exchange_frame() - and behind
	app->device.stream, for sources produced there (encode_output) - and returns at once; the frames that follow
	keep running.  end_read_back() waits for that copy and returns the staging memory (valid until the slot's next
	begin_read_back(); NULL on failure).  A later frame that WRITES the same device buffer (the resolve kernel of the
	wavefront path, the shading kernel otherwise) waits for the copy on the device by itself; callers that want no
	such wait render into a ring of targets (render_shading_pass(app, target[i])) and read them back slot by slot. */
VKR_API int begin_read_back(application_t* app, uint32_t slot, const void* device_source, uint64_t bytes);
VKR_API const void* end_read_back(application_t* app, uint32_t slot);
/*! Upload a visibility buffer produced elsewhere (tests, external rasteriser) */
VKR_API int upload_visibility(application_t* app, const uint32_t* host_primitives);
/*! GPU time of the last render_shading_pass launch in milliseconds, measured with
	HIP events on the device's stream (blocks until the launch has finished) */
VKR_API float get_last_dispatch_milliseconds(application_t* app);
/*! Durations of the most recent `count` timed launches (oldest first, at most 256 are
	kept; see timing_stride).  Returns how many were written. */
VKR_API uint32_t get_dispatch_milliseconds(application_t* app, float* out_milliseconds, uint32_t count);
/*! Durations of the shading kernel alone (the dominant kernel of the pass) in the most
	recent `count` timed launches, HIP events on the stream it ran on */
VKR_API uint32_t get_shading_kernel_milliseconds(application_t* app, float* out_milliseconds, uint32_t count);
/*! Milliseconds a timed frame spends BEFORE its shading kernel starts: the light shaft kernel (and, for a textured
	scene, the material resolve).  Most recent `count` timed frames, oldest first; returns how many were written. */
VKR_API uint32_t get_light_shaft_milliseconds(application_t* app, float* out_milliseconds, uint32_t count);
/*! Time from the end of one timed launch to the end of the next, divided by the number of
	frames in between (timing_stride): the frame period when frames are submitted back to
	back, which is what matters with frames in flight, where a launch's own duration
	(get_dispatch_milliseconds) includes the time it shares the GPU with its neighbour.
	Most recent `count` intervals, oldest first; returns how many were written. */
VKR_API uint32_t get_frame_period_milliseconds(application_t* app, float* out_milliseconds, uint32_t count);
/*! Number of shadow rays the last render_shading_pass traced (0 when the variant
	was built without counters) */
VKR_API uint64_t get_last_ray_count(const application_t* app);

/*! Diagnostics for profiling: replays the shadow rays queued by the last wavefront frame
	and writes {rays, node visits, triangle tests, blocked rays, wave steps (sum over
	groups of 64 rays of the longest ray's visits), longest ray's visits}.  0 on success. */
VKR_API int get_traversal_statistics(application_t* app, uint64_t out_statistics[6]);
/*! Light shafts of the most recent launch: {(patch, light) pairs tested, pairs found clear - no shadow ray queued
	for them -, patches, lights, then the pairs whose rays are traced by reason: patch without a shaded pixel, light and
	patch do not form a shaft (light behind the patch, seen edge-on, too close), walk too long, queue full, too many
	triangles in the way, other; then the pairs with an occluder list - their rays are decided by the shading kernel -
	and the triangles on those lists}; all zero when the launch ran without the shaft test.  0 on success. */
VKR_API int get_light_shaft_statistics(application_t* app, uint64_t out_statistics[12]);
/*! (diagnostics) The verdicts themselves, one word per patch and light ([patch][light]; patches in the order of the
	shading workgroups): the low byte is 1 = clear, 2 = occluder list (bits 8 ... 12: its length) or 16 ... 20 = the
	reasons above; with 20 (triangles in the way) bits 8 ... 31 name one such triangle (its index in the mesh, if below
	2^24).  Returns the number of words written. */
VKR_API uint64_t read_back_light_shafts(application_t* app, uint32_t* out_words, uint64_t capacity);
/*! (diagnostics, only with VKR_SHAFT_COUNTERS=1 in the environment) work of the shaft kernel of the most recent
	launch: {steps of its walks (16 nodes each), batches of triangles, walks}.  0 on success. */
VKR_API int get_light_shaft_work(application_t* app, uint64_t out_work[3]);
/*! The same for the tree of the caller's choice - the binary one (one box per visit) or the
	four-wide one (a visit fetches one node and tests up to four boxes) - whatever the frame itself
	walked, plus (wide tree only) [6] boxes tested, [7] the deepest stack a ray reached, [8] rays whose stack
	outgrows the part the tracing kernel keeps in LDS (they take its spill path), [9] node visits of the rays
	that end up blocked; [10], [11] reserved (0) */
VKR_API int get_traversal_statistics_of_tree(application_t* app, VkBool32 wide_tree, uint64_t out_statistics[12]);

/*! Diagnostics for the arithmetic contract of the kernels (csrc/device_math.h, mirrored by
	oracle/oracle_math.h): evaluates one primitive of the IEEE arithmetic modes element-wise on the
	device.  operation 0: divide(a, b), 1: square_root(a), 2: rsqrt(a) of the polynomial mode, 3: the
	compiler's IEEE a / b, 4: the compiler's IEEE sqrtf(a); the functions of the libm mode
	(csrc/glibc_math.h): 5 atanf, 6 acosf, 7 sinf, 8 cosf, 9 log2f, 10 powf(a, b), 11 atan2f(a, b),
	12 inversesqrt as 1 / sqrt (inverse_square_root_ieee).  a, b (may be NULL for unary operations) and out are host arrays of
	`count` floats.  0 on success. */
VKR_API int evaluate_device_arithmetic(const device_t* device, uint32_t operation, const float* a, const float* b, float* out, uint32_t count);
/*! Two one-argument operations of evaluate_device_arithmetic (1 square_root, 4 the compiler's sqrtf, 5 atanf,
	12 inversesqrt as the kernels evaluate it, plus 16: the compiler's 1 / sqrtf, 17: atanf with its argument
	range from the LDS table, as the libm kernels evaluate it) evaluated on the device
	for the `count` (<= 2^32) consecutive bit patterns from `first_bits` on - e.g. every float - and
	compared bit for bit (NaNs equal each other).  out[0]: arguments with different results, out[1]: the
	smallest such bit pattern (all ones if none).  How a cheaper chain is admitted into the IEEE modes. */
VKR_API int compare_device_arithmetic(const device_t* device, uint32_t operation_a, uint32_t operation_b, uint32_t first_bits, uint64_t count, uint64_t out_mismatches_and_first[2]);
/*! divide(a, b) of the IEEE arithmetic modes (csrc/device_math.h: one correction behind the refined
	v_rcp_f32 estimate) against the compiler's IEEE a / b, on the device, for `divisor_count` (<= 65535)
	divisors with the significands first_significand + i * stride (23 bits) and the biased exponent
	divisor_exponent, each with EVERY one of the 2^23 dividends of the biased exponent dividend_exponent.
	out[0]: quotients that differ, out[1]: the smallest divisor_bits << 32 | dividend_bits among them (all
	ones if none).  A slice of the search over all 2^46 pairs of significands that admitted the chain
	(profiles/tools/division_chains.hip), kept in the test suite.  No reference counterpart (the reference
	leaves division to the driver's compiler). */
/*! Diagnostics (profiles/tools/predict_scaling.py): copies `bytes` bytes (a multiple of 16) from one device buffer to another
	with `workgroups` workgroups of 256 threads on hipStream_t `stream` - a copy that occupies a few compute units for a while, as
	the kernels of a collective do, instead of the whole GPU for a moment, as hipMemcpyAsync does. */
/*! Diagnostics: the kernels that keep a polygon table in device memory find their region by the hardware slot the wave runs in
	(csrc/shading_kernel.h hardware_wave_slot); this launches `workgroups` single-wave workgroups on two streams at once, each of
	which claims its slot, works for a while and releases it.  out[0]: waves that found their slot claimed by a wave that was
	still running (must be 0), out[1]: distinct slots that were used.  extra_lds_bytes: dynamic LDS per workgroup (varies how many
	waves fit a CU). */
VKR_API int check_hardware_wave_slots(const device_t* device, uint32_t workgroups, uint32_t extra_lds_bytes, uint64_t out_shared_and_used[2]);
VKR_API int copy_with_workgroups(void* destination, const void* source, uint64_t bytes, uint32_t workgroups, void* stream);
VKR_API int compare_device_division(const device_t* device, uint32_t first_significand, uint32_t divisor_count, uint32_t stride, uint32_t dividend_exponent, uint32_t divisor_exponent, uint64_t out_mismatches_and_first[2]);

/*! Writes sizeof() of every ABI struct (device_t, polygonal_light_t,
	first_person_camera_t, ltc_constants_t, ltc_table_t, noise_table_t, mesh_t,
	materials_t, acceleration_structure_t, scene_t, scene_specification_t,
	render_settings_t, per_frame_constants_t, swapchain_t, render_targets_t,
	screenshot_t, tile_schedule_t, shading_pass_t, application_t, experiment_t,
	experiment_list_t, slab_exchange_id_t, slab_exchange_t) so that bindings can
	check their mirrors.  Returns the number of structs. */
VKR_API uint32_t get_abi_struct_sizes(uint64_t* sizes, uint32_t capacity);

#endif
