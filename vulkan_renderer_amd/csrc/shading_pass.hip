// Host side of the shading pass: variant selection, constant upload, launches,
// output encoding and read-back behind the reference's entry points
// (create_shading_pass src/main.c:598, write_constants :2114, the vkCmdDraw of
// record_render_frame_commands :1428-1434, implement_screenshot :1719).
#include "wavefront_kernels.h"
#include "light_shafts.h"
#include "host/vkr_internal.h"
#include <hip/hip_fp16.h>

using namespace vkr;

#define VKR_DECLARE_LAUNCH(mode, s) extern "C" int vkr_launch_shade_##mode##_##s(int technique, int capacity, int rays, const shade_params* p, unsigned int grid_x, void* stream);
#define VKR_DECLARE_LAUNCHES(mode) VKR_DECLARE_LAUNCH(mode, 0) VKR_DECLARE_LAUNCH(mode, 1) VKR_DECLARE_LAUNCH(mode, 2) VKR_DECLARE_LAUNCH(mode, 3) VKR_DECLARE_LAUNCH(mode, 4)
VKR_DECLARE_LAUNCHES(libm) VKR_DECLARE_LAUNCHES(fast) VKR_DECLARE_LAUNCHES(exact)
VKR_DECLARE_LAUNCHES(textured_libm) VKR_DECLARE_LAUNCHES(textured_fast) VKR_DECLARE_LAUNCHES(textured_exact)

typedef int (*error_launch_function_t)(int combined_path, int technique, int capacity, int error_mode, const shade_params* p, unsigned int grid_x, void* stream);
extern "C" int vkr_launch_error_display_libm(int combined_path, int technique, int capacity, int error_mode, const shade_params* p, unsigned int grid_x, void* stream);
extern "C" int vkr_launch_error_display_fast(int combined_path, int technique, int capacity, int error_mode, const shade_params* p, unsigned int grid_x, void* stream);
extern "C" int vkr_launch_error_display_exact(int combined_path, int technique, int capacity, int error_mode, const shade_params* p, unsigned int grid_x, void* stream);
typedef int (*resolve_launch_function_t)(const shade_params* p, float* pixel_materials, void* stream);
extern "C" int vkr_launch_resolve_materials_libm(const shade_params* p, float* pixel_materials, void* stream);
extern "C" int vkr_launch_resolve_materials_fast(const shade_params* p, float* pixel_materials, void* stream);
extern "C" int vkr_launch_resolve_materials_exact(const shade_params* p, float* pixel_materials, void* stream);
typedef int (*launch_function_t)(int, int, int, const shade_params*, unsigned int, void*);
// [arithmetic_mode_t + 3 * light textures][strategy]
#define VKR_LAUNCHER_ROW(mode) {vkr_launch_shade_##mode##_0, vkr_launch_shade_##mode##_1, vkr_launch_shade_##mode##_2, vkr_launch_shade_##mode##_3, vkr_launch_shade_##mode##_4}
static const launch_function_t g_launchers[6][5] = {
	VKR_LAUNCHER_ROW(libm), VKR_LAUNCHER_ROW(fast), VKR_LAUNCHER_ROW(exact),
	VKR_LAUNCHER_ROW(textured_libm), VKR_LAUNCHER_ROW(textured_fast), VKR_LAUNCHER_ROW(textured_exact),
};
static const error_launch_function_t g_error_launchers[3] = {vkr_launch_error_display_libm, vkr_launch_error_display_fast, vkr_launch_error_display_exact};
static const resolve_launch_function_t g_resolve_launchers[3] = {vkr_launch_resolve_materials_libm, vkr_launch_resolve_materials_fast, vkr_launch_resolve_materials_exact};

// Events that order streams of one device: a device-scope release is all they need.  The default
// (system-scope fence: L2 write-back and invalidation at every record) is paid by whatever runs
// next on the device, and a frame records several.
constexpr unsigned kSyncEventFlags = hipEventDisableTiming | hipEventReleaseToDevice;

static int hip_failed(hipError_t error, const char* what) {
	if (error == hipSuccess) return 0;
	printf("HIP error while %s: %s\n", what, hipGetErrorString(error));
	return 1;
}

static bvh_view make_bvh_view(const acceleration_structure_t* structure) {
	bvh_view view;
	view.nodes = (const uint4*) structure->nodes;
	view.triangles = (const float4*) structure->triangle_vertices;
	view.node_count = structure->node_count;
	view.grid_origin = f3{structure->grid_origin[0], structure->grid_origin[1], structure->grid_origin[2]};
	view.grid_inverse_cell = f3{structure->grid_inverse_cell[0], structure->grid_inverse_cell[1], structure->grid_inverse_cell[2]};
	return view;
}

static int technique_index(const render_settings_t* settings) {
	switch (settings->polygon_sampling_technique) {
	// Arvo's sampler only exists in the diffuse-only / GGX-MIS branch of the reference shader; its
	// combined diffuse + specular branch uses the paper's own sampler whatever the technique says
	// (shading_pass.frag.glsl:441, :506-547)
	case sample_polygon_projected_solid_angle_arvo:
		return settings->sampling_strategies >= sampling_strategies_diffuse_specular_separately ? kTechniquePsa : kTechniquePsaArvo;
	case sample_polygon_projected_solid_angle: return kTechniquePsa;
	case sample_polygon_projected_solid_angle_biased: return kTechniquePsaBiased;
	case sample_polygon_solid_angle: return kTechniqueSolidAngle;
	case sample_polygon_clipped_solid_angle: return kTechniqueClippedSolidAngle;
	case sample_polygon_baseline: return kTechniqueBaseline;
	case sample_polygon_area_turk: return kTechniqueAreaTurk;
	case sample_polygon_rectangle_solid_angle_urena: return kTechniqueUrena;
	case sample_polygon_solid_angle_arvo: return kTechniqueArvoSolidAngle;
	case sample_polygon_bilinear_cosine_warp_hart: return kTechniqueHartBilinear;
	case sample_polygon_bilinear_cosine_warp_clipping_hart: return kTechniqueHartBilinearClipping;
	case sample_polygon_biquadratic_cosine_warp_hart: return kTechniqueHartBiquadratic;
	case sample_polygon_biquadratic_cosine_warp_clipping_hart: return kTechniqueHartBiquadraticClipping;
	default: return -1;
	}
}

// ---- render targets --------------------------------------------------------------

extern "C" void destroy_render_targets(render_targets_t* targets, const device_t* device) {
	vkr_device_free(targets->visibility_buffer, device);
	vkr_device_free(targets->radiance, device);
	vkr_device_free(targets->encoded, device);
	memset(targets, 0, sizeof(*targets));
}

extern "C" int create_render_targets(render_targets_t* targets, const device_t* device, const swapchain_t* swapchain) {
	memset(targets, 0, sizeof(*targets));
	if (!device) {
		printf("Render targets live in device memory; a HIP device is required.\n");
		return 1;
	}
	size_t pixels = (size_t) swapchain->extent.width * swapchain->extent.height;
	if (pixels == 0) return 2;  // reference main.c:1865: a minimised window is not an error
	targets->extent = swapchain->extent;
	// slabs are padded to whole tiles, so leave room for one extra row and column of 64-pixel tiles
	size_t padded = ((size_t) swapchain->extent.width + 64) * ((size_t) swapchain->extent.height + 64);
	if (vkr_device_alloc(&targets->visibility_buffer, device, sizeof(uint32_t) * pixels, "the visibility buffer")
		|| vkr_device_alloc(&targets->radiance, device, sizeof(float) * 4 * padded, "the radiance target")
		|| vkr_device_alloc(&targets->encoded, device, 4 * pixels, "the encoded output"))
	{
		destroy_render_targets(targets, device);
		return 1;
	}
	hipStream_t stream = (hipStream_t) device->stream;
	if (hip_failed(hipMemsetAsync(targets->visibility_buffer, 0xFF, sizeof(uint32_t) * pixels, stream), "clearing the visibility buffer")) {
		destroy_render_targets(targets, device);
		return 1;
	}
	return 0;
}

// ---- shading pass ----------------------------------------------------------------

// events of a timed frame: its start, the start and the end of the (last band's) shading kernel, its end
constexpr uint32_t kTimingEvents = 4;

// device counter of traced shadow rays, shared by all passes of the process

// Buffers of the wavefront ray path, sized for the worst case (every sample of every
// light on every pixel produces a term and a ray) and owned by the pass.
struct wavefront_buffers {
	uint8_t* codes;
	float* terms_visible;
	float* terms_hidden;
	float4* base_color;
	float4* ray_directions;
	uint32_t* ray_records;
	float4* ray_origins;
	uint32_t* ray_queue_size;  // kRayCounterCount live counters (queue sizes, per-XCD work cursors), then last frame's copy
	uint32_t thread_count, max_terms, max_codes, queue_capacity, thread_bits;
	// the streams that only some settings need are allocated when they first do: values of blocked
	// terms (only the plain optimal MIS heuristic has non-zero ones), colour before the sampled terms
	// (only the light display has one)
	bool has_hidden_terms, has_base_color;
	// stack entries beyond the LDS part of trace_shadow_rays_wide, [entry][thread of the trace grid];
	// allocated only for trees that can need them
	uint32_t* spill;
	size_t spill_entries;
	// light shafts (light_shafts.h): one word per shading workgroup and light, 1 = no ray of that patch toward that
	// light can be blocked; allocated when the feature first runs
	uint32_t* shaft_clear;
	// which launch the verdicts in shaft_clear belong to (blocks, frame size, tiling, lights): the shaft kernel only leans on
	// them when the next launch with these buffers is the same one (a band of another part of the frame is not)
	uint64_t shaft_tag;
	size_t shaft_words;
	// ... and per light the plane-space rectangle that the shading kernel tests its rays against
	float4* shaft_rectangles;
	uint32_t shaft_rectangle_count;
	// ... and per pair kShaftListMax triangle slots: the occluder lists (VKR_SHAFT_LISTS)
	float* shaft_lists;
	size_t shaft_list_words;
	// the second polygon table of every shading workgroup, for the kernel variants that keep one table in LDS only
	// (shading_kernel.h psa_table_in_memory); allocated when such a variant first runs
	float2* psa_table_memory;
	size_t psa_table_bytes;
};

static void free_wavefront_buffers(wavefront_buffers* w) {
	(void) hipFree(w->codes); (void) hipFree(w->terms_visible); (void) hipFree(w->terms_hidden);
	(void) hipFree(w->base_color); (void) hipFree(w->ray_directions); (void) hipFree(w->ray_records); (void) hipFree(w->ray_origins); (void) hipFree(w->ray_queue_size);
	(void) hipFree(w->spill);
	(void) hipFree(w->shaft_clear);
	(void) hipFree(w->shaft_rectangles);
	(void) hipFree(w->shaft_lists);
	(void) hipFree(w->psa_table_memory);
	memset(w, 0, sizeof(*w));
}

// What a frame in flight owns.  With frames_in_flight = n >= 2 consecutive frames take turns
// on n contexts (and n of the device's frame streams); otherwise only context 0 is used, on
// device->stream.
struct frame_context {
	wavefront_buffers buffers;
	// VKR_TRACE_STREAM_PRIORITY=high (experiment of round 3, profiles/r03_trace.md): the tracing and the
	// resolve kernel of a launch run on a stream of their own with the highest priority, behind an event
	// that marks the end of the shading kernel
	hipStream_t trace_stream;
	hipEvent_t shaded;
	hipEvent_t done;  // recorded behind the last kernel of the frame
	bool recorded;    // `done` has been recorded: the next frame's resolve is ordered behind it
	// what the context's most recent launches wrote their output to (a ring of the last eight): a launch only has to be ordered
	// behind another one when they write the same buffer (frames of a slab exchange take turns on several slabs, round 6); the
	// context's `done` event lies behind all of them
	const void* targets[8];
	uint32_t target_cursor;
	bool pending;     // device->stream has not been made to wait for `done` yet
	uint32_t readers_seen;  // frame_pipeline::readers_generation this context's stream has waited for
};
struct frame_pipeline {
	frame_context contexts[VKR_MAX_FRAMES_IN_FLIGHT];
	hipEvent_t inputs_ready;  // marks what device->stream had submitted when a frame started
	// Recorded on device->stream behind every kernel there that reads a target of the frames
	// (output encoding of the frame or of a slab): a later frame in flight must not resolve into
	// that target before the reader is done.  (finish_frames() orders device->stream behind the
	// frames; this is the opposite direction.)
	hipEvent_t readers_done;
	uint32_t readers_generation;
	// bumped whenever an input that the frames read from device memory has been rewritten (visibility buffer, scene):
	// part of the light shafts' tag
	uint32_t inputs_generation;
	// the polygon tables in device memory (wavefront_buffers::psa_table_memory) of the frames without wavefront rays, which
	// own no context (and, with VKR_PSA_TABLE_INDEX=slot, of all launches of the pass: regions by hardware wave slot)
	wavefront_buffers device_stream_buffers;
	uint32_t next;            // context of the next pipelined frame
	uint32_t last;            // context of the most recent frame
	uint32_t depth;           // frames in flight of the most recent pipelined frame
	// tuning / test knobs, read from the environment once when the pipeline is created:
	// VKR_WIDE_STACK_LDS (stack entries per lane that trace_shadow_rays_wide keeps in LDS: tests shrink
	// it to drive rays through the spill path), VKR_LEAF_BATCH (lanes that must have a triangle waiting
	// before a wave tests triangles), VKR_REFILL_THRESHOLD (binary walk)
	// VKR_TRACE_WAVES: persistent waves per SIMD of the tracing kernels (1 ... 8; 0: eight where a lane
	// queues eight rays or more, four otherwise - measured at config 2, whose 0.9 M rays are a batch or
	// two per wave: 0.129 -> 0.121 ms per frame)
	// VKR_TRACE_SINGLE_WAVES: 0 / 1 overrides the choice of the tracing kernel's workgroup size (2: automatic)
	// VKR_WAVEFRONT_BUDGET_MIB: most device memory that all sets of wavefront buffers in flight may take
	// (default 36864 - config 4 then runs as three bands of 12 GB, the fastest of 1 ... 12 bands, profiles/r04c/: a
	// frame whose worst case needs more is rendered in bands); VKR_BAND_COUNT forces
	// the number of bands per frame (0: automatic)
	// VKR_LIGHT_SHAFTS: 0 turns the shaft test off (every shadow ray is traced, as until round 3), 1 on; default 2:
	// on when a pixel may queue 8 rays or more (samples x techniques x lights) - the walk of a patch costs about as
	// much as tracing 2.5 rays per pixel and light, and it is the patches with many rays per light and several lights
	// that repay it (measured, profiles/r05m: config 3, 32 rays per pixel, 1.553 -> 1.443 ms; config 4, 128, 25.9 -> 23.0;
	// the target shape, 8, 0.488 -> 0.500 before the occluder lists and 0.501 -> 0.443 with them, which is what moved
	// the rule from 16 to 8; config 2, 2 rays per pixel, 0.127 -> 0.192)
	// VKR_SHAFT_REST, VKR_SHAFT_MAX_STEPS, VKR_WIDE_REFILL, VKR_WIDE_REFILL_BELOW (round 5): ensure_frames()
	uint32_t wide_stack_lds, leaf_batch, refill_threshold, wide_refill, wide_refill_below, trace_waves, trace_single_waves, wavefront_budget_mib, band_count, light_shafts, shaft_lists, shaft_rest, shaft_max_steps;
};

static uint32_t environment_knob(const char* name, uint32_t fallback, uint32_t low, uint32_t high) {
	const char* text = getenv(name);
	if (!text || !text[0]) return fallback;
	long value = strtol(text, NULL, 10);
	return (uint32_t) (value < (long) low ? (long) low : (value > (long) high ? (long) high : value));
}

static void destroy_wavefront(shading_pass_t* pass) {
	frame_pipeline* frames = (frame_pipeline*) pass->wavefront;
	if (!frames) return;
	for (frame_context& c : frames->contexts) {
		if (c.done) { (void) hipEventSynchronize(c.done); (void) hipEventDestroy(c.done); }
		if (c.trace_stream) { (void) hipStreamSynchronize(c.trace_stream); (void) hipStreamDestroy(c.trace_stream); }
		if (c.shaded) (void) hipEventDestroy(c.shaded);
		free_wavefront_buffers(&c.buffers);
	}
	free_wavefront_buffers(&frames->device_stream_buffers);
	if (frames->inputs_ready) (void) hipEventDestroy(frames->inputs_ready);
	if (frames->readers_done) (void) hipEventDestroy(frames->readers_done);
	free(frames);
	pass->wavefront = NULL;
}

static frame_pipeline* ensure_frames(shading_pass_t* pass) {
	frame_pipeline* frames = (frame_pipeline*) pass->wavefront;
	if (frames) return frames;
	frames = (frame_pipeline*) calloc(1, sizeof(frame_pipeline));
	pass->wavefront = frames;
	bool failed = !frames || hipEventCreateWithFlags(&frames->inputs_ready, kSyncEventFlags) != hipSuccess
		|| hipEventCreateWithFlags(&frames->readers_done, kSyncEventFlags) != hipSuccess;
	for (int i = 0; i != VKR_MAX_FRAMES_IN_FLIGHT && !failed; ++i) failed = hipEventCreateWithFlags(&frames->contexts[i].done, kSyncEventFlags) != hipSuccess;
	if (failed) {
		printf("Failed to create the events of the frame pipeline.\n");
		destroy_wavefront(pass);
		return NULL;
	}
	frames->wide_stack_lds = environment_knob("VKR_WIDE_STACK_LDS", kWideStackLds, 4u, kWideStackLds);
	frames->light_shafts = environment_knob("VKR_LIGHT_SHAFTS", 2u, 0u, 2u);
	// VKR_SHAFT_LISTS=0: a shaft walk ends at the first triangle in the way (no occluder lists, light_shafts.h)
	frames->shaft_lists = environment_knob("VKR_SHAFT_LISTS", 1u, 0u, 1u);
	// VKR_SHAFT_REST: frames (of a frame context) for which a pair is not walked again after a walk that met more triangles
	// than a list holds; 0: every pair is walked in every frame (light_shafts.h, kShaftResting)
	frames->shaft_rest = environment_knob("VKR_SHAFT_REST", kShaftRestFrames, 0u, 200u);
	// VKR_SHAFT_MAX_STEPS: steps after which a walk gives up (plus a fifth of it per light that is walked along).  The shaft
	// kernel of a small launch - a rank's slab at N = 8 - lasts as long as its longest walk.
	// (0: by the size of the launch - kShaftMaxSteps, or kShaftSmallLaunchSteps below 12 288 shading waves)
	frames->shaft_max_steps = environment_knob("VKR_SHAFT_MAX_STEPS", 0u, 0u, 1000u);
	frames->leaf_batch = environment_knob("VKR_LEAF_BATCH", 16u, 1u, 64u);
	frames->refill_threshold = environment_knob("VKR_REFILL_THRESHOLD", 0u, 0u, 64u);
	// VKR_WIDE_REFILL: lanes of a tracing wave (four-wide tree) that have to be idle before they are handed the next rays,
	// once the wave has found its batches less than VKR_WIDE_REFILL_BELOW / 256 busy (wavefront_kernels.h; 256: from the
	// first batch on); 0: a batch of 64 rays is always walked to its end first (until round 4)
	frames->wide_refill = environment_knob("VKR_WIDE_REFILL", kWideRefillLanes, 0u, 64u);
	frames->wide_refill_below = environment_knob("VKR_WIDE_REFILL_BELOW", kWideRefillBelow, 0u, 256u);
	frames->trace_waves = environment_knob("VKR_TRACE_WAVES", 0u, 0u, 8u);
	frames->trace_single_waves = environment_knob("VKR_TRACE_SINGLE_WAVES", 2u, 0u, 2u);
	frames->wavefront_budget_mib = environment_knob("VKR_WAVEFRONT_BUDGET_MIB", 36864u, 64u, 262144u);
	frames->band_count = environment_knob("VKR_BAND_COUNT", 0u, 0u, 4096u);
	const char* priority = getenv("VKR_TRACE_STREAM_PRIORITY");
	if (priority && strcmp(priority, "high") == 0) {
		int least = 0, greatest = 0;
		(void) hipDeviceGetStreamPriorityRange(&least, &greatest);
		for (frame_context& c : frames->contexts)
			if (hipStreamCreateWithPriority(&c.trace_stream, hipStreamNonBlocking, greatest) != hipSuccess || hipEventCreateWithFlags(&c.shaded, kSyncEventFlags) != hipSuccess) {
				printf("Failed to create the high-priority tracing streams.\n");
				destroy_wavefront(pass);
				return NULL;
			}
	}
	return frames;
}

// Slots a shading wave reserves per atomic (shade_params.ray_block): pays off when a lane
// queues many rays; with one or two per lane the unused slots would outnumber the rays.
// VKR_RAY_BLOCK (a multiple of 64, read once per process): the slots per block; unused slots of a wave's last
// block become null rays, a contiguous run that the tracing kernel skips a batch at a time.
static uint32_t ray_block_size(uint32_t max_terms) {
	static uint32_t slots = 0;
	if (!slots) {
		const char* text = getenv("VKR_RAY_BLOCK");
		long value = text ? strtol(text, NULL, 10) : 0;
		slots = (value >= 64 && value <= 8192 && value % 64 == 0) ? (uint32_t) value : 256u;
	}
	return max_terms >= 8 ? slots : 0u;
}

static int ensure_shaft_words(wavefront_buffers* w, size_t words, uint32_t light_count, bool lists, hipStream_t stream) {
	size_t list_words = lists ? words * kShaftListMax * kShaftListEntry : 0;
	if (list_words > w->shaft_list_words) {
		(void) hipFree(w->shaft_lists);
		w->shaft_lists = NULL; w->shaft_list_words = 0;
		if (hipMalloc(&w->shaft_lists, list_words * sizeof(float)) != hipSuccess) {
			printf("Failed to allocate %.1f MiB for the occluder lists of the light shafts.\n", list_words * 4.0 / 1048576.0);
			return 1;
		}
		w->shaft_list_words = list_words;
	}
	if (light_count > w->shaft_rectangle_count) {
		(void) hipFree(w->shaft_rectangles);
		w->shaft_rectangles = NULL; w->shaft_rectangle_count = 0;
		if (hipMalloc(&w->shaft_rectangles, sizeof(float4) * light_count) != hipSuccess) {
			printf("Failed to allocate the rectangles of the light shafts.\n");
			return 1;
		}
		w->shaft_rectangle_count = light_count;
	}
	if (words <= w->shaft_words) return 0;
	(void) hipFree(w->shaft_clear);
	w->shaft_clear = NULL; w->shaft_words = 0; w->shaft_tag = 0;
	if (hipMalloc(&w->shaft_clear, words * sizeof(uint32_t)) != hipSuccess) {
		printf("Failed to allocate %.1f MiB for the light shafts.\n", words * 4.0 / 1048576.0);
		return 1;
	}
	// (the shaft kernel reads the verdicts of the frame before: none yet)
	if (hipMemsetAsync(w->shaft_clear, 0, words * sizeof(uint32_t), stream) != hipSuccess) return 1;
	w->shaft_words = words;
	return 0;
}

static int ensure_psa_table_memory(wavefront_buffers* w, size_t bytes) {
	if (bytes <= w->psa_table_bytes) return 0;
	// (frees while other frames may be in flight: hipFree waits for the device)
	(void) hipFree(w->psa_table_memory);
	w->psa_table_memory = NULL; w->psa_table_bytes = 0;
	if (hipMalloc(&w->psa_table_memory, bytes) != hipSuccess) {
		printf("Failed to allocate %.1f MiB for the polygon tables that do not fit into LDS.\n", bytes / 1048576.0);
		return 1;
	}
	w->psa_table_bytes = bytes;
	return 0;
}

static int ensure_spill(wavefront_buffers* w, uint32_t stack_need, uint32_t in_lds, uint32_t trace_threads) {
	size_t entries = stack_need > in_lds ? (size_t) (stack_need - in_lds) * trace_threads : 0;
	if (entries <= w->spill_entries) return 0;
	// (frees while other frames may be in flight: hipFree waits for the device)
	(void) hipFree(w->spill);
	w->spill = NULL; w->spill_entries = 0;
	if (hipMalloc(&w->spill, entries * sizeof(uint32_t)) != hipSuccess) {
		printf("Failed to allocate %.1f MiB for the traversal stacks that do not fit into LDS.\n", entries * 4.0 / 1048576.0);
		return 1;
	}
	w->spill_entries = entries;
	return 0;
}

// Bytes per term slot of the streams every frame needs (visible value 12, code 1) and per ray slot (20)
static uint32_t queue_capacity_for(uint32_t thread_count, uint32_t max_terms) {
	// a queue sees every 512th wave (8 XCDs x 64 queues, waves dealt round-robin), every
	// lane of which may emit max_terms rays
	return ((thread_count / 64 + kRayQueueCount - 1) / kRayQueueCount + 1) * (64u * max_terms + ray_block_size(max_terms));
}

// bytes that ensure_wavefront() allocates for a launch of thread_count threads
// (with the light shafts' table: a verdict word per 8x8 patch and light and, with occluder lists, kShaftListMax entries each)
static double wavefront_bytes(uint32_t thread_count, uint32_t max_terms, uint32_t light_count, bool hidden_terms, bool base_color, uint32_t table_bytes_per_thread = 0) {
	double terms = (double) max_terms * thread_count;
	double shaft_pairs = (double) (thread_count / 64u) * light_count;
	return terms * (hidden_terms ? 24.0 : 12.0) + (double) ((max_terms + light_count + 2 + 3) & ~3u) * thread_count + (base_color ? 16.0 : 0.0) * thread_count
		+ 16.0 * thread_count + (double) queue_capacity_for(thread_count, max_terms) * kRayQueueCount * 20.0
		+ shaft_pairs * (4.0 + 4.0 * kShaftListMax * kShaftListEntry) + (double) table_bytes_per_thread * thread_count;
}

// `stream`: the stream the frame that uses these buffers is about to run on.  The counters are cleared
// THERE: the frame streams are non-blocking, i.e. not ordered behind a hipMemset on the null stream, and
// a frame that started before that memset landed had its queue sizes reset under its feet (found in
// round 3: the first frame of a fresh context lost a few rays).
static int ensure_wavefront(wavefront_buffers* w, uint32_t thread_count, uint32_t max_terms, uint32_t light_count, bool hidden_terms, bool base_color, hipStream_t stream) {
	uint32_t max_codes = max_terms + light_count + 2;
	size_t terms = (size_t) max_terms * thread_count;
	if (w->codes && w->thread_count == thread_count && w->max_terms == max_terms && w->max_codes == max_codes) {
		// (frames in flight may still use the other streams of this context: allocating does not disturb them)
		if (hidden_terms && !w->has_hidden_terms) {
			if (hipMalloc(&w->terms_hidden, terms * 12) != hipSuccess) { printf("Failed to allocate %.1f MiB for the values of blocked terms.\n", terms * 12.0 / 1048576.0); return 1; }
			w->has_hidden_terms = true;
		}
		if (base_color && !w->has_base_color) {
			if (hipMalloc(&w->base_color, sizeof(float4) * (size_t) thread_count) != hipSuccess) { printf("Failed to allocate the colours of the light display.\n"); return 1; }
			w->has_base_color = true;
		}
		return 0;
	}
	free_wavefront_buffers(w);
	w->thread_count = thread_count; w->max_terms = max_terms; w->max_codes = max_codes;
	// the record word of a ray: thread and code cursor (shading_kernel.h ray_record)
	uint32_t thread_bits = 1;
	while (thread_bits < 32 && (1ull << thread_bits) < thread_count) ++thread_bits;
	if (terms >= 0xFFFFFFFFull || (size_t) (max_codes + 3u) * thread_count >= 0xFFFFFFFFull || thread_bits >= 32 || ((uint64_t) max_codes << thread_bits) > 0xFFFFFFFFull) {
		printf("The wavefront ray queue would need more than 2^32 entries (%u threads x %u terms); render in more bands or use inline rays.\n", thread_count, max_terms);
		return 1;
	}
	w->thread_bits = thread_bits;
	w->queue_capacity = queue_capacity_for(thread_count, max_terms);
	size_t ray_slots = (size_t) w->queue_capacity * kRayQueueCount;
	// (codes are stored four to a word per thread: code_slot() in shading_kernel.h)
	if (hipMalloc(&w->codes, (size_t) ((max_codes + 3u) & ~3u) * thread_count) != hipSuccess
		|| hipMalloc(&w->terms_visible, terms * 12) != hipSuccess
		|| (hidden_terms && hipMalloc(&w->terms_hidden, terms * 12) != hipSuccess)
		|| (base_color && hipMalloc(&w->base_color, sizeof(float4) * (size_t) thread_count) != hipSuccess)
		|| hipMalloc(&w->ray_directions, ray_slots * 16) != hipSuccess
		|| hipMalloc(&w->ray_records, ray_slots * 4) != hipSuccess
		|| hipMalloc(&w->ray_origins, sizeof(float4) * (size_t) thread_count) != hipSuccess
		|| hipMalloc(&w->ray_queue_size, sizeof(uint32_t) * 2 * kRayCounterCount) != hipSuccess
		|| hipMemsetAsync(w->ray_queue_size, 0, sizeof(uint32_t) * 2 * kRayCounterCount, stream) != hipSuccess)
	{
		printf("Failed to allocate %.1f MiB for the wavefront ray queue and term streams.\n", wavefront_bytes(thread_count, max_terms, light_count, hidden_terms, base_color) / 1048576.0);
		free_wavefront_buffers(w);
		return 1;
	}
	w->has_hidden_terms = hidden_terms;
	w->has_base_color = base_color;
	return 0;
}

extern "C" void mark_inputs_changed(application_t* app) {
	app->shading_pass.inputs_changed = 1;
	// (no pipeline yet: no verdict of an earlier arrangement exists either)
	if (app->shading_pass.wavefront) ++((frame_pipeline*) app->shading_pass.wavefront)->inputs_generation;
}

// The stream the next render_shading_pass() will run on if it is pipelined the way the last
// frame was (frame stream `next` of the pipeline), else device->stream
extern "C" void* get_next_frame_stream(const application_t* app) {
	const frame_pipeline* frames = (const frame_pipeline*) app->shading_pass.wavefront;
	if (!frames || !app->shading_pass.last_frame_in_flight || !frames->depth) return app->device.stream;
	return app->device.frame_streams[frames->next % frames->depth];
}

extern "C" int finish_frames(application_t* app) {
	frame_pipeline* frames = (frame_pipeline*) app->shading_pass.wavefront;
	if (!frames) return 0;
	int failed = 0;
	// (only device->stream's view changes: the order among the frames themselves is kept by
	// frame_context::recorded, which stays set)
	for (frame_context& c : frames->contexts)
		if (c.pending) {
			failed |= hip_failed(hipStreamWaitEvent((hipStream_t) app->device.stream, c.done, 0), "waiting for a frame in flight");
			c.pending = false;
		}
	return failed;
}

// Call behind a kernel on device->stream that reads a buffer frames in flight write (the radiance
// target, a caller's slab): the next frames wait for it before they resolve.
static void note_target_reader(application_t* app) {
	frame_pipeline* frames = (frame_pipeline*) app->shading_pass.wavefront;
	if (!frames || !app->shading_pass.last_frame_in_flight) return;
	if (hipEventRecord(frames->readers_done, (hipStream_t) app->device.stream) == hipSuccess) ++frames->readers_generation;
}

// The constant buffer is a small ring: the host may record several frames ahead, so
// every set of constants in flight needs its own staging and device copy (the reference
// keeps one uniform buffer per swapchain image for the same reason, main.c:330-360).
// A frame whose constants are byte-identical to the previous frame's reuses the slot
// that is already on the device (static camera and lights: no upload at all, like the
// reference's host-coherent uniform buffer, which costs no GPU time either).
constexpr uint32_t kConstantSlots = VKR_MAX_FRAMES_IN_FLIGHT + 2;
// (device->stream, the frame streams and - VKR_TRACE_STREAM_PRIORITY=high - the tracing streams, whose resolve
// kernels read the exposure)
constexpr int kConstantReaders = 1 + 2 * VKR_MAX_FRAMES_IN_FLIGHT;
struct constants_ring {
	void* host[kConstantSlots];
	void* device[kConstantSlots];
	// a slot may be read from device->stream and from the frame streams
	hipEvent_t consumed[kConstantSlots][kConstantReaders];
	hipEvent_t uploaded[kConstantSlots];
	// bit i: readers[i] (device->stream, then the frame streams) is ordered behind the upload
	uint32_t ordered[kConstantSlots];
	bool in_flight[kConstantSlots];
	void* scratch;    // write_constants target before it is known whether anything changed
	uint32_t current; // slot whose device copy the next launch reads
	bool valid;       // false until the first upload
};

static void destroy_constants_ring(shading_pass_t* pass, const device_t* device) {
	constants_ring* ring = (constants_ring*) pass->constants_ring;
	if (!ring) return;
	for (uint32_t i = 0; i != kConstantSlots; ++i) {
		vkr_device_free(ring->device[i], device);
		vkr_host_free_pinned(ring->host[i]);
		for (hipEvent_t event : ring->consumed[i]) if (event) (void) hipEventDestroy(event);
		if (ring->uploaded[i]) (void) hipEventDestroy(ring->uploaded[i]);
	}
	free(ring->scratch);
	free(ring);
	pass->constants_ring = NULL;
	pass->constants_device = pass->constants_host = NULL;
}

static int create_constants_ring(shading_pass_t* pass, const device_t* device) {
	constants_ring* ring = (constants_ring*) calloc(1, sizeof(constants_ring));
	pass->constants_ring = ring;
	if (!ring) return 1;
	ring->scratch = calloc(1, pass->constants_size);
	if (!ring->scratch) return 1;
	for (uint32_t i = 0; i != kConstantSlots; ++i) {
		if (vkr_device_alloc(&ring->device[i], device, pass->constants_size, "the constant buffer")
			|| vkr_host_alloc_pinned(&ring->host[i], pass->constants_size)
			|| hip_failed(hipEventCreateWithFlags(&ring->uploaded[i], kSyncEventFlags), "creating upload events"))
			return 1;
		for (hipEvent_t& event : ring->consumed[i])
			if (hip_failed(hipEventCreateWithFlags(&event, kSyncEventFlags), "creating upload events")) return 1;
		memset(ring->host[i], 0, pass->constants_size);
	}
	pass->constants_device = ring->device[0];
	pass->constants_host = ring->host[0];
	return 0;
}

// write_constants and, if the bytes changed, upload them into the next free slot on
// `stream`; in any case `stream` is made to wait for the upload of the slot it will read
static int upload_constants(application_t* app, hipStream_t stream) {
	shading_pass_t* pass = &app->shading_pass;
	constants_ring* ring = (constants_ring*) pass->constants_ring;
	// an entry that does not exist (no tracing streams) is marked by `present`: a NULL stream is a stream too
	hipStream_t readers[kConstantReaders] = {(hipStream_t) app->device.stream};
	bool present[kConstantReaders] = {true};
	const frame_pipeline* pipeline = (const frame_pipeline*) pass->wavefront;
	for (int i = 0; i != VKR_MAX_FRAMES_IN_FLIGHT; ++i) {
		readers[1 + i] = (hipStream_t) app->device.frame_streams[i];
		present[1 + i] = app->device.frame_streams[i] != NULL;
		hipStream_t tracing = pipeline ? pipeline->contexts[i].trace_stream : NULL;
		readers[1 + VKR_MAX_FRAMES_IN_FLIGHT + i] = tracing;
		present[1 + VKR_MAX_FRAMES_IN_FLIGHT + i] = tracing != NULL;
	}
	write_constants(ring->scratch, app);
	if (!ring->valid || memcmp(ring->scratch, ring->host[ring->current], pass->constants_size) != 0) {
		uint32_t slot = ring->valid ? (ring->current + 1) % kConstantSlots : 0;
		// everything launched so far may read the old slot: it is free again once all
		// streams have passed this point
		if (ring->valid) {
			// (an absent reader's event stays unrecorded, which completes at once)
			for (int i = 0; i != kConstantReaders; ++i) if (present[i]) (void) hipEventRecord(ring->consumed[ring->current][i], readers[i]);
			ring->in_flight[ring->current] = true;
		}
		if (ring->in_flight[slot])
			for (int i = 0; i != kConstantReaders; ++i)
				if (hip_failed(hipEventSynchronize(ring->consumed[slot][i]), "waiting for a free constant buffer")) return 1;
		ring->in_flight[slot] = false;
		memcpy(ring->host[slot], ring->scratch, pass->constants_size);
		if (hip_failed(hipMemcpyAsync(ring->device[slot], ring->host[slot], pass->constants_size, hipMemcpyHostToDevice, stream), "uploading the constants")
			|| hip_failed(hipEventRecord(ring->uploaded[slot], stream), "recording the upload"))
			return 1;
		ring->ordered[slot] = 0;
		for (int i = 0; i != kConstantReaders; ++i) if (present[i] && readers[i] == stream) ring->ordered[slot] |= 1u << i;
		ring->current = slot;
		ring->valid = true;
		pass->constants_device = ring->device[slot];
		pass->constants_host = ring->host[slot];
		return 0;
	}
	// unchanged constants that another stream uploaded: order this stream behind that upload
	for (int i = 0; i != kConstantReaders; ++i)
		if (present[i] && readers[i] == stream) {
			if (ring->ordered[ring->current] & (1u << i)) return 0;
			ring->ordered[ring->current] |= 1u << i;
		}
	return hip_failed(hipStreamWaitEvent(stream, ring->uploaded[ring->current], 0), "waiting for the constants");
}

// ---- asynchronous read-back through pinned staging (include/vkr_shading_pass.h begin_read_back) ----------------
// One staging buffer, one event and the device address it was filled from per slot; all copies run on one stream of their
// own (device-to-host copies into pinned memory are served by a DMA engine: they take no compute unit from the frames).
constexpr uint32_t kReadBackSlots = VKR_MAX_FRAMES_IN_FLIGHT + 1;
struct read_back_state {
	hipStream_t stream;
	hipEvent_t source_ready;               // marks device->stream when a copy is queued
	void* staging[kReadBackSlots];
	size_t staging_size[kReadBackSlots];
	hipEvent_t copied[kReadBackSlots];
	const void* source[kReadBackSlots];    // device range [source, source + bytes) of the slot's most recent copy
	size_t bytes[kReadBackSlots];
	bool pending[kReadBackSlots];          // the copy may still be running: a writer of its source waits for `copied`
	// readers outside the pass (vkr_note_target_reader: the slab exchange's collectives): an event of the caller and the
	// device range that is read until it completes
	hipEvent_t reader_event[kReadBackSlots];
	const void* reader_source[kReadBackSlots];
	size_t reader_bytes[kReadBackSlots];
};

// ... and before the caller destroys such an event
extern "C" void vkr_forget_target_reader(application_t* app, void* event) {
	read_back_state* rb = (read_back_state*) app->shading_pass.readback;
	if (!rb || !event) return;
	for (uint32_t i = 0; i != kReadBackSlots; ++i)
		if (rb->reader_event[i] == (hipEvent_t) event) rb->reader_event[i] = NULL;
	if (app->shading_pass.wait_before_next_frame == event) app->shading_pass.wait_before_next_frame = NULL;
}

static read_back_state* ensure_read_back_state(shading_pass_t* pass) {
	if (!pass->readback) pass->readback = calloc(1, sizeof(read_back_state));
	return (read_back_state*) pass->readback;
}

// For host/slab_exchange.c: `event` (a hipEvent_t of the caller, recorded behind a reader of [target, target + bytes)) must
// have completed before a later frame writes that range.  The frame that writes it waits on the device, in front of the
// kernel that does the writing - the resolve kernel of a frame with wavefront rays, the shading kernel otherwise, the
// encoding kernel for an encoded slab - and not in front of its first kernel, as shading_pass_t.wait_before_next_frame
// makes it: shaft walks, shading and tracing of frame k + n overlap the collective of frame k (round 6: a rank's slab at
// N = 8 took 0.220 instead of 0.176 ms through the exchange with a collective that did nothing,
// profiles/r10c/exchange_overhead_before.jsonl).  One entry per range; a new event for a known range replaces the old one.
extern "C" void vkr_note_target_reader(application_t* app, void* event, const void* target, size_t bytes) {
	read_back_state* rb = ensure_read_back_state(&app->shading_pass);
	if (!rb) return;
	uint32_t slot = kReadBackSlots;
	for (uint32_t i = 0; i != kReadBackSlots; ++i) {
		if (rb->reader_event[i] && rb->reader_source[i] == target) { slot = i; break; }
		if (!rb->reader_event[i] && slot == kReadBackSlots) slot = i;
	}
	if (slot == kReadBackSlots) {
		// (more ranges than buffer sets can exist: fall back to the oldest rule - the whole next frame waits)
		app->shading_pass.wait_before_next_frame = event;
		return;
	}
	rb->reader_event[slot] = (hipEvent_t) event;
	rb->reader_source[slot] = target;
	rb->reader_bytes[slot] = bytes;
}

static void destroy_read_back(shading_pass_t* pass) {
	read_back_state* rb = (read_back_state*) pass->readback;
	if (!rb) return;
	if (rb->stream) { (void) hipStreamSynchronize(rb->stream); (void) hipStreamDestroy(rb->stream); }
	if (rb->source_ready) (void) hipEventDestroy(rb->source_ready);
	for (uint32_t i = 0; i != kReadBackSlots; ++i) {
		if (rb->copied[i]) (void) hipEventDestroy(rb->copied[i]);
		vkr_host_free_pinned(rb->staging[i]);
	}
	free(rb);
	pass->readback = NULL;
}

// Makes `stream` wait for every pending copy that reads from [target, target + bytes): called in front of the kernel
// of a frame that writes its output (the resolve kernel with wavefront rays, the shading kernel otherwise)
static void wait_for_read_backs_of(shading_pass_t* pass, const void* target, size_t bytes, hipStream_t stream) {
	read_back_state* rb = (read_back_state*) pass->readback;
	if (!rb) return;
	for (uint32_t i = 0; i != kReadBackSlots; ++i) {
		if (!rb->reader_event[i]) continue;
		const uint8_t* a = (const uint8_t*) rb->reader_source[i];
		const uint8_t* b = (const uint8_t*) target;
		if (!(a < b + bytes && b < a + rb->reader_bytes[i])) continue;
		if (hipEventQuery(rb->reader_event[i]) == hipSuccess) { rb->reader_event[i] = NULL; continue; }
		(void) hipStreamWaitEvent(stream, rb->reader_event[i], 0);
	}
	for (uint32_t i = 0; i != kReadBackSlots; ++i) {
		if (!rb->pending[i]) continue;
		if (hipEventQuery(rb->copied[i]) == hipSuccess) { rb->pending[i] = false; continue; }
		const uint8_t* a = (const uint8_t*) rb->source[i];
		const uint8_t* b = (const uint8_t*) target;
		if (a < b + bytes && b < a + rb->bytes[i]) (void) hipStreamWaitEvent(stream, rb->copied[i], 0);
	}
}

extern "C" int begin_read_back(application_t* app, uint32_t slot, const void* device_source, uint64_t bytes) {
	shading_pass_t* pass = &app->shading_pass;
	if (slot >= kReadBackSlots) {
		printf("begin_read_back(): slot %u does not exist (0 ... %u).\n", slot, kReadBackSlots - 1u);
		return 1;
	}
	if (!device_source) {
		device_source = app->render_targets.radiance;
		bytes = sizeof(float) * 4 * (uint64_t) app->swapchain.extent.width * app->swapchain.extent.height;
	}
	if (!device_source || !bytes) {
		printf("begin_read_back() needs a device buffer (or render targets) to read from.\n");
		return 1;
	}
	read_back_state* rb = ensure_read_back_state(pass);
	if (!rb) return 1;
	if (!rb->stream && (hip_failed(hipStreamCreateWithFlags(&rb->stream, hipStreamNonBlocking), "creating the read-back stream")
		|| hip_failed(hipEventCreateWithFlags(&rb->source_ready, kSyncEventFlags), "creating read-back events")))
		return 1;
	// (the host waits for this one: no device-scope-only release)
	if (!rb->copied[slot] && hip_failed(hipEventCreateWithFlags(&rb->copied[slot], hipEventDisableTiming), "creating read-back events")) return 1;
	// (the slot's previous copy has to have landed before its staging memory is reused or freed)
	if (rb->pending[slot] && hip_failed(hipEventSynchronize(rb->copied[slot]), "waiting for the slot's previous read-back")) return 1;
	rb->pending[slot] = false;
	if (rb->staging_size[slot] < bytes) {
		vkr_host_free_pinned(rb->staging[slot]);
		rb->staging[slot] = NULL; rb->staging_size[slot] = 0;
		if (vkr_host_alloc_pinned(&rb->staging[slot], (size_t) bytes)) {
			printf("Failed to allocate %.1f MiB of pinned host memory for read-backs.\n", bytes / 1048576.0);
			return 1;
		}
		rb->staging_size[slot] = (size_t) bytes;
	}
	// behind the frames in flight (the most recent one completes last: resolves are chained) ...
	frame_pipeline* frames = (frame_pipeline*) pass->wavefront;
	if (frames && pass->last_frame_in_flight) {
		frame_context* last = &frames->contexts[frames->last];
		if (last->recorded && hip_failed(hipStreamWaitEvent(rb->stream, last->done, 0), "ordering the read-back behind the frame")) return 1;
	}
	// ... and behind what device->stream has queued (frames without the pipeline, output encoding, an assembled frame)
	if (hip_failed(hipEventRecord(rb->source_ready, (hipStream_t) app->device.stream), "marking the source")
		|| hip_failed(hipStreamWaitEvent(rb->stream, rb->source_ready, 0), "ordering the read-back behind the device stream")
		|| hip_failed(hipMemcpyAsync(rb->staging[slot], device_source, (size_t) bytes, hipMemcpyDeviceToHost, rb->stream), "queueing the read-back")
		|| hip_failed(hipEventRecord(rb->copied[slot], rb->stream), "marking the read-back"))
		return 1;
	rb->source[slot] = device_source;
	rb->bytes[slot] = (size_t) bytes;
	rb->pending[slot] = true;
	return 0;
}

extern "C" const void* end_read_back(application_t* app, uint32_t slot) {
	read_back_state* rb = (read_back_state*) app->shading_pass.readback;
	if (!rb || slot >= kReadBackSlots || !rb->staging[slot] || !rb->copied[slot]) {
		printf("end_read_back(): slot %u has no read-back in flight.\n", slot);
		return NULL;
	}
	if (hip_failed(hipEventSynchronize(rb->copied[slot]), "waiting for the read-back")) return NULL;
	rb->pending[slot] = false;
	return rb->staging[slot];
}

extern "C" void destroy_shading_pass(shading_pass_t* pass, const device_t* device) {
	// frames in flight still read the buffers that are freed below
	if (device && pass->wavefront) (void) wait_for_device(device);
	destroy_read_back(pass);
	destroy_constants_ring(pass, device);
	if (pass->ray_counter) (void) hipFree(pass->ray_counter);
	if (pass->pixel_materials) (void) hipFree(pass->pixel_materials);
	destroy_wavefront(pass);
	if (pass->timing_ring) {
		hipEvent_t* ring = (hipEvent_t*) pass->timing_ring;
		for (uint32_t i = 0; i != kTimingEvents * pass->timing_ring_size; ++i) if (ring[i]) (void) hipEventDestroy(ring[i]);
		free(ring);
	}
	memset(pass, 0, sizeof(*pass));
}

// The same legality rules the reference enforces in its GUI
// (src/user_interface.cpp:90-180), as hard errors.
static int validate_settings(const application_t* app) {
	const render_settings_t* s = &app->render_settings;
	const scene_specification_t* spec = &app->scene_specification;
	int technique = technique_index(s);
	if (technique < 0) {
		printf("Invalid polygon sampling technique %d.\n", (int) s->polygon_sampling_technique);
		return 1;
	}
	bool is_psa = technique == kTechniquePsa || technique == kTechniquePsaBiased || technique == kTechniquePsaArvo;
	// An out-of-range strategy selects none of the strategy defines of the reference, i.e.
	// the combined diffuse + specular preparation with no estimator behind it.  That is
	// only meaningful with an error display (the reference's own experiment table does it,
	// experiment_list.c:107); everything else is refused.
	bool strategy_in_range = s->sampling_strategies < sampling_strategies_count;
	if ((!strategy_in_range && !(s->error_display != error_display_none && is_psa)) || s->mis_heuristic >= mis_heuristic_count) {
		printf("Invalid sampling strategy or MIS heuristic.\n");
		return 1;
	}
	bool needs_specular = s->sampling_strategies == sampling_strategies_diffuse_specular_separately
		|| s->sampling_strategies == sampling_strategies_diffuse_specular_mis
		|| s->sampling_strategies == sampling_strategies_diffuse_specular_random;
	if ((technique == kTechniqueBaseline || technique == kTechniqueAreaTurk || technique == kTechniqueHartBilinear || technique == kTechniqueHartBilinearClipping
			|| technique == kTechniqueHartBiquadratic || technique == kTechniqueHartBiquadraticClipping)
		&& s->sampling_strategies != sampling_strategies_diffuse_only)
	{
		printf("The baseline, area sampling and cosine warp techniques only exist for the diffuse-only sampling strategy (as in the reference shader).\n");
		return 1;
	}
	if (needs_specular && !is_psa) {
		printf("Sampling strategies with LTC importance sampling require projected solid angle sampling.\n");
		return 1;
	}
	if ((s->mis_heuristic == mis_heuristic_weighted || s->mis_heuristic == mis_heuristic_optimal_clamped || s->mis_heuristic == mis_heuristic_optimal)
		&& s->sampling_strategies == sampling_strategies_diffuse_ggx_mis)
	{
		printf("The weighted and optimal MIS heuristics are only defined for the diffuse+specular MIS strategy.\n");
		return 1;
	}
	if (s->error_display >= error_display_count) {
		printf("Invalid error display mode.\n");
		return 1;
	}
	if (technique == kTechniquePsaArvo && s->error_display == error_display_diffuse_forward) {
		printf("Arvo's sampler only defines the backward errors (the reference shader does not compile with the forward error either).\n");
		return 1;
	}
	if (s->sample_count == 0) {
		printf("The sample count must be positive.\n");
		return 1;
	}
	for (uint32_t i = 0; i != spec->polygonal_light_count; ++i) {
		const polygonal_light_t* light = &spec->polygonal_lights[i];
		if (light->vertex_count < 3 || light->vertex_count > 7) {
			printf("Polygonal light %u has %u vertices; the clipping and sorting code covers 3 to 7.\n", i, light->vertex_count);
			return 1;
		}
		if (light->texturing_technique < 0 || light->texturing_technique >= polygon_texturing_count) {
			printf("Polygonal light %u has the invalid texturing technique %d.\n", i, (int) light->texturing_technique);
			return 1;
		}
	}
	return 0;
}

static int create_timing_ring(shading_pass_t* pass) {
	pass->timing_ring_size = 256;
	// per timed frame: start, end of the shading kernel, end of the frame
	hipEvent_t* ring = (hipEvent_t*) calloc(kTimingEvents * pass->timing_ring_size, sizeof(hipEvent_t));
	pass->timing_ring = ring;
	for (uint32_t i = 0; i != kTimingEvents * pass->timing_ring_size; ++i)
		if (hip_failed(hipEventCreateWithFlags(&ring[i], hipEventReleaseToDevice), "creating timing events")) return 1;
	return 0;
}

extern "C" int create_shading_pass(shading_pass_t* pass, application_t* app) {
	int32_t arithmetic_mode = pass->arithmetic_mode, inline_rays = pass->inline_rays, binary_traversal = pass->binary_traversal;
	uint32_t band_count = pass->band_count;
	void* wait_before_next_frame = pass->wait_before_next_frame;
	uint32_t timing_stride = pass->timing_stride, frames_in_flight = pass->frames_in_flight;
	memset(pass, 0, sizeof(*pass));
	pass->timing_stride = timing_stride;
	pass->frames_in_flight = frames_in_flight;
	pass->inputs_changed = 1;
	if (arithmetic_mode < 0 || arithmetic_mode >= arithmetic_mode_count) {
		printf("Invalid arithmetic mode %d (0: libm, 1: fast, 2: polynomial).\n", arithmetic_mode);
		return 1;
	}
	pass->arithmetic_mode = arithmetic_mode;
	pass->wait_before_next_frame = wait_before_next_frame;
	pass->band_count = band_count;
	pass->inline_rays = inline_rays ? 1 : 0;
	pass->binary_traversal = binary_traversal ? 1 : 0;
	pass->variant = -1;
	const device_t* device = &app->device;
	if (validate_settings(app)) return 1;
	pass->use_ray_tracing = app->render_settings.trace_shadow_rays && app->scene.acceleration_structure.triangle_vertices != NULL;
	if (app->render_settings.trace_shadow_rays && !pass->use_ray_tracing)
		printf("Shadow rays were requested but the scene has no acceleration structure; rendering without shadows.\n");
	pass->max_polygon_vertex_count = get_max_polygon_vertex_count(&app->scene_specification, &app->render_settings);
	pass->variant = (int32_t) app->render_settings.sampling_strategies * kTechniqueCount + technique_index(&app->render_settings);
	pass->constants_size = get_constant_buffer_size(app);
	if (create_constants_ring(pass, device) || create_timing_ring(pass))
	{
		printf("Failed to create the shading pass.\n");
		destroy_shading_pass(pass, device);
		return 1;
	}
	return 0;
}

static void fill_tile_schedule(shade_params& p, const application_t* app, uint32_t& grid_blocks) {
	tile_schedule_t schedule = app->tile_schedule;
	if (schedule.rank_count <= 1) { schedule.rank = 0; schedule.rank_count = 1; }
	// tile_size 0: automatic.  The blocks of a tile are consecutive in the launch, so the tile size decides which 16x16 blocks
	// are in flight together: raster order of blocks (tile 16) spreads the resident waves over a band of the whole frame width,
	// tiles of 64 keep them in compact squares.  Measured on one GPU, frames bit-identical (profiles/r10r/tile_order.jsonl,
	// tile 16 / 32 / 64 / 128): config 3 1.157 / 1.136 / 1.129 / 1.152 ms, large scene 3.96 / 3.90 / 3.90 / 3.95, target shape
	// 0.432 / 0.425 / 0.428 / 0.429, config 2 0.140 / 0.139 / 0.138 / 0.139; the 3840x2160 frame of config 4 16.33 / 16.59 /
	// 16.56 / 16.20: 64 up to three megapixels, 128 above.
	// (only where the tile size does not shape the output: one rank that renders in place)
	if (schedule.tile_size == 0 && schedule.rank_count == 1 && !schedule.slab_layout) schedule.tile_size = ((uint64_t) p.width * p.height <= 3145728ull) ? 64u : 128u;
	if (schedule.tile_size < 16) schedule.tile_size = 16;
	schedule.tile_size = (schedule.tile_size + 15) & ~15u;
	p.tile_size = schedule.tile_size;
	p.rank = schedule.rank;
	p.rank_count = schedule.rank_count;
	p.slab_layout = (schedule.rank_count > 1 || schedule.slab_layout) ? 1u : 0u;
	p.tiles_x = (p.width + p.tile_size - 1) / p.tile_size;
	uint32_t tiles_y = (p.height + p.tile_size - 1) / p.tile_size;
	p.tile_count = p.tiles_x * tiles_y;
	uint32_t own_tiles = (p.tile_count + p.rank_count - 1 - p.rank) / p.rank_count;
	uint32_t blocks_per_tile = (p.tile_size / 16) * (p.tile_size / 16);
	grid_blocks = own_tiles * blocks_per_tile;
}

extern "C" uint64_t get_slab_pixel_count(const application_t* app, uint32_t rank) {
	shade_params p;
	memset(&p, 0, sizeof(p));
	p.width = app->swapchain.extent.width;
	p.height = app->swapchain.extent.height;
	application_t copy = *app;
	copy.tile_schedule.rank = rank;
	uint32_t grid_blocks = 0;
	fill_tile_schedule(p, &copy, grid_blocks);
	return (uint64_t) grid_blocks * 256;
}

__global__ void k_encode_output_rgb8(const float4* radiance, uint32_t* packed, uint64_t quad_count, uint32_t frame_bits, int output_linear_rgb);

static int render_pass(application_t* app, void* out_radiance, void* out_rgb8);
extern "C" int get_traversal_statistics_of_tree(application_t* app, VkBool32 wide_tree, uint64_t out_statistics[8]);

extern "C" int render_shading_pass(application_t* app, void* out_radiance) {
	return render_pass(app, out_radiance, NULL);
}

extern "C" int render_shading_pass_encoded(application_t* app, void* out_radiance, void* out_rgb8) {
	if (!out_rgb8) {
		printf("render_shading_pass_encoded() needs a target for the encoded pixels.\n");
		return 1;
	}
	return render_pass(app, out_radiance, out_rgb8);
}

// out_rgb8: the frame (or slab) is also encoded as packed RGB8 on the stream it was rendered on
static int render_pass(application_t* app, void* out_radiance, void* out_rgb8) {
	shading_pass_t* pass = &app->shading_pass;
	const device_t* device = &app->device;
	if (pass->variant < 0 || !pass->constants_device) {
		printf("render_shading_pass() needs a shading pass created by create_shading_pass().\n");
		return 1;
	}
	// settings may have been edited since create_shading_pass(): the same legality rules apply, and the
	// launcher tables below are indexed with them
	if (validate_settings(app)) return 1;
	if (get_constant_buffer_size(app) != pass->constants_size
		|| get_max_polygon_vertex_count(&app->scene_specification, &app->render_settings) != pass->max_polygon_vertex_count
		|| (int32_t) app->render_settings.sampling_strategies * kTechniqueCount + technique_index(&app->render_settings) != pass->variant)
	{
		printf("Lights or render settings changed in a way that needs a different kernel variant. Recreate the shading pass (the reference recompiles its shader in this situation, main.c:1833-1881).\n");
		return 1;
	}
	hipStream_t stream = (hipStream_t) device->stream;
	shade_params p;
	memset(&p, 0, sizeof(p));
	p.light_count = app->scene_specification.polygonal_light_count;
	p.max_light_vertex_count = get_max_polygonal_light_vertex_count(&app->scene_specification);
	p.sample_count = app->render_settings.sample_count;
	p.mis_heuristic = (int32_t) app->render_settings.mis_heuristic;
	p.show_polygonal_lights = app->render_settings.show_polygonal_lights ? 1 : 0;
	p.positions = (const uint2*) app->scene.mesh.positions;
	p.normals_and_tex_coords = (const uint2*) app->scene.mesh.normals_and_tex_coords;
	p.material_indices = (const uint8_t*) app->scene.mesh.material_indices;
	p.material_constants = (const float*) app->scene.materials.constants;
	p.visibility = (const uint32_t*) app->render_targets.visibility_buffer;
	p.out_radiance = (float4*) (out_radiance ? out_radiance : app->render_targets.radiance);
	p.width = app->swapchain.extent.width;
	p.height = app->swapchain.extent.height;
	p.ltc_rgba = (const uint2*) app->ltc_table.device_rgba;
	p.ltc_rg = (const uint32_t*) app->ltc_table.device_rg;
	p.ltc_resolution = app->ltc_table.roughness_count;
	p.ltc_layer_count = app->ltc_table.fresnel_count;
	p.noise = (const uint2*) app->noise_table.device_data;
	p.noise_width = app->noise_table.resolution.width;
	p.noise_height = app->noise_table.resolution.height;
	p.bvh = make_bvh_view(&app->scene.acceleration_structure);
	if (!p.positions || !p.visibility || !p.out_radiance || !p.ltc_rgba || !p.noise || !p.material_constants) {
		printf("render_shading_pass() needs a loaded scene, LTC table, noise table and render targets on the device.\n");
		return 1;
	}
	if (p.width != app->render_targets.extent.width || p.height != app->render_targets.extent.height) {
		printf("The render targets do not match the swapchain extent.\n");
		return 1;
	}
	uint32_t grid_blocks = 0;
	fill_tile_schedule(p, app, grid_blocks);
	int ray_mode = !pass->use_ray_tracing ? kRaysNone : (pass->inline_rays ? kRaysInline : kRaysDeferred);
	// Error display (ERROR_DISPLAY_DIFFUSE / _SPECULAR / ERROR_INDEX, main.c:728-750): only
	// the projected solid angle paths look at these flags, and the specular display
	// exists only where the specular technique is prepared.  The program returns
	// before it samples, so no ray is ever traced.
	int error_mode = kErrorNone;
	{
		int display = (int) app->render_settings.error_display;
		int technique_now = technique_index(&app->render_settings);
		bool combined = app->render_settings.sampling_strategies >= sampling_strategies_diffuse_specular_separately;
		if (display != error_display_none && (technique_now == kTechniquePsa || technique_now == kTechniquePsaBiased || technique_now == kTechniquePsaArvo)) {
			bool specular = display == error_display_specular_backward || display == error_display_specular_backward_scaled || display == error_display_specular_forward;
			error_mode = specular ? (combined ? kErrorSpecular : kErrorNone) : kErrorDiffuse;
		}
		if (error_mode != kErrorNone) {
			ray_mode = kRaysNone;
			p.error_index = (display == error_display_diffuse_backward || display == error_display_specular_backward) ? 0u
				: ((display == error_display_diffuse_backward_scaled || display == error_display_specular_backward_scaled) ? 1u : 2u);
			// the constants that the GLSL compiler folds in error_to_color (shading_pass.frag.glsl:81-89)
			p.error_max = powf(10.0f, 5.0f - 0.01f);
			p.error_scale = 20.0f / ((5.0f - 0.0f) * log2f(10.0f));
		}
	}
	if (error_mode == kErrorNone && (int) app->render_settings.sampling_strategies >= (int) sampling_strategies_count) {
		// (an out-of-range strategy is only legal together with an error display that is really shown:
		// a specular display without the combined path shows nothing and would index past the launcher table)
		printf("No kernel variant exists for sampling strategy %d without an error display.\n", (int) app->render_settings.sampling_strategies);
		return 1;
	}
	if (pass->use_ray_tracing) {
		// (sixteen counters, one per frame in turn: see the band loop)
		if (!pass->ray_counter && hip_failed(hipMalloc(&pass->ray_counter, 16 * sizeof(unsigned long long)), "allocating the ray counters")) return 1;
		p.ray_counter = (unsigned long long*) pass->ray_counter;
	}
	int strategy = (int) app->render_settings.sampling_strategies;
	int technique = technique_index(&app->render_settings);
	bool is_clipped = technique == kTechniquePsa || technique == kTechniquePsaBiased || technique == kTechniqueClippedSolidAngle || technique == kTechniqueHartBilinearClipping || technique == kTechniqueHartBiquadraticClipping || technique == kTechniquePsaArvo;
	int capacity = (int) p.max_light_vertex_count + (is_clipped ? 1 : 0);
	// the kernel variants that keep one of their two polygon tables in device memory (shading_kernel.h psa_table_in_memory)
	const bool table_in_memory = strategy >= kStrategySeparately && (technique == kTechniquePsa || technique == kTechniquePsaBiased) && error_mode == kErrorNone && psa_table_in_memory(capacity);
	const uint32_t table_bytes_per_workgroup = table_in_memory ? psa_table_memory_bytes_per_workgroup(capacity) : 0u;
	// Launches with wavefront rays may run n at a time: launch k on frame stream k mod n with
	// its own buffers, so that the (latency-bound) tracing of one launch overlaps the
	// (VALU-bound) shading of the next ones.  Everything else runs on device->stream, behind
	// any launch that is still in flight.
	// A frame is one launch - or several, "bands" of consecutive 16x16 blocks of the rank's schedule,
	// when the wavefront buffers of the whole frame (sized for the worst case: every sample of every
	// light on every pixel queues a ray) would be larger than the budget: a band is shaded, traced and
	// resolved like a small frame, with buffers sized for the band, and the bands of one frame - and of
	// the next frames - overlap on the frame streams exactly like whole frames do.
	bool pipelined = false;
	// the tracing kernels are persistent: 8 waves per SIMD on every CU, each lane strides over the queues
	// (8 by default; the count is a knob of the frame pipeline)
	const uint32_t compute_units = (uint32_t) (app->device.compute_unit_count > 0 ? app->device.compute_unit_count : 256);
	uint32_t trace_blocks = compute_units * 8u;
	const bool use_wide_tree = app->scene.acceleration_structure.wide_nodes && !pass->binary_traversal;
	const uint32_t max_terms = 2u * p.light_count * p.sample_count;
	if (ray_mode == kRaysDeferred && (int) app->render_settings.sampling_strategies >= (int) sampling_strategies_diffuse_specular_separately && ray_block_size(max_terms))
		ray_mode = kRaysDeferredBlocks;
	// values of blocked terms exist for the plain optimal heuristic only (its estimate is not proportional
	// to the integrand); a colour before the sampled terms only with the light display
	const bool hidden_terms = app->render_settings.mis_heuristic == mis_heuristic_optimal && (int) app->render_settings.sampling_strategies == (int) sampling_strategies_diffuse_specular_mis;
	const bool base_color = p.show_polygonal_lights != 0;
	uint32_t band_count = 1, blocks_per_band = grid_blocks, depth = 1;
	frame_pipeline* frames = NULL;
	if (is_deferred(ray_mode)) {
		frames = ensure_frames(pass);
		if (!frames) return 1;
		// (a textured scene has one per-pixel material buffer: one frame at a time)
		depth = pass->frames_in_flight < VKR_MAX_FRAMES_IN_FLIGHT ? pass->frames_in_flight : VKR_MAX_FRAMES_IN_FLIGHT;
		// (the device creates four frame streams; a deeper pipeline gets the others now)
		if (depth >= 2 && !device->frame_streams[depth - 1] && vkr_ensure_frame_streams(&app->device, depth)) return 1;
		while (depth >= 2 && !device->frame_streams[depth - 1]) --depth;
		if (depth < 1) depth = 1;
		pipelined = depth >= 2 && !app->scene.materials.textured;
		if (!pipelined) depth = 1;
		// Bands: as few as keep all sets of buffers in flight within the budget, each at least
		// kMinBandBlocks blocks (a launch has to fill the GPU several times over), whole groups of 8 blocks
		// (shade_grid_size).  pass->band_count / VKR_BAND_COUNT force a number.
		const double budget = (double) frames->wavefront_budget_mib * 1048576.0;
		const uint32_t kMinBandBlocks = 4096;
		uint32_t wanted = pass->band_count ? pass->band_count : frames->band_count;
		if (!wanted) {
			wanted = 1;
			while (depth * wavefront_bytes(((grid_blocks + wanted - 1) / wanted) * 256u, max_terms, p.light_count, hidden_terms, base_color, table_bytes_per_workgroup / 64u) > budget
				&& (grid_blocks + wanted) / (wanted + 1) >= kMinBandBlocks)
				++wanted;
		}
		blocks_per_band = (((grid_blocks + wanted - 1) / wanted) + 7u) & ~7u;
		if (blocks_per_band == 0) blocks_per_band = 8;
		band_count = (grid_blocks + blocks_per_band - 1) / blocks_per_band;
		if (band_count == 0) band_count = 1;
		if (!pipelined && finish_frames(app)) return 1;
		if (pipelined && frames->depth != depth) {
			// another pipeline depth: contexts and streams pair up differently, start afresh
			if (frames->depth && wait_for_device(device)) return 1;
			for (frame_context& c : frames->contexts) c.recorded = c.pending = false;
			frames->depth = depth;
			frames->next = 0;
		}
		if (pipelined && pass->inputs_changed) {
			// Inputs that were produced on device->stream (visibility pass, uploads): all frame
			// streams wait for them once.  Frames do not wait for anything else on
			// device->stream - if they did, a consumer of frame k there would hold back frame k + 1.
			if (hip_failed(hipEventRecord(frames->inputs_ready, (hipStream_t) device->stream), "marking the inputs")) return 1;
			for (uint32_t i = 0; i != depth; ++i)
				if (hip_failed(hipStreamWaitEvent((hipStream_t) device->frame_streams[i], frames->inputs_ready, 0), "waiting for the inputs")) return 1;
			pass->inputs_changed = 0;
		}
		// Small launches (round 5, profiles/r07f): a rank's slab at N = 8 is 4 080 shading waves on 3 072 wave slots.  Its
		// kernels then last about as long as their slowest wave, and what is launched for a whole frame - 8 192 tracing
		// waves for 230 k rays, shaft walks of up to 72 steps - is mostly waiting: with 2 tracing waves per SIMD and walks
		// that give up after 12 steps (more rays traced, a shorter chain of kernels) the slab of config 3 takes 0.177
		// instead of 0.199 ms, the target shape's 0.070 instead of 0.084; a quarter of the frame (8 160 waves) 0.316
		// instead of 0.333.  The whole frame (32 640 waves) loses with either: 1.150 -> 1.176 ms with 12 steps.
		const uint32_t launch_waves = shade_grid_size(blocks_per_band);
		const uint32_t small_launch_waves = frames->trace_waves ? 0u : (launch_waves < 6144u ? 2u : (launch_waves < 12288u ? 4u : 0u));
		trace_blocks = compute_units * (frames->trace_waves ? frames->trace_waves : (small_launch_waves ? small_launch_waves : (max_terms >= 8u ? 8u : 4u)));
		// (queues of XCD x are only served by workgroups b with b % 8 == x)
		trace_blocks = (trace_blocks + 7u) & ~7u;
		p.ray_block = ray_mode == kRaysDeferredBlocks ? ray_block_size(max_terms) : 0u;
		p.refill_threshold = frames->refill_threshold;
	}
	else {
		if (finish_frames(app)) return 1;
		// (no wavefront buffers, but the table in device memory lives with them: context 0)
		if (table_in_memory && !(frames = ensure_frames(pass))) return 1;
	}
	pass->last_frame_traced_rays = ray_mode != kRaysNone;
	pass->last_frame_in_flight = pipelined ? depth : 0u;
	pass->last_band_count = band_count;
	// textured scene: sample the material textures of every pixel first (same stream)
	bool textured = app->scene.materials.textured && app->scene.materials.texture_descriptors;
	if (textured) {
		size_t needed = sizeof(float) * 8 * (size_t) p.width * p.height;
		if (pass->pixel_materials_size != needed) {
			if (wait_for_device(device)) return 1;
			(void) hipFree(pass->pixel_materials);
			pass->pixel_materials = NULL;
			pass->pixel_materials_size = 0;
			if (hip_failed(hipMalloc(&pass->pixel_materials, needed), "allocating the per-pixel materials")) return 1;
			pass->pixel_materials_size = needed;
		}
		p.texture_descriptors = (const uint32_t*) app->scene.materials.texture_descriptors;
		p.texels = (const uint32_t*) app->scene.materials.texels;
		p.srgb_table = (const float*) app->scene.materials.srgb_table;
	}
	// light textures: bound whenever a light asks for one (the technique lives in the constants)
	for (uint32_t i = 0; i != app->scene_specification.polygonal_light_count; ++i) {
		const polygonal_light_t* light = &app->scene_specification.polygonal_lights[i];
		if (light->texturing_technique == polygon_texturing_none) continue;
		if (!app->light_textures.descriptors || light->texture_index >= app->light_textures.texture_count) {
			printf("Polygonal light %u uses a texture but the light textures have not been created for the current lights. Call create_and_assign_light_textures() first.\n", i);
			return 1;
		}
		p.light_texture_descriptors = (const uint4*) app->light_textures.descriptors;
		p.light_texels = (const float4*) app->light_textures.texels;
	}
	// every timing_stride-th frame is bracketed by events: start, end of the (last band's) shading
	// kernel, end of the frame (an event record costs about 5 us of idle time on the stream, a tenth
	// of a config-2 frame for the pair)
	hipEvent_t* ring = (hipEvent_t*) pass->timing_ring;
	uint32_t slot = pass->timing_cursor % pass->timing_ring_size;
	bool timed = pass->timing_stride <= 1 || pass->frame_counter % pass->timing_stride == 0;
	// rays of this frame: one of sixteen counters, taken in turn, so that the bands of this frame never
	// meet those of a frame that is still in flight (at most eight launches are)
	if (p.ray_counter) p.ray_counter += pass->frame_counter % 16u;
	++pass->frame_counter;
	hipEvent_t caller_event = (hipEvent_t) pass->wait_before_next_frame;
	pass->wait_before_next_frame = NULL;
	int status = 0;
	frame_context* frame = NULL;
	// bytes of the output this call writes (a slab in slab layout, else the frame): what pending read-backs are checked against
	const size_t frame_output_bytes = sizeof(float4) * (p.slab_layout ? (size_t) grid_blocks * 256u : (size_t) p.width * p.height);
	for (uint32_t band = 0; band != band_count && status == 0; ++band) {
		p.first_block = band * blocks_per_band;
		p.block_count = grid_blocks - p.first_block < blocks_per_band ? grid_blocks - p.first_block : blocks_per_band;
		frame = NULL;
		if (frames && is_deferred(ray_mode)) {
			uint32_t index = 0;
			if (pipelined) {
				index = frames->next;
				frames->next = (index + 1) % depth;
				stream = (hipStream_t) device->frame_streams[index];
			}
			frame = &frames->contexts[index];
			frames->last = index;
			// (with a tracing stream the previous launch of this context did not end on this stream)
			if (pipelined && frame->trace_stream && frame->recorded) (void) hipStreamWaitEvent(stream, frame->done, 0);
			if (ensure_wavefront(&frame->buffers, blocks_per_band * 256u, max_terms, p.light_count, hidden_terms, base_color, stream)) return 1;
			if (use_wide_tree && ensure_spill(&frame->buffers, app->scene.acceleration_structure.wide_stack_need, frames->wide_stack_lds, trace_blocks * 256u)) return 1;
			const wavefront_buffers* w = &frame->buffers;
			p.codes = w->codes; p.terms_visible = w->terms_visible; p.terms_hidden = w->terms_hidden; p.base_color = w->base_color;
			p.ray_directions = w->ray_directions; p.ray_records = w->ray_records; p.ray_origins = w->ray_origins; p.ray_queue_size = w->ray_queue_size;
			p.thread_count = w->thread_count; p.max_terms = w->max_terms; p.max_codes = w->max_codes;
			p.ray_queue_capacity = w->queue_capacity; p.ray_thread_bits = w->thread_bits;
		}
		if (table_in_memory) {
			// a region per workgroup of the launch, in the launch's own buffers (VKR_PSA_TABLE_INDEX=slot: one buffer for the
			// whole pass, indexed by the hardware slot a wave runs in - shading_kernel.h hardware_wave_slot.  Measured, config 4:
			// the same time and the same traffic, 16.40 / 16.42 ms and 8.8 / 9.0 GB per frame, profiles/r10o - the table stores
			// reach the fabric either way -, so the scheme that assumes nothing about the hardware is the default)
			const bool by_block = !(getenv("VKR_PSA_TABLE_INDEX") != NULL && strcmp(getenv("VKR_PSA_TABLE_INDEX"), "slot") == 0);
			wavefront_buffers* owner = (frame && by_block) ? &frame->buffers : &frames->device_stream_buffers;
			if (ensure_psa_table_memory(owner, (size_t) (by_block ? shade_grid_size(blocks_per_band) : kWaveSlots) * table_bytes_per_workgroup)) return 1;
			p.psa_table_memory = owner->psa_table_memory;
			p.psa_table_by_wave_slot = by_block ? 0u : 1u;
		}
		pass->last_frame_stream = stream;
		// (a target that earlier work of the caller still reads: every stream that writes it waits)
		if (caller_event && hip_failed(hipStreamWaitEvent(stream, caller_event, 0), "waiting for the caller's event")) return 1;
		// (wavefront rays: the resolve kernel of the frame's first launch stores the count instead)
		p.first_launch_of_frame = band == 0 ? 1u : 0u;
		if (band == 0 && p.ray_counter && !is_deferred(ray_mode) && hip_failed(hipMemsetAsync(p.ray_counter, 0, sizeof(unsigned long long), stream), "clearing the ray counter")) return 1;
		if (upload_constants(app, stream)) return 1;
		p.constants = (const uint8_t*) pass->constants_device;
		// (the first event of a timed frame: everything the frame launches lies behind it)
		if (timed && band == 0) (void) hipEventRecord(ring[kTimingEvents * slot], stream);
		if (textured && band == 0) {
			if (g_resolve_launchers[pass->arithmetic_mode](&p, (float*) pass->pixel_materials, stream)) {
				printf("Launching the material resolve kernel failed.\n");
				return 1;
			}
			p.pixel_materials = (const float*) pass->pixel_materials;
		}
		// light shafts: which patches need no shadow rays toward which lights (light_shafts.h).  For the techniques
		// whose samples aim at the light polygon itself (every ray then lies inside the shaft); the walk uses the
		// four-wide tree whatever tree the rays walk.
		p.shaft_clear = NULL;
		p.shaft_rectangles = NULL;
		p.shaft_lists = NULL;
		const uint32_t rays_per_pixel = app->render_settings.sample_count * p.light_count * ((int) app->render_settings.sampling_strategies == (int) sampling_strategies_diffuse_only ? 1u : 2u);
		if (frame && (frames->light_shafts == 1u || (frames->light_shafts == 2u && rays_per_pixel >= 8u)) && is_deferred(ray_mode) && error_mode == kErrorNone && app->scene.acceleration_structure.wide_nodes && p.light_count
			&& app->scene.acceleration_structure.node_count < (1u << kShaftLightShift)  // (a queue entry of the walk is a node or triangle index and a light)
			&& (technique == kTechniquePsa || technique == kTechniquePsaBiased || technique == kTechniqueSolidAngle || technique == kTechniqueClippedSolidAngle))
		{
			const acceleration_structure_t* structure = &app->scene.acceleration_structure;
			uint32_t shaft_groups = shade_grid_size(p.block_count);
			const bool lists = frames->shaft_lists != 0u && kShaftListMax != 0u;
			if (ensure_shaft_words(&frame->buffers, (size_t) shaft_groups * p.light_count + 8, p.light_count, lists, stream)) return 1;
			float extent = 0.0f;
			for (int j = 0; j != 3; ++j) extent = fmaxf(extent, kGridMax / structure->grid_inverse_cell[j]);
			// (VKR_SHAFT_COUNTERS=1: the walks count their steps into three words behind the table, for get_light_shaft_work())
			static const bool count_work = getenv("VKR_SHAFT_COUNTERS") != NULL;
			unsigned long long* work = NULL;
			if (count_work) {
				work = (unsigned long long*) (frame->buffers.shaft_clear + (((size_t) shaft_groups * p.light_count + 1u) & ~(size_t) 1u));
				(void) hipMemsetAsync(work, 0, 3 * sizeof(unsigned long long), stream);
			}
			// The tag of the launch: its geometry AND what its verdicts were derived from - the tree (address, sizes,
			// build time: a scene loaded into the same allocation differs in one of them), the contents of the visibility buffer
			// (a generation counter, bumped by render_visibility_pass / upload_visibility / mark_inputs_changed), the camera and the
			// bytes of the light array in the constants.  When any of them changes no pair rests on a verdict of the old
			// arrangement: ray counts and shaft statistics are then a function of the frame and of how long it has stood still,
			// not of what was rendered before.
			uint64_t tag = 0xcbf29ce484222325ull;
			uint32_t build_bits;
			memcpy(&build_bits, &structure->build_milliseconds, sizeof(build_bits));
			for (uint64_t word : {(uint64_t) p.first_block, (uint64_t) p.block_count, (uint64_t) p.width, (uint64_t) p.height, (uint64_t) p.tile_size, (uint64_t) p.rank, (uint64_t) p.rank_count,
					(uint64_t) p.slab_layout, (uint64_t) p.light_count, (uint64_t) (uintptr_t) p.visibility, (uint64_t) lists,
					(uint64_t) (uintptr_t) structure->wide_nodes, (uint64_t) structure->wide_node_count, (uint64_t) structure->node_count, (uint64_t) build_bits, (uint64_t) frames->inputs_generation})
				tag = (tag ^ word) * 0x100000001b3ull;
			{
				const uint8_t* bytes = (const uint8_t*) pass->constants_host;
				const size_t ranges[3][2] = {{offsetof(per_frame_constants_t, world_to_projection_space), offsetof(per_frame_constants_t, mis_visibility_estimate)},
					{offsetof(per_frame_constants_t, mesh_dequantization_factor), offsetof(per_frame_constants_t, error_factor)},
					{sizeof(per_frame_constants_t), pass->constants_size}};
				for (const size_t* range : ranges)
					for (size_t i = range[0]; i + 4 <= range[1]; i += 4) {
						uint32_t word;
						memcpy(&word, bytes + i, sizeof(word));
						tag = (tag ^ word) * 0x100000001b3ull;
					}
			}
			const bool same_launch = frame->buffers.shaft_tag == tag;
			frame->buffers.shaft_tag = tag;
			k_light_shafts<<<shaft_groups, 64, 0, stream>>>(p, (const uint4*) structure->wide_nodes, frame->buffers.shaft_clear, frame->buffers.shaft_rectangles, lists ? frame->buffers.shaft_lists : NULL, extent, work, same_launch ? frames->shaft_rest : 0u,
				// (a small launch lasts as long as its longest walk: above, at trace_blocks)
				frames->shaft_max_steps ? frames->shaft_max_steps : (shaft_groups < 12288u ? kShaftSmallLaunchSteps : kShaftMaxSteps));
			if (hip_failed(hipGetLastError(), "launching the light shaft kernel")) return 1;
			p.shaft_clear = frame->buffers.shaft_clear;
			p.shaft_rectangles = frame->buffers.shaft_rectangles;
			p.shaft_lists = lists ? frame->buffers.shaft_lists : NULL;
			pass->last_shaft_groups = shaft_groups;
		}
		else pass->last_shaft_groups = 0;
		// (a frame without wavefront rays writes its output from the shading kernel)
		if (!is_deferred(ray_mode)) wait_for_read_backs_of(pass, p.out_radiance, frame_output_bytes, stream);
		// (the second event of a timed frame: the shading kernel itself begins here, behind the shaft kernel)
		if (timed && band == 0) (void) hipEventRecord(ring[kTimingEvents * slot + 1], stream);
		status = error_mode != kErrorNone
			? g_error_launchers[pass->arithmetic_mode](strategy >= (int) sampling_strategies_diffuse_specular_separately, technique, capacity, error_mode, &p, p.block_count, stream)
			: g_launchers[pass->arithmetic_mode + (p.light_texture_descriptors ? 3 : 0)][strategy](technique, capacity, ray_mode, &p, p.block_count, stream);
		if (timed && band + 1 == band_count) (void) hipEventRecord(ring[kTimingEvents * slot + 2], stream);
		if (status == 0 && is_deferred(ray_mode)) {
			if (pipelined && frame->trace_stream) {
				// the rest of the launch moves to the high-priority stream; the next launch that reuses this context's
				// frame stream is ordered behind `done` below (which is then recorded on the tracing stream)
				(void) hipEventRecord(frame->shaded, stream);
				(void) hipStreamWaitEvent(frame->trace_stream, frame->shaded, 0);
				stream = frame->trace_stream;
				pass->last_frame_stream = stream;
			}
			ray_stream rays = {p.ray_directions, p.ray_records, p.ray_origins, p.ray_queue_size, p.ray_queue_capacity, p.ray_thread_bits, p.thread_count};
			if (use_wide_tree) {
				const uint4* wide_nodes = (const uint4*) app->scene.acceleration_structure.wide_nodes;
				// single-wave workgroups where a lane queues many rays and the shading kernel runs three waves
				// per SIMD (wavefront_kernels.h has the measurements)
				bool single_waves = ray_mode == kRaysDeferredBlocks && capacity <= 7;
				if (frames->trace_single_waves != 2u) single_waves = frames->trace_single_waves != 0u;
				if (single_waves) {
					trace_shadow_rays_wide<64><<<trace_blocks * 4u, 64, 0, stream>>>(p.bvh, wide_nodes, app->scene.acceleration_structure.wide_node_count, rays, p.ray_queue_size + kRayQueueCount, p.codes, frame->buffers.spill, frames->leaf_batch, frames->wide_stack_lds, frames->wide_refill, frames->wide_refill_below);
				}
				else {
					trace_shadow_rays_wide<256><<<trace_blocks, 256, 0, stream>>>(p.bvh, wide_nodes, app->scene.acceleration_structure.wide_node_count, rays, p.ray_queue_size + kRayQueueCount, p.codes, frame->buffers.spill, frames->leaf_batch, frames->wide_stack_lds, frames->wide_refill, frames->wide_refill_below);
				}
			}
			else
				trace_shadow_rays<<<trace_blocks, 256, 0, stream>>>(p.bvh, rays, p.ray_queue_size + kRayQueueCount, p.codes, p.refill_threshold);
			if (pipelined) {
				// launches in flight may write the same target: keep their order there
				// (the launch before this one ran in the context before this one)
				// (whether device->stream has already been made to wait for that launch - finish_frames() -
				// says nothing about this stream)
				frame_context* previous = &frames->contexts[(frames->last + frames->depth - 1u) % frames->depth];
				// (bands of one frame, and frames that share a target; VKR_ORDER_ALL_RESOLVES=1: always, as until round 5)
				static const bool order_all = getenv("VKR_ORDER_ALL_RESOLVES") != NULL;
				// (a caller's ring of targets need not have the length of the pipeline: every other context one of whose recent
				// launches wrote this target comes first)
				for (uint32_t c = 0; c != frames->depth; ++c) {
					frame_context* other = &frames->contexts[c];
					if (other == frame || !other->recorded) continue;
					bool same = order_all || (other == previous && band != 0);
					for (const void* written : other->targets) same = same || written == (const void*) p.out_radiance;
					if (same) (void) hipStreamWaitEvent(stream, other->done, 0);
				}
				frame->targets[frame->target_cursor++ % 8u] = p.out_radiance;
				// ... and behind whatever still reads the target on device->stream (output encoding)
				if (frame->readers_seen != frames->readers_generation) {
					(void) hipStreamWaitEvent(stream, frames->readers_done, 0);
					frame->readers_seen = frames->readers_generation;
				}
			}
			wait_for_read_backs_of(pass, p.out_radiance, frame_output_bytes, stream);
			resolve_shadow_terms_and_reset<<<p.block_count, 256, 0, stream>>>(p);
			status = hipGetLastError() != hipSuccess;
		}
		if (status == 0 && out_rgb8 && band + 1 == band_count) {
			// (the resolves of the bands are chained, so the last band's stream has seen them all)
			// slab layout: every thread of the grid owns a slot; full-frame layout: the pixels
			uint64_t pixels = p.slab_layout ? (uint64_t) grid_blocks * 256u : (uint64_t) p.width * p.height;
			if (pixels % 4 != 0) {
				printf("The frame cannot be encoded as packed RGB8 (its pixel count has to be a multiple of four).\n");
				status = 1;
			}
			else {
				wait_for_read_backs_of(pass, out_rgb8, (size_t) pixels * 3u, stream);
				k_encode_output_rgb8<<<(uint32_t) ((pixels / 4 + 255) / 256), 256, 0, stream>>>((const float4*) p.out_radiance, (uint32_t*) out_rgb8, pixels / 4, app->screenshot.frame_bits, 0);
				status = hipGetLastError() != hipSuccess;
			}
		}
		if (status == 0 && pipelined) {
			(void) hipEventRecord(frame->done, stream);
			frame->pending = frame->recorded = true;
		}
	}
	if (timed) {
		(void) hipEventRecord(ring[kTimingEvents * slot + 3], stream);
		++pass->timing_cursor;
	}
	if (status < 0) {
		printf("No kernel variant was built for strategy %d, technique %d, vertex capacity %d.\n", strategy, technique, capacity);
		return 1;
	}
	if (status > 0) {
		// the resolve kernel did not run, so the ray queues may not be empty
		if (frame && frame->buffers.ray_queue_size) (void) hipMemsetAsync(frame->buffers.ray_queue_size, 0, sizeof(uint32_t) * kRayCounterCount, stream);
		printf("Launching the shading kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
		return 1;
	}
	return 0;
}

extern "C" uint32_t get_dispatch_milliseconds(application_t* app, float* out, uint32_t count) {
	shading_pass_t* pass = &app->shading_pass;
	if (!pass->timing_ring) return 0;
	hipEvent_t* ring = (hipEvent_t*) pass->timing_ring;
	uint32_t available = pass->timing_cursor < pass->timing_ring_size ? pass->timing_cursor : pass->timing_ring_size;
	if (count > available) count = available;
	for (uint32_t i = 0; i != count; ++i) {
		uint32_t slot = (pass->timing_cursor - count + i) % pass->timing_ring_size;
		float ms = 0.0f;
		if (hipEventSynchronize(ring[kTimingEvents * slot + 3]) != hipSuccess || hipEventElapsedTime(&ms, ring[kTimingEvents * slot], ring[kTimingEvents * slot + 3]) != hipSuccess) ms = 0.0f;
		out[i] = ms;
	}
	return count;
}

// what a timed frame spends before its shading kernel starts: the shaft kernel (and, for textured scenes, the material resolve)
extern "C" uint32_t get_light_shaft_milliseconds(application_t* app, float* out, uint32_t count) {
	shading_pass_t* pass = &app->shading_pass;
	if (!pass->timing_ring) return 0;
	hipEvent_t* ring = (hipEvent_t*) pass->timing_ring;
	uint32_t available = pass->timing_cursor < pass->timing_ring_size ? pass->timing_cursor : pass->timing_ring_size;
	if (count > available) count = available;
	for (uint32_t i = 0; i != count; ++i) {
		uint32_t slot = (pass->timing_cursor - count + i) % pass->timing_ring_size;
		float ms = 0.0f;
		if (hipEventSynchronize(ring[kTimingEvents * slot + 1]) != hipSuccess || hipEventElapsedTime(&ms, ring[kTimingEvents * slot], ring[kTimingEvents * slot + 1]) != hipSuccess) ms = 0.0f;
		out[i] = ms;
	}
	return count;
}

extern "C" uint32_t get_shading_kernel_milliseconds(application_t* app, float* out, uint32_t count) {
	shading_pass_t* pass = &app->shading_pass;
	if (!pass->timing_ring) return 0;
	hipEvent_t* ring = (hipEvent_t*) pass->timing_ring;
	uint32_t available = pass->timing_cursor < pass->timing_ring_size ? pass->timing_cursor : pass->timing_ring_size;
	if (count > available) count = available;
	for (uint32_t i = 0; i != count; ++i) {
		uint32_t slot = (pass->timing_cursor - count + i) % pass->timing_ring_size;
		float ms = 0.0f;
		if (hipEventSynchronize(ring[kTimingEvents * slot + 2]) != hipSuccess || hipEventElapsedTime(&ms, ring[kTimingEvents * slot + 1], ring[kTimingEvents * slot + 2]) != hipSuccess) ms = 0.0f;
		out[i] = ms;
	}
	return count;
}

extern "C" uint32_t get_frame_period_milliseconds(application_t* app, float* out, uint32_t count) {
	shading_pass_t* pass = &app->shading_pass;
	if (!pass->timing_ring || pass->timing_cursor < 2) return 0;
	hipEvent_t* ring = (hipEvent_t*) pass->timing_ring;
	uint32_t stride = pass->timing_stride > 1 ? pass->timing_stride : 1;
	uint32_t available = (pass->timing_cursor < pass->timing_ring_size ? pass->timing_cursor : pass->timing_ring_size) - 1;
	if (count > available) count = available;
	for (uint32_t i = 0; i != count; ++i) {
		uint32_t later = (pass->timing_cursor - count + i) % pass->timing_ring_size;
		uint32_t earlier = (later + pass->timing_ring_size - 1) % pass->timing_ring_size;
		float ms = 0.0f;
		if (hipEventSynchronize(ring[kTimingEvents * later + 3]) != hipSuccess || hipEventElapsedTime(&ms, ring[kTimingEvents * earlier + 3], ring[kTimingEvents * later + 3]) != hipSuccess) ms = 0.0f;
		out[i] = ms / (float) stride;
	}
	return count;
}

extern "C" float get_last_dispatch_milliseconds(application_t* app) {
	float ms = 0.0f;
	if (get_dispatch_milliseconds(app, &ms, 1) != 1) return 0.0f;
	app->shading_pass.last_dispatch_ms = ms;
	return ms;
}

// Diagnostics: replays the rays that the last frame queued and counts the work of the
// traversal (profiles/ cites these numbers; not part of the frame).
__global__ void __launch_bounds__(256) k_traversal_statistics(bvh_view bvh, ray_stream stream, unsigned long long* out) {
	uint32_t queue = blockIdx.y;
	uint32_t size = stream.sizes[queue];
	unsigned long long visits = 0, tests = 0, blocked_rays = 0, rays = 0, wave_steps = 0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < ((size + 63u) & ~63u); i += gridDim.x * 256u) {
		uint32_t my_visits = 0;
		size_t slot = (size_t) queue * stream.capacity + i;
		if (i < size && stream.records[slot] != kNullRay) {
			float4 a = stream.origins[ray_record_thread(stream.thread_bits, stream.records[slot])], b = stream.directions[slot];
			f3 o = mk3(a.x, a.y, a.z), d = mk3(b.x, b.y, b.z);
			float t_max = b.w;
			grid_ray ray = make_grid_ray(bvh, o, d);
			uint32_t node = 0;
			bool blocked = false;
			++rays;
			while (t_max >= 1.0e-3f && node < bvh.node_count && !blocked) {
				uint4 n = bvh.nodes[node];
				bool is_leaf = (n.w & kLeafBit) != 0;
				bool hit = ray_box(n, ray, 1.0e-3f, t_max);
				++my_visits;
				if (hit && is_leaf) {
					const float4* t = bvh.triangles + 3 * (size_t) (n.w & ~kLeafBit);
					float dist;
					++tests;
					blocked = ray_triangle<false>(t[0], t[1], t[2], o, d, 1.0e-3f, t_max, dist);
				}
				node = (hit || is_leaf) ? node + 1 : n.w;
			}
			blocked_rays += blocked ? 1 : 0;
		}
		visits += my_visits;
		// steps the wave needs for these 64 rays = longest ray
		uint32_t longest = my_visits;
		for (int offset = 32; offset > 0; offset >>= 1) longest = max(longest, (uint32_t) __shfl_xor((int) longest, offset));
		if ((threadIdx.x & 63u) == 0) wave_steps += longest;
		atomicMax(out + 5, (unsigned long long) my_visits);
	}
	atomicAdd(out + 0, rays); atomicAdd(out + 1, visits); atomicAdd(out + 2, tests);
	atomicAdd(out + 3, blocked_rays); atomicAdd(out + 4, wave_steps);
}

// The same for the four-wide tree: "visits" are fetched nodes (dependent loads), out[5] the longest
// ray's, wave steps the longest ray of each group of 64; out[6] counts tested boxes, out[7] the
// deepest stack a ray reached, out[8] the rays whose stack outgrows the `lds_entries` entries that
// trace_shadow_rays_wide keeps in LDS (it holds the next item on the stack too: one entry more than the
// scheme here), out[9] the node visits of the rays that end up blocked
__global__ void __launch_bounds__(256) k_traversal_statistics_wide(bvh_view bvh, const uint4* wide_nodes, ray_stream stream, uint32_t lds_entries, unsigned long long* out) {
	uint32_t queue = blockIdx.y;
	uint32_t size = stream.sizes[queue];
	unsigned long long visits = 0, tests = 0, blocked_rays = 0, rays = 0, wave_steps = 0, boxes = 0, beyond_lds = 0, blocked_visits = 0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < ((size + 63u) & ~63u); i += gridDim.x * 256u) {
		uint32_t my_visits = 0;
		size_t slot = (size_t) queue * stream.capacity + i;
		if (i < size && stream.records[slot] != kNullRay) {
			float4 a = stream.origins[ray_record_thread(stream.thread_bits, stream.records[slot])], b = stream.directions[slot];
			f3 o = mk3(a.x, a.y, a.z), d = mk3(b.x, b.y, b.z);
			float t_max = b.w;
			wide_ray ray = make_wide_ray(make_grid_ray(bvh, o, d));
			uint32_t stack[kWideStackMax];
			uint32_t depth = 0, deepest = 0, item = 0;
			bool blocked = false, done = !(t_max >= 1.0e-3f);
			++rays;
			while (!done) {
				if (item & kLeafBit) {
					const float4* t = bvh.triangles + 3 * (size_t) (item & ~kLeafBit);
					float dist;
					++tests;
					blocked = ray_triangle<false>(t[0], t[1], t[2], o, d, 1.0e-3f, t_max, dist);
					if (blocked) break;
					item = 0xFFFFFFFFu;
				}
				else {
					const uint4* n = wide_nodes + 4 * (size_t) item;
					uint4 qx = n[0], qy = n[1], qz = n[2], link = n[3];
					const uint32_t x[4] = {qx.x, qx.y, qx.z, qx.w}, y[4] = {qy.x, qy.y, qy.z, qy.w}, z[4] = {qz.x, qz.y, qz.z, qz.w}, links[4] = {link.x, link.y, link.z, link.w};
					++my_visits;
					item = 0xFFFFFFFFu;
					// the order of trace_shadow_rays_wide: child 0 next, then 1, 2, 3 (pushed last to first)
					for (int c = 3; c >= 0; --c) {
						if (links[c] == kWideEmpty) continue;
						++boxes;
						if (!wide_ray_box(x[c], y[c], z[c], ray, 1.0e-3f, t_max)) continue;
						if (item != 0xFFFFFFFFu && depth < kWideStackMax) stack[depth++] = item;
						item = links[c];
					}
					deepest = max(deepest, depth);
				}
				if (item == 0xFFFFFFFFu) {
					if (depth == 0) done = true;
					else item = stack[--depth];
				}
			}
			blocked_rays += blocked ? 1 : 0;
			blocked_visits += blocked ? my_visits : 0u;
			beyond_lds += (deepest + 1u > lds_entries) ? 1 : 0;
			atomicMax(out + 7, (unsigned long long) deepest);
		}
		visits += my_visits;
		uint32_t longest = my_visits;
		for (int offset = 32; offset > 0; offset >>= 1) longest = max(longest, (uint32_t) __shfl_xor((int) longest, offset));
		if ((threadIdx.x & 63u) == 0) wave_steps += longest;
		atomicMax(out + 5, (unsigned long long) my_visits);
	}
	atomicAdd(out + 0, rays); atomicAdd(out + 1, visits); atomicAdd(out + 2, tests);
	atomicAdd(out + 3, blocked_rays); atomicAdd(out + 4, wave_steps); atomicAdd(out + 6, boxes);
	if (beyond_lds) atomicAdd(out + 8, beyond_lds);
	if (blocked_visits) atomicAdd(out + 9, blocked_visits);
}

// evaluate_device_arithmetic(): the primitives as the shading kernels use them (this translation unit
// is compiled in exact mode, -ffp-contract=off)
__global__ void __launch_bounds__(256) k_evaluate_arithmetic(uint32_t operation, const float* a, const float* b, float* out, uint32_t count) {
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= count) return;
	float x = a[i], y = b ? b[i] : 0.0f;
	switch (operation) {
	case 0: out[i] = divide(x, y); break;
	case 1: out[i] = square_root(x); break;
	case 2: out[i] = rsqrt(x); break;
	case 3: out[i] = x / y; break;
	case 4: out[i] = sqrtf(x); break;
	// the functions of the libm arithmetic mode (glibc_math.h with this file's divide / square_root)
	case 5: out[i] = gm_atanf(x); break;
	case 6: out[i] = gm_acosf(x); break;
	case 7: out[i] = gm_sinf(x); break;
	case 8: out[i] = gm_cosf(x); break;
	case 9: out[i] = gm_log2f(x); break;
	case 10: out[i] = gm_powf(x, y); break;
	case 11: out[i] = gm_atan2f(x, y); break;
	case 12: out[i] = inverse_square_root_ieee(x); break;
	default: out[i] = rsqrt(x); break;  // (what the kernels of this unit's arithmetic mode use)
	}
}

// compare_device_arithmetic(): two one-argument operations of k_evaluate_arithmetic over a range of bit
// patterns, without moving the arguments through the host
__device__ __forceinline__ float evaluate_unary(uint32_t operation, float x, const gm_atan_row_t* atan_rows) {
	switch (operation) {
	case 17: return gm_atanf_rows(x, atan_rows);
	case 1: return square_root(x);
	case 4: return sqrtf(x);
	case 5: return gm_atanf(x);
	case 12: return inverse_square_root_ieee(x);
	case 16: return 1.0f / sqrtf(x);
	default: return rsqrt(x);
	}
}
__global__ void __launch_bounds__(256) k_compare_arithmetic(uint32_t operation_a, uint32_t operation_b, uint32_t first_bits, uint64_t count, unsigned long long* out) {
	// (the table of the arctangent's argument ranges, in LDS as in the shading kernels)
	__shared__ gm_atan_row_t atan_rows[GM_ATAN_ROW_COUNT];
	for (uint32_t i = threadIdx.x; i < GM_ATAN_ROW_COUNT; i += 256u) atan_rows[i] = gm_atan_row(i);
	__syncthreads();
	unsigned long long mismatches = 0;
	for (uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t) gridDim.x * 256u) {
		uint32_t bits = first_bits + (uint32_t) i;
		float x = __uint_as_float(bits), a = evaluate_unary(operation_a, x, atan_rows), b = evaluate_unary(operation_b, x, atan_rows);
		bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
		if (!same) { ++mismatches; atomicMin(out + 1, (unsigned long long) bits); }
	}
	if (mismatches) atomicAdd(out, mismatches);
}

extern "C" int compare_device_arithmetic(const device_t* device, uint32_t operation_a, uint32_t operation_b, uint32_t first_bits, uint64_t count, uint64_t out_mismatches_and_first[2]) {
	if (!device || !out_mismatches_and_first || count > (1ull << 32)) {
		printf("compare_device_arithmetic() needs a device, an output and at most 2^32 arguments.\n");
		return 1;
	}
	unsigned long long* counters = NULL;
	if (hip_failed(hipMalloc(&counters, 2 * sizeof(unsigned long long)), "allocating counters")) return 1;
	hipStream_t stream = (hipStream_t) device->stream;
	unsigned long long initial[2] = {0ull, ~0ull};
	int failed = hip_failed(hipMemcpyAsync(counters, initial, sizeof(initial), hipMemcpyHostToDevice, stream), "clearing counters");
	if (!failed) {
		k_compare_arithmetic<<<8192, 256, 0, stream>>>(operation_a, operation_b, first_bits, count, counters);
		failed = vkr_copy_to_host(out_mismatches_and_first, counters, 2 * sizeof(unsigned long long), device);
	}
	(void) hipFree(counters);
	return failed;
}

// compare_device_division(): divide() against the compiler's IEEE a / b for a block of divisor
// significands and EVERY dividend significand (blockIdx.y = divisor, the threads of its blocks share the dividends)
__global__ void __launch_bounds__(256) k_compare_division(uint32_t first_significand, uint32_t stride, uint32_t dividend_exponent, uint32_t divisor_exponent, unsigned long long* out) {
	const uint32_t b_bits = (divisor_exponent << 23) | ((first_significand + blockIdx.y * stride) & 0x7FFFFFu);
	const float b = __uint_as_float(b_bits);
	unsigned long long mismatches = 0;
	for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < (1u << 23); m += gridDim.x * 256u) {
		const uint32_t a_bits = (dividend_exponent << 23) | m;
		const float a = __uint_as_float(a_bits);
		float mine = divide(a, b), theirs = __fdiv_rn(a, b);
		bool same = __float_as_uint(mine) == __float_as_uint(theirs) || (mine != mine && theirs != theirs);
		if (!same) { ++mismatches; atomicMin(out + 1, ((unsigned long long) b_bits << 32) | a_bits); }
	}
	if (mismatches) atomicAdd(out, mismatches);
}

extern "C" int compare_device_division(const device_t* device, uint32_t first_significand, uint32_t divisor_count, uint32_t stride, uint32_t dividend_exponent, uint32_t divisor_exponent, uint64_t out_mismatches_and_first[2]) {
	if (!device || !out_mismatches_and_first || divisor_count == 0 || divisor_count > 65535u || dividend_exponent > 254u || divisor_exponent > 254u) {
		printf("compare_device_division() needs a device, an output, 1 ... 65535 divisors and biased exponents below 255.\n");
		return 1;
	}
	unsigned long long* counters = NULL;
	if (hip_failed(hipMalloc(&counters, 2 * sizeof(unsigned long long)), "allocating counters")) return 1;
	hipStream_t stream = (hipStream_t) device->stream;
	unsigned long long initial[2] = {0ull, ~0ull};
	int failed = hip_failed(hipMemcpyAsync(counters, initial, sizeof(initial), hipMemcpyHostToDevice, stream), "clearing counters");
	if (!failed) {
		k_compare_division<<<dim3(32, divisor_count), 256, 0, stream>>>(first_significand, stride, dividend_exponent, divisor_exponent, counters);
		failed = vkr_copy_to_host(out_mismatches_and_first, counters, 2 * sizeof(unsigned long long), device);
	}
	(void) hipFree(counters);
	return failed;
}

// Every wave claims the word of its hardware slot, stays for a while and leaves it again; a wave that finds the word taken
// shares its slot with a wave that is still there.  out[0]: such waves, out[1]: slots that were used
__global__ void __launch_bounds__(64) k_check_hardware_wave_slots(uint32_t* owners, unsigned long long* out, uint32_t spin) {
	uint32_t slot = hardware_wave_slot();
	uint32_t old = 0;
	if (threadIdx.x == 0) old = atomicExch(owners + slot, blockIdx.x + 1u);
	old = __builtin_amdgcn_readfirstlane(old);
	if (threadIdx.x == 0 && old != 0u) atomicAdd(out, 1ull);
	if (threadIdx.x == 0 && old == 0u && atomicOr(owners + kWaveSlots + slot, 1u) == 0u) atomicAdd(out + 1, 1ull);
	// (work that the compiler cannot remove and whose length differs between waves)
	float x = (float) threadIdx.x;
	for (uint32_t i = 0; i != spin * (1u + (blockIdx.x & 7u)); ++i) x = fmaf(x, 1.0000001f, 1.0e-7f);
	if (x == 12345.678f) out[2] = 1ull;
	// (with the value it returns: the wave waits for the exchange before it ends.  Without, the wave may be gone - and the next
	// one in its slot - while the exchange is still on its way, and the next wave finds the slot "taken": 149 of 400 000 waves in
	// the first version of this check.  The same holds for any store: that a wave's last stores have landed when its successor in
	// the slot starts is nothing the hardware promises, which is why regions by wave slot are an option of the pass, not its default)
	if (threadIdx.x == 0 && atomicExch(owners + slot, 0u) == 0xFFFFFFFFu) out[2] = 2ull;
}

extern "C" int check_hardware_wave_slots(const device_t* device, uint32_t workgroups, uint32_t extra_lds_bytes, uint64_t out_shared_and_used[2]) {
	uint32_t* owners = NULL;
	unsigned long long* out = NULL;
	hipStream_t stream = (hipStream_t) device->stream;
	int failed = hip_failed(hipMalloc(&owners, sizeof(uint32_t) * 2 * kWaveSlots), "allocating the slot owners") || hip_failed(hipMalloc(&out, 3 * sizeof(unsigned long long)), "allocating counters")
		|| hip_failed(hipMemsetAsync(owners, 0, sizeof(uint32_t) * 2 * kWaveSlots, stream), "clearing") || hip_failed(hipMemsetAsync(out, 0, 3 * sizeof(unsigned long long), stream), "clearing");
	if (!failed) {
		// (two launches on two streams at once: slots are unique across kernels, not only within one)
		hipStream_t other = (hipStream_t) device->frame_streams[0];
		if (other) {
			hipEvent_t ready;
			failed = hip_failed(hipEventCreateWithFlags(&ready, hipEventDisableTiming), "creating an event") || hip_failed(hipEventRecord(ready, stream), "recording") || hip_failed(hipStreamWaitEvent(other, ready, 0), "waiting");
			if (!failed) k_check_hardware_wave_slots<<<workgroups, 64, extra_lds_bytes, other>>>(owners, out, 2000u);
			(void) hipEventDestroy(ready);
		}
		k_check_hardware_wave_slots<<<workgroups, 64, extra_lds_bytes, stream>>>(owners, out, 3000u);
		failed = failed || hip_failed(hipGetLastError(), "launching the slot check");
		if (other) failed = failed || hip_failed(hipStreamSynchronize(other), "waiting for the slot check");
	}
	unsigned long long host[3] = {0, 0, 0};
	failed = failed || hip_failed(hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, stream), "reading the counters") || hip_failed(hipStreamSynchronize(stream), "waiting for the slot check");
	out_shared_and_used[0] = host[0]; out_shared_and_used[1] = host[1];
	(void) hipFree(owners); (void) hipFree(out);
	return failed;
}

__global__ void __launch_bounds__(256) k_copy_with_workgroups(uint4* destination, const uint4* source, uint64_t count) {
	for (uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t) gridDim.x * 256u) destination[i] = source[i];
}

extern "C" int copy_with_workgroups(void* destination, const void* source, uint64_t bytes, uint32_t workgroups, void* stream) {
	if (!destination || !source || bytes % 16 != 0 || workgroups == 0) {
		printf("copy_with_workgroups() needs two device buffers, a multiple of 16 bytes and at least one workgroup.\n");
		return 1;
	}
	if (bytes == 0) return 0;
	k_copy_with_workgroups<<<workgroups, 256, 0, (hipStream_t) stream>>>((uint4*) destination, (const uint4*) source, bytes / 16);
	return hip_failed(hipGetLastError(), "launching the copy");
}

extern "C" int evaluate_device_arithmetic(const device_t* device, uint32_t operation, const float* a, const float* b, float* out, uint32_t count) {
	if (!device || !a || !out || operation > 12 || ((operation == 0 || operation == 3 || operation == 10 || operation == 11) && !b)) {
		printf("evaluate_device_arithmetic() needs a device, operands and an operation in 0 ... 12.\n");
		return 1;
	}
	if (!count) return 0;
	hipStream_t stream = (hipStream_t) device->stream;
	float* buffers = NULL;
	size_t bytes = sizeof(float) * (size_t) count;
	if (hip_failed(hipMalloc(&buffers, 3 * bytes), "allocating the operands")) return 1;
	int failed = hip_failed(hipMemcpyAsync(buffers, a, bytes, hipMemcpyHostToDevice, stream), "uploading the operands")
		|| (b && hip_failed(hipMemcpyAsync(buffers + count, b, bytes, hipMemcpyHostToDevice, stream), "uploading the operands"));
	if (!failed) {
		k_evaluate_arithmetic<<<(count + 255u) / 256u, 256, 0, stream>>>(operation, buffers, b ? buffers + count : NULL, buffers + 2 * (size_t) count, count);
		failed = hip_failed(hipMemcpyAsync(out, buffers + 2 * (size_t) count, bytes, hipMemcpyDeviceToHost, stream), "reading the results back")
			|| hip_failed(hipStreamSynchronize(stream), "evaluating the arithmetic");
	}
	(void) hipFree(buffers);
	return failed;
}

extern "C" int get_traversal_statistics(application_t* app, uint64_t out_statistics[6]) {
	const frame_pipeline* frames = (const frame_pipeline*) app->shading_pass.wavefront;
	const wavefront_buffers* w = frames ? &frames->contexts[frames->last].buffers : NULL;
	if (!w || !w->ray_directions || !app->shading_pass.use_ray_tracing || app->shading_pass.inline_rays) {
		printf("get_traversal_statistics() needs a frame rendered with wavefront shadow rays.\n");
		return 1;
	}
	uint64_t all[12];
	int failed = get_traversal_statistics_of_tree(app, app->scene.acceleration_structure.wide_nodes && !app->shading_pass.binary_traversal, all);
	memcpy(out_statistics, all, sizeof(uint64_t) * 6);
	return failed;
}

extern "C" int get_traversal_statistics_of_tree(application_t* app, VkBool32 wide_tree, uint64_t out_statistics[12]) {
	const frame_pipeline* frames = (const frame_pipeline*) app->shading_pass.wavefront;
	const wavefront_buffers* w = frames ? &frames->contexts[frames->last].buffers : NULL;
	const acceleration_structure_t* structure = &app->scene.acceleration_structure;
	if (!w || !w->ray_directions || !app->shading_pass.use_ray_tracing || app->shading_pass.inline_rays || (wide_tree && !structure->wide_nodes)) {
		printf("get_traversal_statistics_of_tree() needs a frame rendered with wavefront shadow rays (and the tree it is asked about).\n");
		return 1;
	}
	unsigned long long* counters = NULL;
	if (hip_failed(hipMalloc(&counters, sizeof(unsigned long long) * 12), "allocating traversal counters")) return 1;
	hipStream_t stream = (hipStream_t) app->device.stream;
	(void) finish_frames(app);
	(void) hipMemsetAsync(counters, 0, sizeof(unsigned long long) * 12, stream);
	bvh_view bvh = make_bvh_view(structure);
	// (the queues of the most recent launch - the last band of the last frame - with the sizes the resolve kernel kept)
	ray_stream rays = {w->ray_directions, w->ray_records, w->ray_origins, w->ray_queue_size + kRayCounterCount, w->queue_capacity, w->thread_bits, w->thread_count};
	if (wide_tree) k_traversal_statistics_wide<<<dim3(16, kRayQueueCount), 256, 0, stream>>>(bvh, (const uint4*) structure->wide_nodes, rays, frames->wide_stack_lds, counters);
	else k_traversal_statistics<<<dim3(16, kRayQueueCount), 256, 0, stream>>>(bvh, rays, counters);
	int failed = vkr_copy_to_host(out_statistics, counters, sizeof(uint64_t) * 12, &app->device);
	(void) hipFree(counters);
	return failed;
}

// sums the words of the most recent launch's shaft table
// out[0]: clear pairs, out[1 ... 5]: pairs that are traced, by reason (kShaftNoPixels ... kShaftTriangle, light_shafts.h),
// out[6]: anything else, out[7]: pairs with an occluder list, out[8]: triangles on those lists
__global__ void __launch_bounds__(256) k_count_clear_shafts(const uint32_t* words, size_t count, unsigned long long* out) {
	for (size_t i = (size_t) blockIdx.x * 256u + threadIdx.x; i < count; i += (size_t) gridDim.x * 256u) {
		uint32_t verdict = words[i] & 0xFFu;
		uint32_t slot = verdict == kShaftClear ? 0u : (verdict == kShaftList ? 7u : (verdict >= kShaftNoPixels && verdict <= kShaftTriangle ? 1u + (verdict - kShaftNoPixels) : 6u));
		atomicAdd(out + slot, 1ull);
		if (verdict == kShaftList) atomicAdd(out + 8, (unsigned long long) ((words[i] >> 8) & 0x1Fu));
	}
}

extern "C" int get_light_shaft_statistics(application_t* app, uint64_t out_statistics[12]) {
	memset(out_statistics, 0, sizeof(uint64_t) * 12);
	const shading_pass_t* pass = &app->shading_pass;
	const frame_pipeline* frames = (const frame_pipeline*) pass->wavefront;
	const wavefront_buffers* w = frames ? &frames->contexts[frames->last].buffers : NULL;
	if (!w || !pass->last_shaft_groups || !w->shaft_clear) return 0;  // the last frame ran without the shaft test: all zero
	if (finish_frames(app)) return 1;
	size_t words = (size_t) pass->last_shaft_groups * app->scene_specification.polygonal_light_count;
	unsigned long long* counter = NULL;
	if (hip_failed(hipMalloc(&counter, sizeof(unsigned long long) * 16), "allocating counters")) return 1;
	hipStream_t stream = (hipStream_t) app->device.stream;
	(void) hipMemsetAsync(counter, 0, sizeof(unsigned long long) * 16, stream);
	k_count_clear_shafts<<<256, 256, 0, stream>>>(w->shaft_clear, words, counter);
	unsigned long long counts[16] = {0};
	int failed = vkr_copy_to_host(counts, counter, sizeof(counts), &app->device);
	(void) hipFree(counter);
	out_statistics[0] = words;
	out_statistics[1] = counts[0];
	out_statistics[2] = pass->last_shaft_groups;
	out_statistics[3] = app->scene_specification.polygonal_light_count;
	for (int i = 0; i != 6; ++i) out_statistics[4 + i] = counts[1 + i];
	out_statistics[10] = counts[7];
	out_statistics[11] = counts[8];
	return failed;
}

// (diagnostics, VKR_SHAFT_COUNTERS=1) {steps, triangle batches, walks} of the most recent launch's shaft kernel
extern "C" int get_light_shaft_work(application_t* app, uint64_t out_work[3]) {
	out_work[0] = out_work[1] = out_work[2] = 0;
	const shading_pass_t* pass = &app->shading_pass;
	const frame_pipeline* frames = (const frame_pipeline*) pass->wavefront;
	const wavefront_buffers* w = frames ? &frames->contexts[frames->last].buffers : NULL;
	if (!w || !pass->last_shaft_groups || !w->shaft_clear || !getenv("VKR_SHAFT_COUNTERS") || finish_frames(app)) return 0;
	size_t words = ((size_t) pass->last_shaft_groups * app->scene_specification.polygonal_light_count + 1u) & ~(size_t) 1u;
	return vkr_copy_to_host(out_work, w->shaft_clear + words, 3 * sizeof(uint64_t), &app->device);
}

// (diagnostics) the verdict words of the most recent launch, [patch][light]; returns how many were written
extern "C" uint64_t read_back_light_shafts(application_t* app, uint32_t* out_words, uint64_t capacity) {
	const shading_pass_t* pass = &app->shading_pass;
	const frame_pipeline* frames = (const frame_pipeline*) pass->wavefront;
	const wavefront_buffers* w = frames ? &frames->contexts[frames->last].buffers : NULL;
	if (!w || !pass->last_shaft_groups || !w->shaft_clear || finish_frames(app)) return 0;
	uint64_t words = (uint64_t) pass->last_shaft_groups * app->scene_specification.polygonal_light_count;
	if (words > capacity) words = capacity;
	if (vkr_copy_to_host(out_words, w->shaft_clear, sizeof(uint32_t) * words, &app->device)) return 0;
	return words;
}

extern "C" uint64_t get_last_ray_count(const application_t* app) {
	unsigned long long rays = 0;
	const shading_pass_t* pass = &app->shading_pass;
	if (!pass->ray_counter || !pass->use_ray_tracing || !pass->last_frame_traced_rays || !pass->frame_counter) return 0;
	// (the kernels of the frame - the shading kernel with inline rays, else the resolve kernel of every
	// band - added their rays to the frame's counter)
	if (finish_frames((application_t*) app)) return 0;
	if (vkr_copy_to_host(&rays, (const unsigned long long*) pass->ray_counter + (pass->frame_counter - 1u) % 16u, sizeof(rays), &app->device)) return 0;
	return rays;
}

// ---- slabs -> frame ----------------------------------------------------------------

template <typename PIXEL>
__global__ void __launch_bounds__(256) k_assemble_frame(const PIXEL* slabs, PIXEL* frame, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tiles_x, uint32_t rank_count, uint64_t slab_stride) {
	uint32_t px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
	if (px >= width || py >= height) return;
	uint32_t tx = px / tile_size, ty = py / tile_size;
	uint32_t tile = ty * tiles_x + tx;
	uint32_t rank = tile % rank_count, local_tile = tile / rank_count;
	uint32_t ix = px - tx * tile_size, iy = py - ty * tile_size;
	frame[(size_t) py * width + px] = slabs[rank * slab_stride + (size_t) local_tile * tile_size * tile_size + (size_t) iy * tile_size + ix];
}

template <typename PIXEL>
static int assemble_slabs(application_t* app, const void* gathered_slabs, void* out_frame, hipStream_t stream) {
	shade_params p;
	memset(&p, 0, sizeof(p));
	p.width = app->swapchain.extent.width;
	p.height = app->swapchain.extent.height;
	uint32_t grid_blocks = 0;
	application_t first = *app;
	first.tile_schedule.rank = 0;
	fill_tile_schedule(p, &first, grid_blocks);
	uint64_t slab_stride = (uint64_t) grid_blocks * 256;
	dim3 grid((p.width + 15) / 16, (p.height + 15) / 16);
	k_assemble_frame<PIXEL><<<grid, 256, 0, stream>>>((const PIXEL*) gathered_slabs, (PIXEL*) out_frame,
		p.width, p.height, p.tile_size, p.tiles_x, p.rank_count, slab_stride);
	return hip_failed(hipGetLastError(), "assembling the frame");
}

extern "C" int assemble_frame_from_slabs(application_t* app, const void* gathered_slabs, void* out_radiance) {
	// (the radiance target may still be written by frames in flight)
	if (finish_frames(app)) return 1;
	return assemble_slabs<float4>(app, gathered_slabs, out_radiance ? out_radiance : app->render_targets.radiance, (hipStream_t) app->device.stream);
}

extern "C" int assemble_encoded_frame_from_slabs(application_t* app, const void* gathered_slabs, void* out_encoded) {
	if (finish_frames(app)) return 1;
	return assemble_slabs<uint32_t>(app, gathered_slabs, out_encoded ? out_encoded : app->render_targets.encoded, (hipStream_t) app->device.stream);
}

// slabs of packed RGB8 (encode_slab_rgb8) -> RGBA8 frame; alpha of the encoded output is always
// 255.  A thread moves four pixels of one tile row: twelve bytes = three aligned dwords in
// (tile sizes are multiples of four), four pixels out.
__global__ void __launch_bounds__(256) k_assemble_frame_rgb8(const uint32_t* slabs, uint32_t* frame, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tiles_x, uint32_t rank_count, uint64_t slab_stride) {
	uint32_t px = 4u * (blockIdx.x * 64u + (threadIdx.x & 63u)), py = blockIdx.y * 4u + (threadIdx.x >> 6);
	if (px >= width || py >= height) return;
	uint32_t tx = px / tile_size, ty = py / tile_size;
	uint32_t tile = ty * tiles_x + tx;
	uint32_t rank = tile % rank_count, local_tile = tile / rank_count;
	uint32_t ix = px - tx * tile_size, iy = py - ty * tile_size;
	size_t pixel = rank * slab_stride + (size_t) local_tile * tile_size * tile_size + (size_t) iy * tile_size + ix;
	const uint32_t* source = slabs + 3 * (pixel / 4);
	uint32_t d0 = source[0], d1 = source[1], d2 = source[2];
	uint32_t out[4] = {d0 | 0xFF000000u, (d0 >> 24) | (d1 << 8) | 0xFF000000u, (d1 >> 16) | (d2 << 16) | 0xFF000000u, (d2 >> 8) | 0xFF000000u};
	uint32_t* target = frame + (size_t) py * width + px;
	for (uint32_t i = 0; i != 4 && px + i < width; ++i) target[i] = out[i];
}

static int assemble_rgb8_slabs(application_t* app, const void* gathered_slabs, void* out_encoded, hipStream_t stream) {
	shade_params p;
	memset(&p, 0, sizeof(p));
	p.width = app->swapchain.extent.width;
	p.height = app->swapchain.extent.height;
	uint32_t grid_blocks = 0;
	application_t first = *app;
	first.tile_schedule.rank = 0;
	fill_tile_schedule(p, &first, grid_blocks);
	if (p.tile_size % 4 != 0) {
		printf("assemble_rgb8_frame_from_slabs() needs a tile size that is a multiple of four.\n");
		return 1;
	}
	dim3 grid((p.width + 255) / 256, (p.height + 3) / 4);
	k_assemble_frame_rgb8<<<grid, 256, 0, stream>>>((const uint32_t*) gathered_slabs, (uint32_t*) (out_encoded ? out_encoded : app->render_targets.encoded),
		p.width, p.height, p.tile_size, p.tiles_x, p.rank_count, (uint64_t) grid_blocks * 256);
	return hip_failed(hipGetLastError(), "assembling the frame");
}

extern "C" int assemble_rgb8_frame_from_slabs(application_t* app, const void* gathered_slabs, void* out_encoded) {
	if (finish_frames(app)) return 1;
	return assemble_rgb8_slabs(app, gathered_slabs, out_encoded, (hipStream_t) app->device.stream);
}

// For host/slab_exchange.c: the scatter of gathered slabs on a stream of the caller's choice
// (the exchange stream, so that it does not wait for later frames), format as slab_format_t
extern "C" int vkr_assemble_slabs_on_stream(application_t* app, const void* gathered_slabs, void* out_frame, int format, void* stream) {
	if (format == 0) return assemble_slabs<float4>(app, gathered_slabs, out_frame ? out_frame : app->render_targets.radiance, (hipStream_t) stream);
	return assemble_rgb8_slabs(app, gathered_slabs, out_frame, (hipStream_t) stream);
}

// ---- output encoding (shading_pass.frag.glsl:871-892, srgb_utility.glsl) --------------

__device__ __forceinline__ float linear_to_srgb(float v) {
	v = gclamp(v, 0.0f, 1.0f);
	return (v <= 0.0031308f) ? (12.92f * v) : (1.055f * powf(v, 1.0f / 2.4f) - 0.055f);
}
__device__ __forceinline__ float srgb_to_linear(float v) {
	v = gclamp(v, 0.0f, 1.0f);
	return (v <= 0.04045f) ? ((1.0f / 12.92f) * v) : powf(fmaf(v, 1.0f / 1.055f, 0.055f / 1.055f), 2.4f);
}
__device__ __forceinline__ uint32_t to_unorm8(float v) {
	v = gclamp(v, 0.0f, 1.0f);
	return (uint32_t) (v * 255.0f + 0.5f);
}

// to_unorm8(linear_to_srgb(v)) without the pow: the code is the number of thresholds
// T[c] = srgb_to_linear((c - 0.5) / 255), c = 1 ... 255, that v has reached.  A hardware
// log2 / exp2 estimate is off by less than one code; the two neighbouring thresholds settle it.
// (Three powf per pixel made the encode kernel as expensive as tracing config 2's shadow rays.)
__device__ float g_srgb_code_thresholds[257];

__global__ void k_fill_srgb_code_thresholds() {
	uint32_t c = threadIdx.x;
	double x = ((double) c - 0.5) / 255.0;
	double linear = (x <= 0.04045) ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
	float t = (float) linear;
	if ((double) t < linear) t = __uint_as_float(__float_as_uint(t) + 1u);
	g_srgb_code_thresholds[c] = (c == 0) ? 0.0f : t;
	if (c == 0) g_srgb_code_thresholds[256] = __builtin_inff();
}

// Called by create_hip_device() with the device selected: fills the table on THAT device and waits, so that
// every stream of every thread that later encodes on the device finds it (the table is per device and its
// content does not depend on who fills it: no host-side state, filling it again is harmless).
extern "C" int vkr_fill_device_tables(void* stream) {
	k_fill_srgb_code_thresholds<<<1, 256, 0, (hipStream_t) stream>>>();
	if (hip_failed(hipGetLastError(), "filling the sRGB thresholds")) return 1;
	return hip_failed(hipStreamSynchronize((hipStream_t) stream), "filling the sRGB thresholds");
}

__device__ __forceinline__ uint32_t srgb_code(float v) {
	v = gclamp(v, 0.0f, 1.0f);  // (NaN -> 0 like to_unorm8(linear_to_srgb(NaN)))
	float estimate = (v <= 0.0031308f) ? (12.92f * v) : fmaf(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (1.0f / 2.4f)), -0.055f);
	uint32_t c = (uint32_t) fmaf(gclamp(estimate, 0.0f, 1.0f), 255.0f, 0.5f);
	c = (v < g_srgb_code_thresholds[c]) ? c - 1u : c;
	c = (v >= g_srgb_code_thresholds[c + 1u]) ? c + 1u : c;
	return c;
}

__device__ __forceinline__ uint32_t encode_pixel(float4 c, uint32_t frame_bits, int output_linear_rgb) {
	uint32_t r, g, b, a;
	if (frame_bits == 0) {
		// an *_SRGB target encodes in hardware, any other gets the transfer function in the shader
		r = srgb_code(c.x); g = srgb_code(c.y); b = srgb_code(c.z);
		a = to_unorm8(c.w);
	}
	else {
		uint32_t mask = (frame_bits == 1) ? 0xFF : 0xFF00, shift = (frame_bits == 1) ? 0 : 8;
		uint32_t h0 = (uint32_t) __half_as_ushort(__float2half_rn(c.x)) | ((uint32_t) __half_as_ushort(__float2half_rn(c.y)) << 16);
		uint32_t h1 = (uint32_t) __half_as_ushort(__float2half_rn(c.z));
		float v[3] = {
			(float) ((h0 & mask) >> shift) * (1.0f / 255.0f),
			(float) ((((h0 & 0xFFFF0000u) >> 16) & mask) >> shift) * (1.0f / 255.0f),
			(float) ((h1 & mask) >> shift) * (1.0f / 255.0f)};
		uint32_t out[3];
		for (int j = 0; j != 3; ++j) out[j] = to_unorm8(output_linear_rgb ? linear_to_srgb(srgb_to_linear(v[j])) : v[j]);
		r = out[0]; g = out[1]; b = out[2];
		a = 255;
	}
	return r | (g << 8) | (b << 16) | (a << 24);
}

__global__ void __launch_bounds__(256) k_encode_output(const float4* radiance, uint32_t* encoded, uint64_t pixel_count, uint32_t frame_bits, int output_linear_rgb) {
	uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= pixel_count) return;
	encoded[i] = encode_pixel(radiance[i], frame_bits, output_linear_rgb);
}

// four pixels per thread: twelve bytes of packed RGB as three dwords
__global__ void __launch_bounds__(256) k_encode_output_rgb8(const float4* radiance, uint32_t* packed, uint64_t quad_count, uint32_t frame_bits, int output_linear_rgb) {
	uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= quad_count) return;
	uint32_t p0 = encode_pixel(radiance[4 * i], frame_bits, output_linear_rgb) & 0xFFFFFFu, p1 = encode_pixel(radiance[4 * i + 1], frame_bits, output_linear_rgb) & 0xFFFFFFu;
	uint32_t p2 = encode_pixel(radiance[4 * i + 2], frame_bits, output_linear_rgb) & 0xFFFFFFu, p3 = encode_pixel(radiance[4 * i + 3], frame_bits, output_linear_rgb) & 0xFFFFFFu;
	packed[3 * i] = p0 | (p1 << 24);
	packed[3 * i + 1] = (p1 >> 8) | (p2 << 16);
	packed[3 * i + 2] = (p2 >> 16) | (p3 << 8);
}

extern "C" int encode_output(application_t* app, VkBool32 output_linear_rgb) {
	if (finish_frames(app)) return 1;
	uint64_t pixels = (uint64_t) app->swapchain.extent.width * app->swapchain.extent.height;
	if (!app->render_targets.radiance || !app->render_targets.encoded) return 1;
	k_encode_output<<<(uint32_t) ((pixels + 255) / 256), 256, 0, (hipStream_t) app->device.stream>>>((const float4*) app->render_targets.radiance, (uint32_t*) app->render_targets.encoded,
		pixels, app->screenshot.frame_bits, output_linear_rgb ? 1 : 0);
	note_target_reader(app);
	return hip_failed(hipGetLastError(), "encoding the output");
}

extern "C" int encode_slab(application_t* app, const void* slab_radiance, void* slab_encoded, uint64_t pixel_count, VkBool32 output_linear_rgb) {
	if (finish_frames(app)) return 1;
	if (!slab_radiance || !slab_encoded) return 1;
	k_encode_output<<<(uint32_t) ((pixel_count + 255) / 256), 256, 0, (hipStream_t) app->device.stream>>>((const float4*) slab_radiance, (uint32_t*) slab_encoded,
		pixel_count, app->screenshot.frame_bits, output_linear_rgb ? 1 : 0);
	note_target_reader(app);
	return hip_failed(hipGetLastError(), "encoding the slab");
}

extern "C" int encode_slab_rgb8(application_t* app, const void* slab_radiance, void* slab_rgb8, uint64_t pixel_count, VkBool32 output_linear_rgb) {
	if (finish_frames(app)) return 1;
	if (!slab_radiance || !slab_rgb8 || pixel_count % 4 != 0) {
		printf("encode_slab_rgb8() needs buffers and a pixel count that is a multiple of four (slabs are).\n");
		return 1;
	}
	k_encode_output_rgb8<<<(uint32_t) ((pixel_count / 4 + 255) / 256), 256, 0, (hipStream_t) app->device.stream>>>((const float4*) slab_radiance, (uint32_t*) slab_rgb8,
		pixel_count / 4, app->screenshot.frame_bits, output_linear_rgb ? 1 : 0);
	note_target_reader(app);
	return hip_failed(hipGetLastError(), "encoding the slab");
}

// ---- primary visibility ------------------------------------------------------------

__global__ void __launch_bounds__(256) k_primary_visibility(const uint8_t* constants, bvh_view bvh, uint32_t* visibility, uint32_t width, uint32_t height, float near, float far) {
	uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint32_t px = blockIdx.x * 16 + ((wave & 1) << 3) + (lane & 7);
	uint32_t py = blockIdx.y * 16 + ((wave >> 1) << 3) + (lane >> 3);
	if (px >= width || py >= height) return;
	float fx = (float) px, fy = (float) py;
	f3 ray = mk3(
		(load_f(constants, 96) * fx + load_f(constants, 100) * fy) + load_f(constants, 104),
		(load_f(constants, 112) * fx + load_f(constants, 116) * fy) + load_f(constants, 120),
		(load_f(constants, 128) * fx + load_f(constants, 132) * fy) + load_f(constants, 136));
	f3 origin = load_f3(constants, 144);
	// The unnormalised ray direction has view-space depth 1 (it is the unprojection of
	// clip-space w = 1), so the depth range [near, far] is the parameter range.
	visibility[(size_t) py * width + px] = closest_front_hit(bvh, origin, ray, near, far);
}

extern "C" int render_visibility_pass(application_t* app) {
	// frames in flight read the visibility buffer that this pass overwrites
	if (finish_frames(app)) return 1;
	mark_inputs_changed(app);
	shading_pass_t* pass = &app->shading_pass;
	const acceleration_structure_t* as = &app->scene.acceleration_structure;
	if (!as->triangle_vertices || !pass->constants_device) {
		printf("The visibility pass needs an acceleration structure and a shading pass.\n");
		return 1;
	}
	if (upload_constants(app, (hipStream_t) app->device.stream)) return 1;
	bvh_view bvh = make_bvh_view(as);
	uint32_t width = app->swapchain.extent.width, height = app->swapchain.extent.height;
	dim3 grid((width + 15) / 16, (height + 15) / 16);
	k_primary_visibility<<<grid, 256, 0, (hipStream_t) app->device.stream>>>((const uint8_t*) pass->constants_device, bvh, (uint32_t*) app->render_targets.visibility_buffer,
		width, height, app->scene_specification.camera.near, app->scene_specification.camera.far);
	return hip_failed(hipGetLastError(), "rendering the visibility pass");
}

// ---- transfers -------------------------------------------------------------------

extern "C" int read_back_radiance(application_t* app, float* host_rgba) {
	if (finish_frames(app)) return 1;
	size_t pixels = (size_t) app->swapchain.extent.width * app->swapchain.extent.height;
	return vkr_copy_to_host(host_rgba, app->render_targets.radiance, sizeof(float) * 4 * pixels, &app->device);
}
extern "C" int read_back_encoded(application_t* app, uint8_t* host_rgba8) {
	if (finish_frames(app)) return 1;
	size_t pixels = (size_t) app->swapchain.extent.width * app->swapchain.extent.height;
	return vkr_copy_to_host(host_rgba8, app->render_targets.encoded, 4 * pixels, &app->device);
}
extern "C" int read_back_visibility(application_t* app, uint32_t* host_primitives) {
	if (finish_frames(app)) return 1;
	size_t pixels = (size_t) app->swapchain.extent.width * app->swapchain.extent.height;
	return vkr_copy_to_host(host_primitives, app->render_targets.visibility_buffer, sizeof(uint32_t) * pixels, &app->device);
}
extern "C" int upload_visibility(application_t* app, const uint32_t* host_primitives) {
	// a blocking copy outside the streams: nothing may still be reading the old buffer
	if (wait_for_device(&app->device)) return 1;
	mark_inputs_changed(app);
	size_t pixels = (size_t) app->swapchain.extent.width * app->swapchain.extent.height;
	if (hip_failed(hipMemcpy(app->render_targets.visibility_buffer, host_primitives, sizeof(uint32_t) * pixels, hipMemcpyHostToDevice), "uploading the visibility buffer")) return 1;
	return 0;
}
