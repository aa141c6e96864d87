// The per-pixel shading program as a HIP kernel for gfx950 (CDNA4).
//
// One lane per pixel; a workgroup is one wave64 and covers an 8x8 pixel patch (the four patches
// of a 16x16 block run on the same XCD), so neighbouring lanes read neighbouring triangles,
// LTC texels and noise texels and mostly agree on the clipping / sector branches.
// All uniform inputs (frame constants, light records) are read through wave-uniform
// addresses in the constant address space and end up in SGPRs.
//
// Follows reference src/shaders/shading_pass.frag.glsl (main :824-866,
// evaluate_polygonal_light_shading :329-711, get_shading_data :721-822) with the
// variant axes of src/main.c:752-792 as template parameters:
//   STRATEGY  sampling_strategies_t          (SAMPLING_STRATEGIES_*)
//   TECHNIQUE 0 projected solid angle, 1 its biased variant, 2 solid angle,
//             3 clipped solid angle          (SAMPLE_POLYGON_*)
//   V         MAX_POLYGON_VERTEX_COUNT
//   RAYS      TRACE_SHADOW_RAYS
// Light count, sample count and the MIS heuristic are wave-uniform run-time values.
#pragma once
#include "polygon_sampling.h"
#include "related_work.h"
#include "lbvh.h"

namespace vkr {

enum { kStrategyDiffuseOnly = 0, kStrategyDiffuseGgxMis = 1, kStrategySeparately = 2, kStrategyMis = 3, kStrategyRandom = 4 };
enum { kTechniquePsa = 0, kTechniquePsaBiased = 1, kTechniqueSolidAngle = 2, kTechniqueClippedSolidAngle = 3, kTechniqueBaseline = 4, kTechniqueAreaTurk = 5,
	kTechniqueUrena = 6, kTechniqueArvoSolidAngle = 7, kTechniqueHartBilinear = 8, kTechniqueHartBilinearClipping = 9,
	kTechniqueHartBiquadratic = 10, kTechniqueHartBiquadraticClipping = 11, kTechniquePsaArvo = 12, kTechniqueCount = 13 };
enum { kMisBalance = 0, kMisPower = 1, kMisWeighted = 2, kMisOptimalClamped = 3, kMisOptimal = 4 };

struct shade_params {
	// per_frame_constants_t + packed lights, byte image of write_constants()
	const uint8_t* constants;
	uint32_t light_count, max_light_vertex_count, sample_count;
	int32_t mis_heuristic;
	int32_t show_polygonal_lights;
	// mesh (layouts: include/vkr_scene.h)
	const uint2* positions;
	const uint2* normals_and_tex_coords;
	const uint8_t* material_indices;
	const float* material_constants;
	// textured scenes: 8 floats per pixel of the frame written by k_resolve_materials (else NULL),
	// and the inputs of that kernel
	const float* pixel_materials;
	const uint32_t* texture_descriptors;
	const uint32_t* texels;
	const float* srgb_table;
	// light textures (include/vkr_shading_pass.h light_textures_t): (first texel, width, height, 0)
	// per texture and RGBA fp32 texels; NULL when no light uses a texturing technique
	const uint4* light_texture_descriptors;
	const float4* light_texels;
	// G-buffer in, radiance out
	const uint32_t* visibility;
	float4* out_radiance;
	uint32_t width, height;
	// tables
	const uint2* ltc_rgba;
	const uint32_t* ltc_rg;
	uint32_t ltc_resolution, ltc_layer_count;
	const uint2* noise;
	uint32_t noise_width, noise_height;
	bvh_view bvh;
	// tile schedule (include/vkr_shading_pass.h tile_schedule_t)
	uint32_t tile_size, rank, rank_count, tiles_x, tile_count;
	uint32_t slab_layout;  // 0: out_radiance is the row-major frame (single rank), 1: this rank's dense slab
	unsigned long long* ray_counter;
	// wavefront mode (RAYS == kRaysDeferred): per-thread streams of pending terms and
	// the compacted shadow-ray queue, see "wavefront" below
	uint8_t* codes;
	float* terms_visible;
	float* terms_hidden;
	// colour of a pixel before its shadowed terms (light display), one per thread; kept apart
	// from out_radiance so that two frames in flight may share the output target
	float4* base_color;
	// kRayQueueCount independent queues, each with its own counter (one counter for the
	// whole chip saturates at ~88 atomics / us).  Each XCD owns 64 of them: they are filled
	// by the shading workgroups and drained by the tracing workgroups of that XCD only.
	// A queued ray is 20 bytes in two parallel arrays, [queue][slot]: (direction, t_max) and a
	// record word = thread | code cursor << ray_thread_bits (ray_record()); its origin is the
	// shading position of the pixel, stored once per thread (ray_origins) - a pixel queues up to
	// 2 L S rays from the same point (config 3: 14 on average).
	float4* ray_directions;
	uint32_t* ray_records;
	float4* ray_origins;
	uint32_t* ray_queue_size;
	uint32_t ray_queue_capacity, ray_thread_bits;
	uint32_t thread_count, max_terms, max_codes;
	// light shafts (light_shafts.h): [shading workgroup][light] = 1 when no shadow ray of that 8x8 patch toward that
	// light can be blocked - its terms are then written as final ones and no ray is queued (any other value: why the
	// pair is not clear); NULL: every ray is traced
	const uint32_t* shaft_clear;
	// ... and per light the rectangle (u_min, v_min, u_max, v_max in the light's plane space) that the shafts were built
	// around: only a ray that meets the light's plane inside it may skip the tracing (shaft_holds_ray)
	const float4* shaft_rectangles;
	// ... and per pair whose verdict is kShaftList the triangles that its rays may meet: kShaftListMax entries of
	// kShaftListEntry floats each (a vertex and the two edges that leave it, as ray_triangle computes them); NULL: no lists
	const float* shaft_lists;
	// first 16x16 pixel block of this launch in the rank's schedule (a frame may be rendered as
	// several launches, "bands", each with wavefront buffers of its own size)
	uint32_t first_block, block_count;
	// slots a wave reserves in its queue per atomic (0: exactly as many as it needs, one atomic per
	// push; used when a lane queues only a ray or two).  Unused slots are left as null rays.
	uint32_t ray_block;
	// tuning knobs (host: environment, see shading_pass.hip)
	uint32_t refill_threshold;
	// wavefront mode: 1 for the first launch of a frame, whose resolve kernel STORES its ray count in ray_counter; those of
	// the frame's later bands add to it (the resolves of consecutive launches are ordered).  Until round 4 a memset in
	// front of every frame cleared the counter: one more (tiny) kernel in the chain of every frame, which the GPU schedules
	// when it finds room - 4 to 800 us under load (profiles/r07b).
	uint32_t first_launch_of_frame;
	// the table of the second prepared polygon of every shading workgroup, for the kernel variants that keep only one in
	// LDS (psa_table_in_memory(V)): [workgroup of the launch][slot][thread] float2; NULL for all other variants
	float2* psa_table_memory;
	// 1: a wave's region of psa_table_memory is that of the hardware slot it runs in (hardware_wave_slot(), kWaveSlots regions),
	// 0: that of its workgroup
	uint32_t psa_table_by_wave_slot;
	// error display (ERROR_INDEX of the reference; the two constants of error_to_color
	// that the GLSL compiler folds: 10^4.99 and 20 / (5 log2 10), computed on the host)
	uint32_t error_index;
	float error_max, error_scale;
};

constexpr uint32_t kRayQueueCount = 512;  // 8 XCDs x 64 (one queue per lane when scanning sizes)
constexpr uint32_t kCursorStride = 32;    // one 128-byte line per XCD work cursor
constexpr uint32_t kRayCounterCount = kRayQueueCount + 16 * kCursorStride;  // queue sizes, then per-XCD cursors, then per-XCD ray counts
constexpr uint32_t kRayCountOffset = kRayQueueCount + 8 * kCursorStride;  // real rays queued from XCD x: counter kRayCountOffset + x * kCursorStride
constexpr uint32_t kNullRay = 0xFFFFFFFFu;  // code index of a queue slot that was reserved but not used
constexpr uint32_t kRayChunk = 256;       // most rays a wave claims per atomic in trace_shadow_rays

// How shadow rays are traced (template parameter RAYS):
//   kRaysNone      TRACE_SHADOW_RAYS = 0
//   kRaysInline    every lane walks the BVH inside the shading kernel
//   kRaysDeferred  wavefront: the shading kernel appends rays to a queue compacted
//                  with wave ballots, a lean high-occupancy kernel traces them, a
//                  resolve kernel replays the per-pixel sums in the original order
//   kRaysDeferredBlocks  the same with queue slots reserved a block at a time (push_ray)
enum { kRaysNone = 0, kRaysInline = 1, kRaysDeferred = 2, kRaysDeferredBlocks = 3 };
constexpr bool is_deferred(int rays) { return rays == kRaysDeferred || rays == kRaysDeferredBlocks; }

// Round 6: from V = 6 on only ONE of the two polygon tables of the two-technique kernels is in LDS; the specular polygon's is
// in device memory, [region][slot][thread] like the LDS table (shade_params::psa_table_memory).  Why: these kernels live on
// resident waves.  The V = 7 kernel of BASELINE config 4 runs 16.93 / 18.50 / 20.51 / 23.15 ms with 9 / 8 / 7 / 6 waves per CU
// (profiles/r10f/waves_per_cu.jsonl: LDS that is asked for and not used) - every wave more is worth 7 - 9 % -, and its two
// tables (15 360 B + 1 312 B of other LDS = 14 granules of 1 280 B) held it at nine where its registers allow twelve:
// 16.9 -> 14.4 ms with one table.  V <= 5 fits twelve waves with both tables in LDS and keeps them there.
// Tried and not adopted (profiles/r10m, r10o): FOUR waves per SIMD for these kernels (128 VGPRs, 35 - 45 dwords per lane in
// scratch memory, one table in memory for every V): config 3 1.153 -> 1.119 ms per frame, the target shape 0.431 -> 0.409,
// config 4 16.45 -> 15.90 - but the spills go through the L2 to the fabric: FETCH_SIZE + WRITE_SIZE of a config-3 launch
// 0.20 -> 1.33 GB, of a config-4 frame 2 -> 25 GB.  3 % of time for six to twelve times the memory traffic: no.  (Five waves
// lose outright, 1.255 ms; the one-technique kernels lose at four, config 2 0.139 -> 0.155.)
#ifndef VKR_PSA_MEMORY_FROM
#define VKR_PSA_MEMORY_FROM 6
#endif
constexpr bool psa_table_in_memory(int v) { return v >= VKR_PSA_MEMORY_FROM; }
constexpr uint32_t psa_table_memory_bytes_per_workgroup(int v) { return (2u * (uint32_t) v + 1u) * 64u * 8u; }
// Where a wave's table lies in that memory: in the region of its workgroup (the default), or - shade_params::
// psa_table_by_wave_slot, VKR_PSA_TABLE_INDEX=slot - in the region of the HARDWARE SLOT the wave occupies, (XCD, shader engine,
// shader array, CU, SIMD, wave slot) from HW_REG_XCC_ID and HW_REG_HW_ID, unique among the waves that are resident at any time
// whatever kernel, stream or frame they belong to (check_hardware_wave_slots() of the C-ABI tests that on the device): one
// buffer of kWaveSlots regions for the whole pass, written again and again by the waves that follow each other in a slot.  The
// idea was that such a region stays in its XCD's L2; the counters say the table stores reach the fabric either way (config 4:
// WRITE_SIZE 5.2 GB per frame with both schemes, profiles/r10o), so it is an option, not the default.
constexpr uint32_t kWaveSlots = 1u << 17;
VKR_DEV uint32_t hardware_wave_slot() {
	// s_getreg_b32 simm16 = (size - 1) << 11 | offset << 6 | register: HW_REG_HW_ID = 4 (wave_id [3:0], simd_id [5:4], pipe_id [7:6],
	// cu_id [11:8], sh_id [12], se_id [15:13]), HW_REG_XCC_ID = 20 (xcc_id [3:0])
	uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
	uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
	// (the pipe that dispatched the wave is not part of where it runs)
	return ((xcc & 7u) << 14) | (((hw >> 8) & 0xFFu) << 6) | (hw & 0x3Fu);
}
// codes of the per-thread term stream written in deferred mode
// (kCodePendingHiddenNaN: the value of the blocked term is not stored because it can only be NaN - every
// estimator but the plain optimal MIS heuristic computes it as 0 x something, i.e. +-0 or NaN, and a NaN
// in any channel sends the whole pixel to the shader's NaN guard, shading_pass.frag.glsl:861-864)
// triangles on the occluder list of a (patch, light) pair (light_shafts.h); 0: no lists
#ifndef VKR_SHAFT_LIST
#define VKR_SHAFT_LIST 12
#endif
constexpr uint32_t kShaftListMax = VKR_SHAFT_LIST;
constexpr uint32_t kShaftListEntry = 12;  // floats of a list entry: p0, e1, e2, each padded to four
// (In every arithmetic mode since round 6: the list test must be the tracing kernel's test to the bit, and the fast mode's
// translation units contract a b + c into fused operations where the tracing kernel's do not - so the triangle test itself,
// lbvh.h ray_triangle_edges, is compiled without contraction wherever it is compiled.)
constexpr bool kUseShaftLists = kShaftListMax != 0u;
// The final terms of a light that needs no ray are added up here instead of by the resolve kernel.  (In the fast mode the
// term is made opaque before it is added: a product contracted into the sum would make the sum formed here differ from the one
// the resolve kernel forms, and a frame would depend on whether the shaft test is on.)
#ifndef VKR_SUM_FINAL_TERMS
#define VKR_SUM_FINAL_TERMS 1
#endif
enum { kCodeEnd = 0, kCodePending = 1, kCodeVisible = 2, kCodePendingWithHidden = 3, kCodeEndOfLight = 4, kCodeFinal = 5, kCodePendingHiddenNaN = 6 };
// Byte index of code `cursor` of thread `tid`: four consecutive codes of a thread share one 32-bit
// word ([cursor / 4][thread] words), so that the resolve kernel fetches four codes per load and can
// request their terms together instead of walking a chain of dependent one-byte loads
VKR_DEV size_t code_slot(uint32_t thread_count, uint32_t cursor, uint32_t tid) {
	return (((size_t) (cursor >> 2) * thread_count + tid) << 2) | (cursor & 3u);
}
// The record word of a queued ray: which thread's term (thread_bits low bits) and which code of
// that thread's stream the tracing kernel flips when the ray reaches the light.  The host checks
// that both fit (thread_count <= 2^thread_bits, max_codes <= 2^(32 - thread_bits)).
VKR_DEV uint32_t ray_record(uint32_t thread_bits, uint32_t tid, uint32_t cursor) { return tid | (cursor << thread_bits); }
VKR_DEV uint32_t ray_record_thread(uint32_t thread_bits, uint32_t record) { return record & ((1u << thread_bits) - 1u); }
VKR_DEV uint32_t ray_record_cursor(uint32_t thread_bits, uint32_t record) { return record >> thread_bits; }

// Reads of the constant buffer (frame constants and light records; written by the host before the
// launch, never by a kernel) go through the constant address space: the compiler may then use scalar
// loads for the wave-uniform addresses and keep or re-load the values as it likes.  As plain global
// loads they became vector loads with a full s_waitcnt inside the sampling loops - after the first
// store of a term nothing proves to the compiler that the buffer is still what it was.
typedef const __attribute__((address_space(4))) float* constant_float_pointer;
typedef const __attribute__((address_space(4))) uint32_t* constant_uint_pointer;
VKR_DEV float load_f(const uint8_t* base, uint32_t offset) { return *(constant_float_pointer) (uintptr_t) (base + offset); }
VKR_DEV uint32_t load_u(const uint8_t* base, uint32_t offset) { return *(constant_uint_pointer) (uintptr_t) (base + offset); }
VKR_DEV f3 load_f3(const uint8_t* base, uint32_t offset) { return mk3(load_f(base, offset), load_f(base, offset + 4), load_f(base, offset + 8)); }

VKR_DEV float unorm16(uint32_t v) {
#if VKR_FAST_MATH
	return (float) v * (1.0f / 65535.0f);
#else
	return divide((float) v, 65535.0f);
#endif
}

// ---- noise (noise_utility.glsl:63-103) ---------------------------------------------

// The texel of the NEXT fetch is requested as soon as the current one has been unpacked (VKR_NOISE_AHEAD):
// a fetch is the one vector load of a sample, and it used to be waited for right where it was issued -
// which, on this hardware, also means waiting for every store before it: loads and stores share the
// vector-memory counter and complete out of order with each other, so the wait for a load with stores in
// flight is a wait for all of them (s_waitcnt vmcnt(0): the terms and rays of the sample before, a round
// trip to the L2).  With the request a sample ahead and the wait pinned in front of the sample's own stores
// (settle_noise(), called by accumulate()) both the load and the earlier stores have long completed
// when the wave asks.  Two more live registers.
#ifndef VKR_NOISE_AHEAD
#define VKR_NOISE_AHEAD 1
#endif
struct noise_accessor {
	float n0, n1, n2, n3;
	uint32_t available, px, py, sample_index;
	uint32_t ahead_x, ahead_y;  // texel of fetch number sample_index, on its way or there
};

VKR_DEV uint2 fetch_noise_texel(const shade_params& p, const noise_accessor& a) {
	const uint8_t* c = p.constants;
	uint32_t s = a.sample_index;
	uint32_t r0 = load_u(c, 208), r1 = load_u(c, 212), r2 = load_u(c, 216), r3 = load_u(c, 220);
	if (s & 2) { uint32_t t0 = r0, t1 = r1; r0 = r2; r1 = r3; r2 = t0; r3 = t1; }
	if (s & 1) { r0 = r1; r1 = r2; r2 = r3; }
	uint32_t shift = (s & 124) >> 2;
	uint32_t layer = (r2 + s) & load_u(c, 192);
	uint32_t sx = (a.px + (r0 >> shift)) & load_u(c, 184);
	uint32_t sy = (a.py + (r1 >> shift)) & load_u(c, 188);
	return p.noise[((size_t) layer * p.noise_height + sy) * p.noise_width + sx];
}

VKR_DEV noise_accessor make_noise_accessor(const shade_params& p, uint32_t px, uint32_t py) {
	noise_accessor a;
	a.n0 = a.n1 = a.n2 = a.n3 = 0.0f;
	a.available = 0; a.px = px; a.py = py; a.sample_index = 0;
	a.ahead_x = a.ahead_y = 0u;
#if VKR_NOISE_AHEAD
	uint2 texel = fetch_noise_texel(p, a);
	a.ahead_x = texel.x; a.ahead_y = texel.y;
#endif
	return a;
}

// The requested texel has to have arrived from here on (an empty statement that "uses" its registers)
VKR_DEV void settle_noise(noise_accessor& a) {
#if VKR_NOISE_AHEAD
	asm volatile("" : "+v"(a.ahead_x), "+v"(a.ahead_y));
#else
	(void) a;
#endif
}

VKR_DEV f2 next_noise_2(const shade_params& p, noise_accessor& a) {
	if (a.available <= 1) {
#if VKR_NOISE_AHEAD
		uint2 texel = make_uint2(a.ahead_x, a.ahead_y);
#else
		uint2 texel = fetch_noise_texel(p, a);
#endif
		a.n0 = unorm16(texel.x & 0xFFFF); a.n1 = unorm16(texel.x >> 16);
		a.n2 = unorm16(texel.y & 0xFFFF); a.n3 = unorm16(texel.y >> 16);
		a.available = 4;
		++a.sample_index;
#if VKR_NOISE_AHEAD
		// (one texel beyond the last one a pixel uses: the coordinates wrap, the read is harmless)
		uint2 ahead = fetch_noise_texel(p, a);
		a.ahead_x = ahead.x; a.ahead_y = ahead.y;
#endif
	}
	a.available -= 2;
	f2 r = mk2(a.n0, a.n1);
	a.n0 = a.n2; a.n1 = a.n3;
	return r;
}

// ---- G-buffer reconstruction (shading_pass.frag.glsl:721-822) ------------------------

struct shading_data {
	f3 position, normal, outgoing;
	float lambert_outgoing;
	f3 diffuse_albedo, fresnel_0;
	float roughness;
};

VKR_DEV f3 decode_position(uint2 q, f3 factor, f3 summand) {
	float px = (float) (q.x & 0x1FFFFF);
	float py = (float) (((q.x & 0xFFE00000u) >> 21) | ((q.y & 0x3FF) << 11));
	float pz = (float) ((q.y & 0x7FFFFC00u) >> 10);
	return mk3(fmaf(px, factor.x, summand.x), fmaf(py, factor.y, summand.y), fmaf(pz, factor.z, summand.z));
}

VKR_DEV f3 decode_normal(float ox, float oy) {
	const float factor = 2.0f * (65534.0f / 65535.0f);
	const float summand = -(32768.0f / 65535.0f) * factor;
	ox = fmaf(ox, factor, summand);
	oy = fmaf(oy, factor, summand);
	f3 n = mk3(ox, oy, 1.0f - fabsf(ox) - fabsf(oy));
	float sx = (ox >= 0.0f) ? 1.0f : -1.0f;
	float sy = (oy >= 0.0f) ? 1.0f : -1.0f;
	if (n.z < 0.0f) {
		float nx = (1.0f - fabsf(n.y)) * sx;
		float ny = (1.0f - fabsf(n.x)) * sy;
		n.x = nx; n.y = ny;
	}
	return normalize(n);
}

// the first half of get_shading_data (:723-752): vertex fetch and the barycentrics of the view ray
struct triangle_hit {
	f3 pos[3], nrm[3];
	f2 uv[3];
	f3 e0, e1, to0, ray_cross_e1, e0_cross_to0;
	float rcp_det, b0, b1, b2;
};

VKR_DEV triangle_hit intersect_primitive(const shade_params& p, uint32_t primitive, f3 ray_direction) {
	const uint8_t* c = p.constants;
	f3 factor = load_f3(c, 0), summand = load_f3(c, 16), camera = load_f3(c, 144);
	triangle_hit t;
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		size_t vi = (size_t) primitive * 3 + i;
		t.pos[i] = decode_position(p.positions[vi], factor, summand);
		uint2 q = p.normals_and_tex_coords[vi];
		t.nrm[i] = decode_normal(unorm16(q.x & 0xFFFF), unorm16(q.x >> 16));
		t.uv[i] = mk2(fmaf(unorm16(q.y & 0xFFFF), 8.0f, 0.0f), fmaf(unorm16(q.y >> 16), -8.0f, 1.0f));
	}
	t.e0 = t.pos[1] - t.pos[0]; t.e1 = t.pos[2] - t.pos[0];
	t.ray_cross_e1 = cross(ray_direction, t.e1);
	t.rcp_det = rcp(dot(t.e0, t.ray_cross_e1));
	t.to0 = camera - t.pos[0];
	t.b1 = t.rcp_det * dot(t.to0, t.ray_cross_e1);
	t.e0_cross_to0 = cross(t.e0, t.to0);
	t.b2 = -t.rcp_det * dot(ray_direction, t.e0_cross_to0);
	t.b0 = 1.0f - (t.b1 + t.b2);
	return t;
}

// ---- material textures: the software sampler (oracle_sample_texture in oracle/oracle_shading.c) ----

struct texture_view {
	const uint32_t* texels;  // RGBA8, all mip levels, finest first
	uint32_t width, height, mip_count;
	bool srgb;
};

VKR_DEV float4 fetch_texel(const shade_params& p, const texture_view& t, const uint32_t* level, int width, int height, int x, int y) {
	x = ((x % width) + width) % width;
	y = ((y % height) + height) % height;
	uint32_t texel = level[(size_t) y * (size_t) width + (size_t) x];
	uint32_t r = texel & 255u, g = (texel >> 8) & 255u, b = (texel >> 16) & 255u, a = texel >> 24;
	float4 out;
	out.x = t.srgb ? p.srgb_table[r] : (float) r * (1.0f / 255.0f);
	out.y = t.srgb ? p.srgb_table[g] : (float) g * (1.0f / 255.0f);
	out.z = t.srgb ? p.srgb_table[b] : (float) b * (1.0f / 255.0f);
	out.w = (float) a * (1.0f / 255.0f);
	return out;
}

VKR_DEV float4 sample_level(const shade_params& p, const texture_view& t, uint32_t level, float u, float v) {
	const uint32_t* texels = t.texels;
	int width = (int) t.width, height = (int) t.height;
	for (uint32_t l = 0; l != level; ++l) {
		texels += (size_t) width * (size_t) height;
		width = width > 1 ? width / 2 : 1;
		height = height > 1 ? height / 2 : 1;
	}
	float x = u * (float) width - 0.5f, y = v * (float) height - 0.5f;
	float x0 = floorf(x), y0 = floorf(y);
	float fx = x - x0, fy = y - y0;
	float4 t00 = fetch_texel(p, t, texels, width, height, (int) x0, (int) y0);
	float4 t10 = fetch_texel(p, t, texels, width, height, (int) x0 + 1, (int) y0);
	float4 t01 = fetch_texel(p, t, texels, width, height, (int) x0, (int) y0 + 1);
	float4 t11 = fetch_texel(p, t, texels, width, height, (int) x0 + 1, (int) y0 + 1);
	float4 out;
	out.x = (t00.x * (1.0f - fx) + t10.x * fx) * (1.0f - fy) + (t01.x * (1.0f - fx) + t11.x * fx) * fy;
	out.y = (t00.y * (1.0f - fx) + t10.y * fx) * (1.0f - fy) + (t01.y * (1.0f - fx) + t11.y * fx) * fy;
	out.z = (t00.z * (1.0f - fx) + t10.z * fx) * (1.0f - fy) + (t01.z * (1.0f - fx) + t11.z * fx) * fy;
	out.w = (t00.w * (1.0f - fx) + t10.w * fx) * (1.0f - fy) + (t01.w * (1.0f - fx) + t11.w * fx) * fy;
	return out;
}

// textureGrad with the sampler of src/scene.c:546-552 (linear filters, repeat addressing, 16x anisotropy), operation for
// operation oracle_sample_texture() of oracle/oracle_shading.c: what the Vulkan specification sketches as anisotropic
// filtering - N = min(ceil(P_max / P_min), 16, ceil(P_max)) trilinear taps at level log2(P_max / N), spread along the longer
// axis of the footprint at uv + (i / (N + 1) - 1 / 2) d(uv) and averaged in order; N = 1 is the isotropic trilinear sample of
// rounds 1 - 4 in every bit.  (The quotients may have any operands - a derivative of zero, a footprint of a thousand texels:
// the full-range division.)
constexpr float kMaxAnisotropy = 16.0f;
VKR_DEV float4 sample_texture(const shade_params& p, const texture_view& t, f2 uv, f2 duv_dx, f2 duv_dy) {
	float w = (float) t.width, h = (float) t.height;
	float ax = duv_dx.x * w, ay = duv_dx.y * h, bx = duv_dy.x * w, by = duv_dy.y * h;
	float px = square_root(ax * ax + ay * ay), py = square_root(bx * bx + by * by);
	bool x_major = px >= py;
	float p_max = gmax(px, py), p_min = x_major ? py : px;
	float taps = ceilf(divide_full_range(p_max, p_min));
	taps = gmin(gmin(taps, kMaxAnisotropy), gmax(ceilf(p_max), 1.0f));
	if (!(taps >= 1.0f)) taps = 1.0f;
	float rho = divide_full_range(p_max, taps);
	float max_level = (float) (t.mip_count - 1);
	float lambda = (rho > 1.0f) ? gmin(log2_poly(rho), max_level) : 0.0f;
	float level_0 = floorf(lambda);
	float fraction = lambda - level_0;
	uint32_t l0 = (uint32_t) level_0;
	uint32_t l1 = (l0 + 1 < t.mip_count) ? l0 + 1 : l0;
	uint32_t count = (uint32_t) taps;
	float du = x_major ? duv_dx.x : duv_dy.x, dv = x_major ? duv_dx.y : duv_dy.y;
	float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	for (uint32_t i = 0; i != count; ++i) {
		float u = uv.x, v = uv.y;
		if (count > 1) {
			float offset = divide_full_range((float) (i + 1), taps + 1.0f) - 0.5f;
			u = u + du * offset;
			v = v + dv * offset;
		}
		float4 c0 = sample_level(p, t, l0, u, v), c1 = sample_level(p, t, l1, u, v);
		float4 tap = make_float4(c0.x * (1.0f - fraction) + c1.x * fraction, c0.y * (1.0f - fraction) + c1.y * fraction,
			c0.z * (1.0f - fraction) + c1.z * fraction, c0.w * (1.0f - fraction) + c1.w * fraction);
		sum = (count > 1) ? make_float4(sum.x + tap.x, sum.y + tap.y, sum.z + tap.z, sum.w + tap.w) : tap;
	}
	if (count > 1) sum = make_float4(divide_full_range(sum.x, taps), divide_full_range(sum.y, taps), divide_full_range(sum.z, taps), divide_full_range(sum.w, taps));
	return sum;
}

// The texture reads of get_shading_data (:754-785) for one pixel: screen-space derivatives of the
// barycentrics and of the texture coordinate, then base colour, specular and normal texture.
// Writes the eight numbers that a constant material stores (see materials_t.host_constants).
VKR_DEV void resolve_material(const shade_params& p, uint32_t primitive, f3 ray_direction, float (&out)[8]) {
	triangle_hit t = intersect_primitive(p, primitive, ray_direction);
	const uint8_t* c = p.constants;
	f3 derivs[2];
#pragma unroll
	for (int i = 0; i < 2; ++i) {
		f3 ray_deriv = mk3(load_f(c, 96 + 4 * i), load_f(c, 112 + 4 * i), load_f(c, 128 + 4 * i));
		f3 ray_cross_e1_deriv = cross(ray_deriv, t.e1);
		float rcp_det_deriv = -dot(t.e0, ray_cross_e1_deriv) * t.rcp_det * t.rcp_det;
		float det_0_dir_e1 = dot(t.to0, t.ray_cross_e1);
		float det_0_dir_e1_deriv = dot(t.to0, ray_cross_e1_deriv);
		derivs[i].y = rcp_det_deriv * det_0_dir_e1 + t.rcp_det * det_0_dir_e1_deriv;
		float det_dir_e0_0 = dot(ray_direction, t.e0_cross_to0);
		float det_dir_e0_0_deriv = dot(ray_deriv, t.e0_cross_to0);
		derivs[i].z = -rcp_det_deriv * det_dir_e0_0 - t.rcp_det * det_dir_e0_0_deriv;
		derivs[i].x = -(derivs[i].y + derivs[i].z);
	}
	f2 tex_coord = fma2(t.b0, t.uv[0], fma2(t.b1, t.uv[1], t.uv[2] * t.b2));
	f2 tex_derivs[2];
#pragma unroll
	for (int i = 0; i < 2; ++i) {
		f2 sum = mk2(0.0f, 0.0f);
		sum = sum + t.uv[0] * derivs[i].x;
		sum = sum + t.uv[1] * derivs[i].y;
		sum = sum + t.uv[2] * derivs[i].z;
		tex_derivs[i] = sum;
	}
	uint32_t material = p.material_indices[primitive];
	const float* constants = p.material_constants + 8 * (size_t) material;
#pragma unroll
	for (int type = 0; type < 3; ++type) {
		const uint32_t* descriptor = p.texture_descriptors + 4 * (3 * (size_t) material + type);
		float4 texel = make_float4(constants[3 * type], constants[3 * type + 1], type < 2 ? constants[3 * type + 2] : 0.0f, 1.0f);
		if (descriptor[1] != 0) {
			texture_view view;
			view.texels = p.texels + descriptor[0];
			view.width = descriptor[1]; view.height = descriptor[2];
			view.mip_count = descriptor[3] & 0xFFFFu;
			view.srgb = (descriptor[3] >> 16) != 0;
			texel = sample_texture(p, view, tex_coord, tex_derivs[0], tex_derivs[1]);
		}
		out[3 * type] = texel.x;
		out[3 * type + 1] = texel.y;
		if (type < 2) out[3 * type + 2] = texel.z;
	}
}

VKR_DEV shading_data get_shading_data(const shade_params& p, uint32_t primitive, f3 ray_direction, size_t pixel_index) {
	const uint8_t* c = p.constants;
	f3 camera = load_f3(c, 144);
	shading_data r;
	triangle_hit t = intersect_primitive(p, primitive, ray_direction);
	const f3 (&pos)[3] = t.pos;
	const f3 (&nrm)[3] = t.nrm;
	const f2 (&uv)[3] = t.uv;
	f3 e0 = t.e0, e1 = t.e1;
	float b0 = t.b0, b1 = t.b1, b2 = t.b2;
	r.position = fma3(b0, pos[0], fma3(b1, pos[1], pos[2] * b2));
	f3 interpolated_normal = normalize(fma3(b0, nrm[0], fma3(b1, nrm[1], nrm[2] * b2)));
	// constant material, or what the material resolve kernel sampled for this pixel
	const float* mc = p.pixel_materials ? p.pixel_materials + 8 * pixel_index : p.material_constants + 8 * (size_t) p.material_indices[primitive];
	f3 base_color = mk3(mc[0], mc[1], mc[2]);
	float linear_roughness = mc[4], metalicity = mc[5];
	f3 nt;
	nt.x = fmaf(mc[6], 2.0f, -1.0f);
	nt.y = fmaf(mc[7], 2.0f, -1.0f);
	nt.z = square_root(gmax(0.0f, fmaf(-nt.x, nt.x, fmaf(-nt.y, nt.y, 1.0f))));
	r.diffuse_albedo = mk3(fmaf(base_color.x, -metalicity, base_color.x), fmaf(base_color.y, -metalicity, base_color.y), fmaf(base_color.z, -metalicity, base_color.z));
	float dielectric = 0.02f * (1.0f - metalicity);
	r.fresnel_0 = mk3(dielectric + base_color.x * metalicity, dielectric + base_color.y * metalicity, dielectric + base_color.z * metalicity);
	r.roughness = linear_roughness * linear_roughness;
	r.roughness = gclamp(r.roughness * load_f(c, 180), 0.0064f, 1.0f);
	f2 uv_e0 = uv[1] - uv[0], uv_e1 = uv[2] - uv[0];
	f3 n_cross_e0 = cross(interpolated_normal, e0);
	f3 e1_cross_n = cross(e1, interpolated_normal);
	f3 tangent = e1_cross_n * uv_e0.x + n_cross_e0 * uv_e1.x;
	f3 bitangent = e1_cross_n * uv_e0.y + n_cross_e0 * uv_e1.y;
	float mean_tangent_length = square_root(0.5f * (dot(tangent, tangent) + dot(bitangent, bitangent)));
	m3 t2w;
	t2w.c[0] = tangent; t2w.c[1] = bitangent; t2w.c[2] = interpolated_normal;
	nt.z *= gmax(1.0e-10f, mean_tangent_length);
	r.normal = normalize(mul(t2w, nt));
	r.outgoing = normalize(camera - r.position);
	float normal_offset = gmax(0.0f, 1.0e-3f - dot(r.normal, r.outgoing));
	r.normal = fma3(normal_offset, r.outgoing, r.normal);
	r.normal = normalize(r.normal);
	r.lambert_outgoing = dot(r.normal, r.outgoing);
	return r;
}

// ---- LTC (ltc_utility.glsl:58-108) ---------------------------------------------------

struct ltc_coefficients {
	m43 world_to_shading;
	m43 world_to_cosine;
	// shading_to_cosine = [[m00, 0, m02], [0, m11, 0], [m20, 0, m22]] (row, column)
	float m00, m02, m11, m20, m22;
	// cosine_to_shading, same sparsity
	float i00, i02, i11, i20, i22;
	float albedo, determinant;
};

VKR_DEV f3 shading_to_cosine(const ltc_coefficients& l, f3 v) {
	// column-major product with the structural zeros kept (they matter for -0)
	return mk3((l.m00 * v.x + 0.0f * v.y) + l.m02 * v.z, (0.0f * v.x + l.m11 * v.y) + 0.0f * v.z, (l.m20 * v.x + 0.0f * v.y) + l.m22 * v.z);
}
VKR_DEV f3 cosine_to_shading(const ltc_coefficients& l, f3 v) {
	return mk3((l.i00 * v.x + 0.0f * v.y) + l.i02 * v.z, (0.0f * v.x + l.i11 * v.y) + 0.0f * v.z, (l.i20 * v.x + 0.0f * v.y) + l.i22 * v.z);
}

VKR_DEV float bilinear(float t00, float t10, float t01, float t11, float wx, float wy) {
	float top = t00 * (1.0f - wx) + t10 * wx;
	float bottom = t01 * (1.0f - wx) + t11 * wx;
	return top * (1.0f - wy) + bottom * wy;
}

VKR_DEV ltc_coefficients get_ltc_coefficients(const shade_params& p, float fresnel_0, float roughness, f3 position, f3 normal, f3 outgoing) {
	const uint8_t* c = p.constants;
	ltc_coefficients l;
	float n_dot_o = dot(normal, outgoing);
	float inclination = arccos_unit(gclamp(n_dot_o, 0.0f, 1.0f));
	float u = fmaf(square_root(gclamp(roughness, 0.0f, 1.0f)), load_f(c, 232), load_f(c, 236));
	float v = fmaf(inclination, load_f(c, 240), load_f(c, 244));
	float w = fmaf(gclamp(fresnel_0, 0.0f, 1.0f), load_f(c, 224), load_f(c, 228));
	// bilinear, clamp to edge, nearest layer; exact fp32 weights, x first
	int res = (int) p.ltc_resolution;
	float fx = u * (float) res - 0.5f, fy = v * (float) res - 0.5f;
	float flx = floorf(fx), fly = floorf(fy);
	float wx = fx - flx, wy = fy - fly;
	int x0 = (int) flx, y0 = (int) fly;
	int x1 = min(max(x0 + 1, 0), res - 1), y1 = min(max(y0 + 1, 0), res - 1);
	x0 = min(max(x0, 0), res - 1);
	y0 = min(max(y0, 0), res - 1);
	int layer = min(max((int) rintf(w), 0), (int) p.ltc_layer_count - 1);
	size_t base = (size_t) layer * res * res;
	size_t i00 = base + (size_t) y0 * res + x0, i10 = base + (size_t) y0 * res + x1;
	size_t i01 = base + (size_t) y1 * res + x0, i11 = base + (size_t) y1 * res + x1;
	uint2 a00 = p.ltc_rgba[i00], a10 = p.ltc_rgba[i10], a01 = p.ltc_rgba[i01], a11 = p.ltc_rgba[i11];
	uint32_t b00 = p.ltc_rg[i00], b10 = p.ltc_rg[i10], b01 = p.ltc_rg[i01], b11 = p.ltc_rg[i11];
	float d0 = bilinear(unorm16(a00.x & 0xFFFF), unorm16(a10.x & 0xFFFF), unorm16(a01.x & 0xFFFF), unorm16(a11.x & 0xFFFF), wx, wy);
	float d1 = bilinear(unorm16(a00.x >> 16), unorm16(a10.x >> 16), unorm16(a01.x >> 16), unorm16(a11.x >> 16), wx, wy);
	float d2 = bilinear(unorm16(a00.y & 0xFFFF), unorm16(a10.y & 0xFFFF), unorm16(a01.y & 0xFFFF), unorm16(a11.y & 0xFFFF), wx, wy);
	float d3 = bilinear(unorm16(a00.y >> 16), unorm16(a10.y >> 16), unorm16(a01.y >> 16), unorm16(a11.y >> 16), wx, wy);
	float d4 = bilinear(unorm16(b00 & 0xFFFF), unorm16(b10 & 0xFFFF), unorm16(b01 & 0xFFFF), unorm16(b11 & 0xFFFF), wx, wy);
	float d5 = bilinear(unorm16(b00 >> 16), unorm16(b10 >> 16), unorm16(b01 >> 16), unorm16(b11 >> 16), wx, wy);
	// mat3(d0, 0, -d1,  0, d2, 0,  d3, 0, d4) column by column
	l.m00 = d0; l.m20 = -d1; l.m11 = d2; l.m02 = d3; l.m22 = d4;
	l.albedo = d5;
	float det2 = d0 * d4 + d1 * d3;
	l.determinant = d2 * det2;
	float inv_det2 = rcp(det2);
	l.i00 = d4 * inv_det2; l.i20 = d1 * inv_det2; l.i11 = rcp(d2); l.i02 = -d3 * inv_det2; l.i22 = d0 * inv_det2;
	f3 x_axis = normalize(fma3(-n_dot_o, normal, outgoing));
	f3 y_axis = cross(normal, x_axis);
	f3 r0 = mk3(x_axis.x, y_axis.x, normal.x);
	f3 r1 = mk3(x_axis.y, y_axis.y, normal.y);
	f3 r2 = mk3(x_axis.z, y_axis.z, normal.z);
	l.world_to_shading.c[0] = r0;
	l.world_to_shading.c[1] = r1;
	l.world_to_shading.c[2] = r2;
	m3 neg;
	neg.c[0] = -r0; neg.c[1] = -r1; neg.c[2] = -r2;
	l.world_to_shading.c[3] = mul(neg, position);
#pragma unroll
	for (int i = 0; i < 4; ++i) l.world_to_cosine.c[i] = shading_to_cosine(l, l.world_to_shading.c[i]);
	return l;
}

VKR_DEV float evaluate_ltc_density(const ltc_coefficients& l, f3 dir_shading, float rcp_psa) {
	f3 dc = shading_to_cosine(l, dir_shading);
	float len_sq = dot(dc, dc);
	float density = value_divide(gmax(0.0f, dc.z) * l.determinant, len_sq * len_sq);
	return density * rcp_psa;
}

// ---- BRDF (brdfs.glsl) -------------------------------------------------------------

VKR_DEV float schlick(float f0, float f90, float cos_theta) {
	float flipped = 1.0f - cos_theta;
	float flipped_squared = flipped * flipped;
	return f0 + (f90 - f0) * (flipped_squared * flipped * flipped_squared);
}

template <bool DIFFUSE, bool SPECULAR>
VKR_DEV f3 evaluate_brdf(const shading_data& d, f3 incoming) {
	f3 half_sum = incoming + d.outgoing;
	f3 half_vector = half_sum * value_rsqrt(dot(half_sum, half_sum));
	float lambert_incoming = dot(d.normal, incoming);
	float outgoing_dot_half = dot(d.outgoing, half_vector);
	f3 brdf = mk3(0.0f, 0.0f, 0.0f);
	if (DIFFUSE) {
		float f90 = fmaf(outgoing_dot_half * outgoing_dot_half, 2.0f * d.roughness, 0.5f);
		float product = schlick(1.0f, f90, d.lambert_outgoing) * schlick(1.0f, f90, lambert_incoming);
		brdf = brdf + d.diffuse_albedo * product;
	}
	if (SPECULAR) {
		float normal_dot_half = dot(d.normal, half_vector);
		float a2 = d.roughness * d.roughness;
		float ggx = fmaf(fmaf(normal_dot_half, a2, -normal_dot_half), normal_dot_half, 1.0f);
		ggx = value_divide(a2, ggx * ggx);
		float masking = lambert_incoming * value_square_root(fmaf(fmaf(-d.lambert_outgoing, a2, d.lambert_outgoing), d.lambert_outgoing, a2));
		float shadowing = d.lambert_outgoing * value_square_root(fmaf(fmaf(-lambert_incoming, a2, lambert_incoming), lambert_incoming, a2));
		float smith = value_divide(0.5f, masking + shadowing);
		float ct = gclamp(outgoing_dot_half, 0.0f, 1.0f);
		float gs = ggx * smith;
		brdf = brdf + mk3(gs * schlick(d.fresnel_0.x, 1.0f, ct), gs * schlick(d.fresnel_0.y, 1.0f, ct), gs * schlick(d.fresnel_0.z, 1.0f, ct));
	}
	return brdf * kInvPi;
}

VKR_DEV float ggx_vndf_density(float out_dot_n, float micro_dot_n, float micro_dot_out, float roughness) {
	float a2 = roughness * roughness;
	float ggx = fmaf(fmaf(micro_dot_n, a2, -micro_dot_n), micro_dot_n, 1.0f);
	ggx = divide(a2, ggx * ggx);
	ggx *= kInvPi;
	float masking = square_root(fmaf(fmaf(-out_dot_n, a2, out_dot_n), out_dot_n, a2));
	masking = divide(2.0f, out_dot_n + masking);
	return masking * micro_dot_out * ggx;
}

VKR_DEV f3 sample_ggx_vndf(f3 out_shading, float rx, float ry, f2 u) {
	m3 e2h;
	e2h.c[2] = normalize(mk3(rx * out_shading.x, ry * out_shading.y, 1.0f * out_shading.z));
	float length_sq = e2h.c[2].x * e2h.c[2].x + e2h.c[2].y * e2h.c[2].y;
	float inv_len = rsqrt(length_sq);
	e2h.c[0] = mk3(-e2h.c[2].y * inv_len, e2h.c[2].x * inv_len, 0.0f * inv_len);
	if (length_sq <= 0.0f) e2h.c[0] = mk3(1.0f, 0.0f, 0.0f);
	e2h.c[1] = cross(e2h.c[2], e2h.c[0]);
	float radius = square_root(u.x);
	float azimuth = (2.0f * kPi) * u.y;
	float sn, cs;
	sincos_poly(azimuth, sn, cs);
	f2 disk = mk2(radius * cs, radius * sn);
	f3 s;
	s.x = disk.x;
	float lerp = fmaf(0.5f, e2h.c[2].z, 0.5f);
	float a = square_root(fmaf(-disk.x, disk.x, 1.0f));
	s.y = a * (1.0f - lerp) + disk.y * lerp;
	s.z = square_root(gmax(0.0f, 1.0f - (s.x * s.x + s.y * s.y)));
	f3 hemi = mul(e2h, s);
	return normalize(mk3(rx * hemi.x, ry * hemi.y, 1.0f * hemi.z));
}

VKR_DEV f3 sample_ggx_reflected(float& out_density, f3 out_shading, float roughness, f2 u) {
	f3 micro = sample_ggx_vndf(out_shading, roughness, roughness, u);
	float micro_dot_out = dot(micro, out_shading);
	float density = ggx_vndf_density(out_shading.z, micro.z, micro_dot_out, roughness);
	f3 incoming = fma3(2.0f * micro_dot_out, micro, -out_shading);
	out_density = divide(density, 4.0f * micro_dot_out);
	return incoming;
}

VKR_DEV float ggx_reflected_density(float out_dot_n, f3 out_dir, f3 in_dir, f3 normal, float roughness) {
	f3 micro = normalize(out_dir + in_dir);
	float micro_dot_out = dot(micro, out_dir);
	float micro_dot_n = dot(micro, normal);
	float density = ggx_vndf_density(out_dot_n, micro_dot_n, micro_dot_out, roughness);
	return divide(density, 4.0f * micro_dot_out);
}

// ---- lights ----------------------------------------------------------------------

// Wave-uniform view of one packed light record (include/vkr_polygonal_light.h)
struct light_ref {
	const uint8_t* base;   // start of the 160-byte fixed part
	const uint8_t* world;  // world-space vertices, 16 bytes each
};

VKR_DEV light_ref get_light(const shade_params& p, uint32_t index) {
	uint32_t vmax = p.max_light_vertex_count;
	uint32_t stride = 160 + 16 * vmax * 2 + 16 * (vmax - 2);
	light_ref l;
	l.base = p.constants + 256 + stride * index;
	l.world = l.base + 160 + 16 * vmax;
	return l;
}
VKR_DEV f3 light_vertex(const light_ref& l, uint32_t i) { return load_f3(l.world, 16 * i); }
VKR_DEV uint32_t light_vertex_count(const light_ref& l) { return load_u(l.base, 80); }
VKR_DEV f3 light_radiance(const light_ref& l) { return load_f3(l.base, 48); }
VKR_DEV float plane_distance(const light_ref& l, f3 p) {
	return ((p.x * load_f(l.base, 64) + p.y * load_f(l.base, 68)) + p.z * load_f(l.base, 72)) + 1.0f * load_f(l.base, 76);
}
VKR_DEV f3 plane_normal(const light_ref& l) { return load_f3(l.base, 64); }
VKR_DEV f3 light_translation(const light_ref& l) { return load_f3(l.base, 16); }
// column k of the plane-to-world rotation (the uniform block is row_major)
VKR_DEV f3 light_rotation_column(const light_ref& l, uint32_t k) { return mk3(load_f(l.base, 96 + 4 * k), load_f(l.base, 112 + 4 * k), load_f(l.base, 128 + 4 * k)); }
VKR_DEV float light_area(const light_ref& l) { return load_f(l.base, 144); }
// (area of fan triangle i, area of the fan up to triangle i); the last entry holds the total
VKR_DEV f2 light_fan_area(const light_ref& l, uint32_t vmax, uint32_t i) { return mk2(load_f(l.world, 16 * vmax + 16 * i), load_f(l.world, 16 * vmax + 16 * i + 4)); }

// polygonal_light_ray_intersection, polygonal_light_utility.glsl:93-112
VKR_DEV bool light_ray_intersection(const light_ref& light, uint32_t vmax, f3 origin, f3 end_xyz, float end_w) {
	float side_a = plane_distance(light, origin);
	f3 n = plane_normal(light);
	float side_b = ((n.x * end_xyz.x + n.y * end_xyz.y) + n.z * end_xyz.z) + load_f(light.base, 76) * end_w;
	if (side_a * side_b > 0.0f) return false;
	f3 dir = end_xyz - origin * end_w;
	float previous_sign = 0.0f;
	bool result = true;
	uint32_t count = light_vertex_count(light);
	for (uint32_t i = 0; i != vmax; ++i) {
		f3 a = light_vertex(light, i) - origin;
		f3 b = light_vertex(light, (i + 1) % vmax) - origin;
		float sign = dir.x * (a.y * b.z - b.y * a.z) - a.x * (dir.y * b.z - b.y * dir.z) + b.x * (dir.y * a.z - a.y * dir.z);
		result = result && ((i >= 3 && i >= count) || previous_sign * sign >= 0.0f);
		previous_sign = sign;
	}
	return result;
}

// textureLod(g_light_textures[i], uv, 0.0f) with the sampler of reference main.c:611-621 (linear,
// u repeats, v clamps); same arithmetic as oracle_sample_light_texture
VKR_DEV f3 sample_light_texture(const shade_params& p, uint32_t texture_index, f2 uv) {
	uint4 d = p.light_texture_descriptors[texture_index];
	if (d.y == 0) return mk3(1.0f, 1.0f, 1.0f);
	float u = uv.x - floorf(uv.x), v = uv.y;
	if (!(u >= 0.0f && u <= 1.0f)) u = 0.0f;
	v = (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f;
	int w = (int) d.y, h = (int) d.z;
	float fx = u * (float) w - 0.5f, fy = v * (float) h - 0.5f;
	float flx = floorf(fx), fly = floorf(fy);
	float wx = fx - flx, wy = fy - fly;
	int ix = (int) flx, iy = (int) fly;
	int x0 = (ix < 0) ? w - 1 : ix, x1 = (ix + 1 >= w) ? 0 : ix + 1;
	int y0 = (iy < 0) ? 0 : iy, y1 = (iy + 1 >= h) ? h - 1 : iy + 1;
	const float4* texels = p.light_texels + d.x;
	float4 t00 = texels[(size_t) y0 * w + x0], t10 = texels[(size_t) y0 * w + x1];
	float4 t01 = texels[(size_t) y1 * w + x0], t11 = texels[(size_t) y1 * w + x1];
	float one_minus_wx = 1.0f - wx, one_minus_wy = 1.0f - wy;
	f3 top = mk3(t00.x * one_minus_wx + t10.x * wx, t00.y * one_minus_wx + t10.y * wx, t00.z * one_minus_wx + t10.z * wx);
	f3 bottom = mk3(t01.x * one_minus_wx + t11.x * wx, t01.y * one_minus_wx + t11.y * wx, t01.z * one_minus_wx + t11.z * wx);
	return mk3(top.x * one_minus_wy + bottom.x * wy, top.y * one_minus_wy + bottom.y * wy, top.z * one_minus_wy + bottom.z * wy);
}

// get_polygon_radiance, shading_pass.frag.glsl:151-185.  TEXTURED is a property of the kernel
// variant (the host picks it when any light has a texturing technique): the extra live registers
// would otherwise cost the untextured variants a wave of occupancy.
template <bool TEXTURED>
VKR_DEV f3 polygon_radiance(const shade_params& p, f3 dir, f3 position, const light_ref& light) {
	f3 radiance = light_radiance(light);
	if constexpr (!TEXTURED) return radiance;
	uint32_t technique = load_u(light.base, 84);
	if (technique != 0 && p.light_texture_descriptors) {
		f2 uv;
		if (technique == 1) {
			// polygon_texturing_area: plane-space coordinates of the point that the ray hits
			float t = divide(-plane_distance(light, position), dot(dir, plane_normal(light)));
			f3 x = (position + dir * t) - light_translation(light);
			uv.x = dot(light_rotation_column(light, 0), x) * load_f(light.base, 44);
			uv.y = dot(light_rotation_column(light, 1), x) * load_f(light.base, 60);
		}
		else {
			f3 lookup;
			if (technique == 3) {
				// polygon_texturing_ies_profile: plane space; the profile contains the cosine
				lookup = mk3(dot(light_rotation_column(light, 0), dir), dot(light_rotation_column(light, 1), dir), dot(light_rotation_column(light, 2), dir));
				radiance = radiance * divide(1.0f, fabsf(lookup.z));
			}
			else
				// polygon_texturing_portal: light probe parametrisation
				lookup = mk3(-dir.x, dir.y, dir.z);
			uv.x = arctan2(lookup.y, lookup.x) * (0.5f * kInvPi);
			uv.y = arccos(lookup.z) * kInvPi;
		}
		radiance = radiance * sample_light_texture(p, load_u(light.base, 88), uv);
	}
	return radiance;
}

struct pixel_context {
	const shade_params& p;
	uint32_t rays;
	// deferred mode: this thread's slot in the term streams and its write cursors
	uint32_t tid, code_cursor, term_cursor;
	bool light_has_terms;
	// nothing can block a ray toward the current light that stays inside its shaft (shade_params::shaft_clear;
	// wave-uniform), and the number of that light
	bool light_clear;
	uint32_t light_index;
	// ... or only this pair's list of triangles can (list_count of them, wave-uniform; 0: nothing can)
	uint32_t list_count;
	const float* list;
	uint32_t queue;
	// deferred mode, clear lights: as long as no term of the light is in the stream (`leading`), the terms that need no
	// ray are added up as they come - the resolve kernel would add them one after the other to its sum of the light, which
	// is still +0, and 0 + ((0 + t1) + t2) is the same bits as (0 + t1) + t2 - and go to the stream as ONE final term when
	// a term with a ray comes up or the light ends: a light whose shaft is clear sends one term per pixel to the resolve
	// kernel instead of 2 S.  (After the first term with a ray the sum is no longer +0, and (s + t1) + t2 is not
	// s + (t1 + t2): later final terms are written one by one.)
	f3 final_sum;
	uint32_t final_state;  // bit 0: final_sum holds terms, bit 1: leading (one word: two bools ended up in scratch memory)
	// this thread's column of the LDS tables of the prepared polygons (strategies with two
	// techniques per light: 2 x kPsaTableSlots(V) slots, [slot][thread]), else NULL
	float2* psa_tables;
	// ... and of the one table that lives in device memory when LDS has room for one only (psa_table_in_memory), else NULL
	float2* psa_table_memory;
	// the pixel's noise accessor (accumulate() settles its outstanding request before it stores), or NULL
	noise_accessor* noise;
};

// get_polygon_radiance_visibility_brdf_product, shading_pass.frag.glsl:203-231, without
// the ray query: `candidate` is the visibility before tracing (n.l > 0), the return
// value is radiance * BRDF under the assumption that the shadow ray reaches the light.
template <bool DIFFUSE, bool SPECULAR, bool TEXTURED>
VKR_DEV f3 radiance_brdf(const shade_params& p, float& out_lambert, bool& out_candidate, f3 dir, const shading_data& sd, const light_ref& light) {
	float lambert = dot(sd.normal, dir);
	out_lambert = lambert;
	out_candidate = lambert > 0.0f;
	if (out_candidate) return polygon_radiance<TEXTURED>(p, dir, sd.position, light) * evaluate_brdf<DIFFUSE, SPECULAR>(sd, dir);
	return mk3(0.0f, 0.0f, 0.0f);
}

VKR_DEV bool all_zero(f3 v) { return ((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z)) & 0x7FFFFFFFu) == 0; }

// The block of queue slots that a wave has reserved and not used up yet: (first free slot,
// slots left, real rays written so far, unused).  Wave-uniform state that lanes update while
// the wave is diverged, hence in LDS (one entry per wave of the workgroup) and volatile.
// (As an LDS pointer, not a generic one: the compiler does not infer the address space of volatile
// accesses, and FLAT loads return through the vector-memory counter IN ORDER - every push_ray() then waited
// for the global stores of the term before it, a round trip to the L2 per ray.)
typedef __attribute__((address_space(3))) volatile uint32_t lds_state_word;
VKR_DEV lds_state_word* ray_block_state() {
	// (a shading workgroup is one wave, kShadeThreads: one entry of four words)
	__shared__ uint32_t state[4];
	return (lds_state_word*) state;
}

// Appends one shadow ray to the queue of this wave.  Lanes of the wave that arrive here
// together take their slots with one ballot -> popcount -> lane prefix.  Without BLOCKS every
// push reserves exactly its slots with an atomic, whose round trip stalls the wave; a wave that
// queues many rays (config 3: 14 per lane) reserves p.ray_block slots at a time and hands them
// out from LDS, one atomic per block.  (A kernel variant, not a run-time switch: the block
// bookkeeping costs 7 VGPRs, which is a wave of occupancy for config 2's kernel.)
template <bool BLOCKS>
VKR_DEV void push_ray(const shade_params& p, uint32_t queue, f3 dir, float t_max, uint32_t record) {
	uint64_t mask = __ballot(1);
	uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
	uint32_t count = (uint32_t) __popcll(mask);
	uint32_t slot_in_queue;
	if constexpr (!BLOCKS) {
		uint32_t base = 0;
		if (prefix == 0) base = atomicAdd(p.ray_queue_size + queue, count);
		slot_in_queue = __builtin_amdgcn_readfirstlane(base) + prefix;
	}
	else {
		lds_state_word* state = ray_block_state();
		uint32_t old_base = __builtin_amdgcn_readfirstlane(state[0]), left = __builtin_amdgcn_readfirstlane(state[1]);
		uint32_t new_base = 0;
		if (count > left) {
			// the rest of the old block is used up first, then a new block is opened
			if (prefix == 0) new_base = atomicAdd(p.ray_queue_size + queue, p.ray_block);
			new_base = __builtin_amdgcn_readfirstlane(new_base);
		}
		slot_in_queue = (prefix < left) ? old_base + prefix : new_base + (prefix - left);
		if (prefix == 0) {
			state[0] = (count > left) ? new_base + (count - left) : old_base + count;
			state[1] = (count > left) ? p.ray_block - (count - left) : left - count;
			state[2] = state[2] + count;
		}
	}
	size_t slot = (size_t) queue * p.ray_queue_capacity + slot_in_queue;
	p.ray_directions[slot] = make_float4(dir.x, dir.y, dir.z, t_max);
	p.ray_records[slot] = record;
}

// At the end of a shading wave: slots of its last block that no ray took become null rays (the
// tracing kernel skips them), and the wave's ray count goes to its XCD's counter.
VKR_DEV void close_ray_block(const shade_params& p, uint32_t queue) {
	lds_state_word* state = ray_block_state();
	uint64_t mask = __ballot(1);
	uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
	uint32_t lanes = (uint32_t) __popcll(mask);
	uint32_t base = __builtin_amdgcn_readfirstlane(state[0]), left = __builtin_amdgcn_readfirstlane(state[1]), rays = __builtin_amdgcn_readfirstlane(state[2]);
	for (uint32_t i = rank; i < left; i += lanes) {
		size_t slot = (size_t) queue * p.ray_queue_capacity + base + i;
		p.ray_directions[slot] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
		p.ray_records[slot] = kNullRay;
	}
	if (rank == 0 && rays) atomicAdd(p.ray_queue_size + kRayCountOffset + (queue >> 6) * kCursorStride, rays);
}

// Light shafts (light_shafts.h) are built around a rectangle in the light's plane that contains the polygon with room to
// spare.  Samples aim at the polygon, but where the sampling breaks down numerically (a polygon that is a sliver in the
// space it is sampled in) the reference shader's directions leave it - and the ray query toward the light's PLANE is
// still part of the result.  So a ray only skips the tracing if it meets the plane inside that rectangle, in front of
// the origin: u_min <= u <= u_max with u = col0 . (o + t d - T) / s_x, t = a / b, a = -(n . o + w), b = n . d, written
// without the division (both sides times b).  NaNs fail every comparison: such a ray is queued as before.
VKR_DEV bool shaft_holds_ray(const shade_params& p, uint32_t light_index, const light_ref& light, f3 origin, f3 dir) {
	constant_float_pointer words = (constant_float_pointer) (uintptr_t) (p.shaft_rectangles + light_index);
	float4 rectangle = make_float4(words[0], words[1], words[2], words[3]);
	f3 n = plane_normal(light);
	float a = -plane_distance(light, origin), b = dot(dir, n);
	f3 relative = origin - light_translation(light);
	f3 column_u = light_rotation_column(light, 0), column_v = light_rotation_column(light, 1);
	// plane-space coordinates of the hit point, times b
	float u = (dot(column_u, relative) * b + a * dot(column_u, dir)) * load_f(light.base, 44);
	float v = (dot(column_v, relative) * b + a * dot(column_v, dir)) * load_f(light.base, 60);
	bool forward = b > 0.0f;
	float lo_u = forward ? rectangle.x * b : rectangle.z * b, hi_u = forward ? rectangle.z * b : rectangle.x * b;
	float lo_v = forward ? rectangle.y * b : rectangle.w * b, hi_v = forward ? rectangle.w * b : rectangle.y * b;
	// t > 0: a and b of one sign
	return a * b > 0.0f && u >= lo_u && u <= hi_u && v >= lo_v && v <= hi_v;
}

// Adds one estimator term to the per-light sum.  `visible_term` is the value of the
// term if the shadow ray reaches the light (or, without a candidate ray, simply the
// value), `hidden_term` the value if it is blocked.
// writes the sum of the final terms seen so far as one term of the stream
VKR_DEV void flush_final_sum(pixel_context& ctx) {
	const shade_params& p = ctx.p;
	if (ctx.term_cursor < p.max_terms && ctx.code_cursor + 2 < p.max_codes) {
		if (ctx.noise) settle_noise(*ctx.noise);
		size_t term_index = ((size_t) ctx.term_cursor * p.thread_count + ctx.tid) * 3;
		p.codes[code_slot(p.thread_count, ctx.code_cursor, ctx.tid)] = (uint8_t) kCodeFinal;
		p.terms_visible[term_index] = ctx.final_sum.x; p.terms_visible[term_index + 1] = ctx.final_sum.y; p.terms_visible[term_index + 2] = ctx.final_sum.z;
		++ctx.code_cursor;
		++ctx.term_cursor;
		ctx.light_has_terms = true;
	}
	ctx.final_sum = mk3(0.0f, 0.0f, 0.0f);
	ctx.final_state &= ~1u;
}

template <int RAYS>
VKR_DEV void accumulate(pixel_context& ctx, f3& result, bool candidate, f3 visible_term, f3 hidden_term, f3 dir, const shading_data& sd, const light_ref& light) {
	if constexpr (RAYS == kRaysNone) {
		result = result + visible_term;
	}
	else if constexpr (RAYS == kRaysInline) {
		bool visible = candidate;
		if (candidate) {
			float max_t = divide(-plane_distance(light, sd.position), dot(dir, plane_normal(light)));
			++ctx.rays;
			visible = !any_hit(ctx.p.bvh, sd.position, dir, 1.0e-3f, max_t);
		}
		result = result + ((visible || !candidate) ? visible_term : hidden_term);
	}
	else {
		// Adding +-0 never changes the running sum (it starts at +0), so such terms are
		// dropped; everything else is written in program order.
		const shade_params& p = ctx.p;
		// (a ray inside a shaft that is clear would arrive: its term is the visible one, right away; a ray inside a shaft
		// with an occluder list meets those triangles or none: decided here, by the tracing kernel's own triangle test)
		bool hidden_matters = !all_zero(hidden_term);
		bool arrives = false, blocked = false;
		if (ctx.light_clear && candidate && shaft_holds_ray(p, ctx.light_index, light, sd.position, dir)) {
			if (ctx.list_count != 0u) {
				float max_t = divide(-plane_distance(light, sd.position), dot(dir, plane_normal(light)));
				for (uint32_t j = 0; j != ctx.list_count; ++j) {
					constant_float_pointer t = (constant_float_pointer) (uintptr_t) (ctx.list + kShaftListEntry * j);
					float dist;
					blocked = blocked || ray_triangle_edges<false>(mk3(t[0], t[1], t[2]), mk3(t[4], t[5], t[6]), mk3(t[8], t[9], t[10]), sd.position, dir, 1.0e-3f, max_t, dist);
				}
			}
			arrives = !blocked;
		}
		// what a blocked ray leaves of its term: nothing, the hidden value, or - without a buffer for hidden values - NaN
		// (kCodePendingHiddenNaN above)
		f3 value = visible_term;
		if (blocked) value = !hidden_matters ? mk3(0.0f, 0.0f, 0.0f) : (p.terms_hidden ? hidden_term : mk3(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")));
		bool needs_ray = candidate && !arrives && !blocked && (hidden_matters || !all_zero(visible_term));
		bool is_final = (!candidate || arrives || blocked) && !all_zero(value);
#if VKR_SUM_FINAL_TERMS
		if (ctx.light_clear && (ctx.final_state & 2u)) {
			if (is_final) {
#if VKR_FAST_MATH
				value = mk3(opaque(value.x), opaque(value.y), opaque(value.z));
#endif
				ctx.final_sum = ctx.final_sum + value;
				ctx.final_state |= 1u;
				return;
			}
			if (needs_ray) {
				if (ctx.final_state & 1u) flush_final_sum(ctx);
				ctx.final_state = 0u;
			}
		}
#endif
		if ((needs_ray || is_final) && ctx.term_cursor < p.max_terms && ctx.code_cursor + 2 < p.max_codes) {
			if (ctx.noise) settle_noise(*ctx.noise);
			size_t code_index = code_slot(p.thread_count, ctx.code_cursor, ctx.tid);
			size_t term_index = ((size_t) ctx.term_cursor * p.thread_count + ctx.tid) * 3;
			// (terms_hidden exists for the one estimator whose blocked terms have values, the optimal heuristic)
			p.codes[code_index] = (uint8_t) (is_final ? kCodeFinal : (hidden_matters ? (p.terms_hidden ? kCodePendingWithHidden : kCodePendingHiddenNaN) : kCodePending));
			p.terms_visible[term_index] = value.x; p.terms_visible[term_index + 1] = value.y; p.terms_visible[term_index + 2] = value.z;
			if (needs_ray && hidden_matters && p.terms_hidden) {
				p.terms_hidden[term_index] = hidden_term.x; p.terms_hidden[term_index + 1] = hidden_term.y; p.terms_hidden[term_index + 2] = hidden_term.z;
			}
			if (needs_ray) {
				float max_t = divide(-plane_distance(light, sd.position), dot(dir, plane_normal(light)));
				push_ray<RAYS == kRaysDeferredBlocks>(p, ctx.queue, dir, max_t, ray_record(p.ray_thread_bits, ctx.tid, ctx.code_cursor));
			}
			++ctx.code_cursor;
			++ctx.term_cursor;
			ctx.light_has_terms = true;
		}
	}
}

VKR_DEV float mis_weight_over_density(int heuristic, float sampled, float other) {
	if (heuristic == kMisBalance) return value_rcp(sampled + other);
	if (heuristic == kMisPower) return value_divide(sampled, sampled * sampled + other * other);
	return 0.0f;
}

VKR_DEV float mis_estimate_channel(int heuristic, float in, float s, float sd, float o, float od, float ve) {
	if (heuristic == kMisWeighted) {
		float weighted_sum = s * sd + o * od;
		return value_divide(s * in, weighted_sum);
	}
	if (heuristic == kMisOptimalClamped || heuristic == kMisOptimal) {
		float balance = value_rcp(sd + od);
		float weighted_sum = s * sd + o * od;
		if (heuristic == kMisOptimalClamped) {
			float weighted = value_divide(s, weighted_sum);
			float mixed = fmaf(-ve, balance, balance);
			mixed = fmaf(ve, weighted, mixed);
			return mixed * in;
		}
		return ve * s + balance * (in - ve * weighted_sum);
	}
	return mis_weight_over_density(heuristic, sd, od) * in;
}

VKR_DEV f3 mis_estimate(int heuristic, f3 integrand, f3 sw, float sd, f3 ow, float od, float ve) {
	return mk3(mis_estimate_channel(heuristic, integrand.x, sw.x, sd, ow.x, od, ve),
		mis_estimate_channel(heuristic, integrand.y, sw.y, sd, ow.y, od, ve),
		mis_estimate_channel(heuristic, integrand.z, sw.z, sd, ow.z, od, ve));
}

// get_mis_estimate (shading_pass.frag.glsl:270-293) for the two integrands a deferred term needs - its
// value if the shadow ray reaches the light and its value if not - in one go: everything that does not
// depend on the integrand (the balance weight 1 / (sd + od), per channel the weighted sum and its
// quotient) is evaluated once instead of once per integrand and channel.  The operations on each value
// are those of mis_estimate_channel, so the results are the same bits; what it saves is issue slots -
// per term 4 instead of 12 divisions with the clamped optimal heuristic of BASELINE configs 3 and 4.
VKR_DEV void mis_estimate_pair(int heuristic, f3 lit, f3 dark, f3 sw, float sd, f3 ow, float od, float ve, f3& out_lit, f3& out_dark) {
	const float s[3] = {sw.x, sw.y, sw.z}, o[3] = {ow.x, ow.y, ow.z};
	const float a[3] = {lit.x, lit.y, lit.z}, b[3] = {dark.x, dark.y, dark.z};
	float ra[3], rb[3];
	if (heuristic == kMisWeighted) {
#pragma unroll
		for (int c = 0; c != 3; ++c) {
			float weighted_sum = s[c] * sd + o[c] * od;
			ra[c] = value_divide(s[c] * a[c], weighted_sum);
			rb[c] = value_divide(s[c] * b[c], weighted_sum);
		}
	}
	else if (heuristic == kMisOptimalClamped || heuristic == kMisOptimal) {
		float balance = value_rcp(sd + od);
#pragma unroll
		for (int c = 0; c != 3; ++c) {
			float weighted_sum = s[c] * sd + o[c] * od;
			if (heuristic == kMisOptimalClamped) {
				float weighted = value_divide(s[c], weighted_sum);
				float mixed = fmaf(-ve, balance, balance);
				mixed = fmaf(ve, weighted, mixed);
				ra[c] = mixed * a[c];
				rb[c] = mixed * b[c];
			}
			else {
				ra[c] = ve * s[c] + balance * (a[c] - ve * weighted_sum);
				rb[c] = ve * s[c] + balance * (b[c] - ve * weighted_sum);
			}
		}
	}
	else {
		float weight = mis_weight_over_density(heuristic, sd, od);
#pragma unroll
		for (int c = 0; c != 3; ++c) {
			ra[c] = weight * a[c];
			rb[c] = weight * b[c];
		}
	}
	out_lit = mk3(ra[0], ra[1], ra[2]);
	out_dark = mk3(rb[0], rb[1], rb[2]);
}

// get_polygonal_light_mis_estimate, shading_pass.frag.glsl:305-323
template <int STRATEGY, int RAYS, bool TEXTURED>
VKR_DEV void add_light_mis_estimate(pixel_context& ctx, f3& result, f3 dir, float density, const shading_data& sd, const light_ref& light) {
	// (once per sample and on every path through it, so that no request is outstanding at the top of the loop)
	if (ctx.noise) settle_noise(*ctx.noise);
	float lambert;
	bool candidate;
	f3 rb = radiance_brdf<true, true, TEXTURED>(ctx.p, lambert, candidate, dir, sd, light);
	const f3 zero = mk3(0.0f, 0.0f, 0.0f);
	f3 visible_term = zero, hidden_term = zero;
	if (STRATEGY == kStrategyDiffuseOnly) {
		float factor = divide(lambert, density);
		visible_term = (density > 0.0f) ? rb * factor : zero;
		hidden_term = (density > 0.0f) ? zero * factor : zero;
	}
	else if (STRATEGY == kStrategyDiffuseGgxMis) {
		float ggx_density = ggx_reflected_density(sd.lambert_outgoing, sd.outgoing, dir, sd.normal, sd.roughness);
		float weight = mis_weight_over_density(ctx.p.mis_heuristic, density, ggx_density);
		visible_term = (rb * lambert) * weight;
		hidden_term = (zero * lambert) * weight;
	}
	accumulate<RAYS>(ctx, result, candidate, visible_term, hidden_term, dir, sd, light);
}

// kLightTextures is not an error display: it shares the template slot because the two never
// combine usefully (an error display returns before any radiance is looked up)
enum { kErrorNone = 0, kErrorDiffuse = 1, kErrorSpecular = 2, kLightTextures = 3 };

// error_to_color, shading_pass.frag.glsl:80-114: tab20b colours (linear Rec. 709), four
// shades per decade over five decades
__device__ const float k_tab20b[20][3] = {
	{0.04092f, 0.04374f, 0.19120f}, {0.08438f, 0.08866f, 0.36625f}, {0.14703f, 0.15593f, 0.62396f}, {0.33245f, 0.34191f, 0.73046f},
	{0.12477f, 0.19120f, 0.04092f}, {0.26225f, 0.36131f, 0.08438f}, {0.46208f, 0.62396f, 0.14703f}, {0.61721f, 0.70838f, 0.33245f},
	{0.26225f, 0.15293f, 0.03071f}, {0.50888f, 0.34191f, 0.04092f}, {0.79910f, 0.49102f, 0.08438f}, {0.79910f, 0.59720f, 0.29614f},
	{0.23074f, 0.04519f, 0.04092f}, {0.41789f, 0.06663f, 0.06848f}, {0.67244f, 0.11954f, 0.14703f}, {0.79910f, 0.30499f, 0.33245f},
	{0.19807f, 0.05286f, 0.17144f}, {0.37626f, 0.08228f, 0.29614f}, {0.61721f, 0.15293f, 0.50888f}, {0.73046f, 0.34191f, 0.67244f}};

VKR_DEV f3 error_to_color(const shade_params& p, float error) {
	float error_factor = load_f(p.constants, 28);
	error = gmin(gmax(fabsf(error_factor * error), 1.0f), p.error_max);
	// NaN (degenerate sample) is undefined in the reference (int(NaN)); defined as the first colour
	error = (error == error) ? error : 1.0f;
	float color_index = fmaf(log2_poly(error), p.error_scale, -0.0f);
	int index = (int) color_index;
	return mk3(k_tab20b[index][0], k_tab20b[index][1], k_tab20b[index][2]);
}

// the ERROR_DISPLAY_* branches, shading_pass.frag.glsl:489-494, :549-563
template <int V, bool BIASED>
VKR_DEV f3 display_sampling_error(const shade_params& p, const psa_polygon<V>& polygon, noise_accessor& noise) {
	f2 u = next_noise_2(p, noise);
	f3 dir = sample_psa<V, BIASED>(polygon, u);
	f3 e = psa_sampling_error<V>(polygon, u, dir);
	float error = (p.error_index == 0) ? e.x : ((p.error_index == 1) ? e.y : e.z);
	f3 color = error_to_color(p, error);
	float exposure = load_f(p.constants, 176);
	return mk3(divide(color.x, exposure), divide(color.y, exposure), divide(color.z, exposure));
}

// evaluate_polygonal_light_shading, shading_pass.frag.glsl:329-711
template <int STRATEGY, int TECHNIQUE, int V, int RAYS, int ERROR = kErrorNone>
VKR_DEV f3 evaluate_light(pixel_context& ctx, const shading_data& sd, const ltc_coefficients& ltc_in, const light_ref& light, noise_accessor& noise) {
	const shade_params& p = ctx.p;
	constexpr bool kTextured = ERROR == kLightTextures;
	constexpr bool kBiased = TECHNIQUE == kTechniquePsaBiased;
	constexpr bool kIsPsa = TECHNIQUE == kTechniquePsa || TECHNIQUE == kTechniquePsaBiased;
	// (no request is outstanding when a sampling loop is entered - the first one was made when the pixel
	// started - or the wait for it would sit at the top of the loop and be paid by every sample)
	settle_noise(noise);
	const uint32_t S = p.sample_count;
	const uint32_t count = light_vertex_count(light);
	const f3 zero = mk3(0.0f, 0.0f, 0.0f);
	f3 result = zero;
	float density_factor = 0.0f;
	m43 world_to_shading = ltc_in.world_to_shading;

	if constexpr (TECHNIQUE == kTechniqueBaseline) {
		// shading_pass.frag.glsl:332-342: not a sampler, the run time baseline of the paper
		f3 corner_offset = light_translation(light) - sd.position;
		f3 r0 = light_rotation_column(light, 0), r1 = light_rotation_column(light, 1);
		for (uint32_t s = 0; s != S; ++s) {
			f2 u = next_noise_2(p, noise);
			f3 dir = normalize((corner_offset + r0 * u.x) + r1 * u.y);
			add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, 1.0f, sd, light);
		}
	}
	else if constexpr (TECHNIQUE == kTechniqueAreaTurk) {
		// uniform area sampling (Turk), polygon_sampling_related_work.glsl:38-85
		const uint32_t vmax = p.max_light_vertex_count;
		for (uint32_t s = 0; s != S; ++s) {
			f2 u = next_noise_2(p, noise);
			float target_area = light_fan_area(light, vmax, vmax - 3).y * u.x;
			float subtriangle_area = target_area;
			float triangle_area = light_fan_area(light, vmax, 0).x;
			f3 t0 = light_vertex(light, 1), t1 = light_vertex(light, 0), t2 = light_vertex(light, 2);
			bool done = false;
			for (uint32_t i = 0; i + 3 < vmax; ++i) {
				f2 fan = light_fan_area(light, vmax, i);
				done = done || i + 3 >= count || fan.y >= target_area;
				if (!done) {
					subtriangle_area = target_area - fan.y;
					triangle_area = light_fan_area(light, vmax, i + 1).x;
					t0 = light_vertex(light, i + 2);
					t2 = light_vertex(light, i + 3);
				}
			}
			float sqrt_u = square_root(divide(subtriangle_area, triangle_area));
			float b0 = 1.0f - sqrt_u, b1 = sqrt_u * u.y, b2 = fmaf(-sqrt_u, u.y, sqrt_u);
			f3 light_sample = (t0 * b0 + t1 * b1) + t2 * b2;
			f3 d = light_sample - sd.position;
			float distance_squared = dot(d, d);
			f3 dir = d * rsqrt(distance_squared);
			float projected_area = fabsf(dot(plane_normal(light), dir)) * light_area(light);
			add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, divide(distance_squared, projected_area), sd, light);
		}
	}
	else if constexpr (TECHNIQUE == kTechniqueUrena) {
		// shading_pass.frag.glsl:352-362: the light is taken to be the unit square of its plane
		urena_rectangle pd = prepare_urena(light_translation(light), load_f(light.base, 12), load_f(light.base, 28),
			light_rotation_column(light, 0), light_rotation_column(light, 1), light_rotation_column(light, 2), sd.position);
		density_factor = rcp(pd.solid_angle);
		for (uint32_t s = 0; s != S; ++s) {
			f3 dir = sample_urena(pd, next_noise_2(p, noise));
			add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density_factor, sd, light);
		}
	}
	else if constexpr (TECHNIQUE == kTechniqueArvoSolidAngle) {
		// :364-373
		f3 vw[V];
#pragma unroll
		for (int i = 0; i < V; ++i) vw[i] = light_vertex(light, min((uint32_t) i, p.max_light_vertex_count - 1));
		arvo_polygon<V> pd;
		prepare_arvo<V>(pd, count, vw, sd.position);
		density_factor = rcp(pd.solid_angle);
		for (uint32_t s = 0; s != S; ++s) {
			f3 dir = sample_arvo<V>(pd, next_noise_2(p, noise));
			add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density_factor, sd, light);
		}
	}
	else if constexpr (TECHNIQUE == kTechniqueHartBilinear || TECHNIQUE == kTechniqueHartBilinearClipping
		|| TECHNIQUE == kTechniqueHartBiquadratic || TECHNIQUE == kTechniqueHartBiquadraticClipping)
	{
		// :386-438; without clipping the polygon keeps MAX_POLYGONAL_LIGHT_VERTEX_COUNT slots
		constexpr bool kClipping = TECHNIQUE == kTechniqueHartBilinearClipping || TECHNIQUE == kTechniqueHartBiquadraticClipping;
		constexpr bool kBiquadratic = TECHNIQUE == kTechniqueHartBiquadratic || TECHNIQUE == kTechniqueHartBiquadraticClipping;
		constexpr int kLightSlots = kClipping ? V - 1 : V;
		f3 vs[V];
#pragma unroll
		for (int i = 0; i < kLightSlots; ++i) vs[i] = mul_point(world_to_shading, light_vertex(light, min((uint32_t) i, p.max_light_vertex_count - 1)));
		if constexpr (kClipping) vs[V - 1] = zero;
		uint32_t clipped = count;
		if constexpr (kClipping) {
			clipped = clip_polygon<V>(count, vs);
			if (clipped == 0) return zero;
		}
		if constexpr (kBiquadratic) {
			hart_biquadratic<V> pd;
			prepare_hart_biquadratic<V>(pd, clipped, vs);
			for (uint32_t s = 0; s != S; ++s) {
				float density;
				f3 dir = sample_hart_biquadratic<V>(density, pd, next_noise_2(p, noise));
				dir = mul_transposed(world_to_shading, dir);
				add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density, sd, light);
			}
		}
		else {
			hart_bilinear<V> pd;
			prepare_hart_bilinear<V>(pd, clipped, vs);
			for (uint32_t s = 0; s != S; ++s) {
				float density;
				f3 dir = sample_hart_bilinear<V>(density, pd, next_noise_2(p, noise));
				dir = mul_transposed(world_to_shading, dir);
				add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density, sd, light);
			}
		}
	}
	else if constexpr (TECHNIQUE == kTechniqueSolidAngle) {
		f3 vw[V];
#pragma unroll
		for (int i = 0; i < V; ++i) vw[i] = light_vertex(light, min((uint32_t) i, p.max_light_vertex_count - 1));
		sa_polygon<V> pd;
		prepare_sa<V>(pd, count, vw, sd.position);
		density_factor = rcp(pd.solid_angle);
		for (uint32_t s = 0; s != S; ++s) {
			f3 dir = sample_sa<V>(pd, next_noise_2(p, noise));
			add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density_factor, sd, light);
		}
	}
	else if constexpr (TECHNIQUE == kTechniqueClippedSolidAngle) {
		f3 vs[V];
#pragma unroll
		for (int i = 0; i < V - 1; ++i) vs[i] = mul_point(world_to_shading, light_vertex(light, min((uint32_t) i, p.max_light_vertex_count - 1)));
		vs[V - 1] = zero;
		uint32_t clipped = clip_polygon<V>(count, vs);
		if (clipped == 0) return zero;
		sa_polygon<V> pd;
		prepare_sa<V>(pd, clipped, vs, zero);
		density_factor = rcp(pd.solid_angle);
		for (uint32_t s = 0; s != S; ++s) {
			f3 dir = sample_sa<V>(pd, next_noise_2(p, noise));
			dir = mul_transposed(world_to_shading, dir);
			add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density_factor, sd, light);
		}
	}
	else {
		// projected solid angle sampling; flip the frame if the shading point is
		// behind the light so that the winding stays clockwise (:444-449)
		m43 world_to_cosine = ltc_in.world_to_cosine;
		float side = plane_distance(light, sd.position);
		if (side < 0.0f) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				world_to_shading.c[i].y = -world_to_shading.c[i].y;
				world_to_cosine.c[i].y = -world_to_cosine.c[i].y;
			}
		}
		if constexpr (STRATEGY == kStrategyDiffuseOnly || STRATEGY == kStrategyDiffuseGgxMis) {
			f3 vs[V];
#pragma unroll
			for (int i = 0; i < V - 1; ++i) vs[i] = mul_point(world_to_shading, light_vertex(light, min((uint32_t) i, p.max_light_vertex_count - 1)));
			vs[V - 1] = zero;
			uint32_t clipped = clip_polygon<V>(count, vs);
			if (clipped == 0) return zero;
			if constexpr (TECHNIQUE == kTechniquePsaArvo) {
				// Arvo's sampler, three Newton iterations (:462-481); the GGX tail below uses its
				// density without the cosine, like the reference
				psa_arvo<V> pa;
				prepare_psa_arvo<V>(pa, clipped, vs);
				if (pa.total <= 0.0f) return zero;
				if constexpr (ERROR == kErrorDiffuse) {
					f2 u = next_noise_2(p, noise);
					f3 dir = sample_psa_arvo<V>(pa, u, 3u);
					f2 e = psa_sampling_error_arvo<V>(pa, u, dir);
					f3 color = error_to_color(p, (p.error_index == 0) ? e.x : e.y);
					float exposure = load_f(p.constants, 176);
					return mk3(divide(color.x, exposure), divide(color.y, exposure), divide(color.z, exposure));
				}
				for (uint32_t s = 0; s != S; ++s) {
					f3 dir = sample_psa_arvo<V>(pa, next_noise_2(p, noise), 3u);
					float density = divide(dir.z, pa.total);
					dir = mul_transposed(world_to_shading, dir);
					add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density, sd, light);
				}
				density_factor = rcp(pa.total);
			}
			else {
				psa_polygon<V> pd;
				prepare_psa<V, kBiased>(pd, clipped, vs);
				if (pd.total <= 0.0f) return zero;
				if constexpr (ERROR == kErrorDiffuse) return display_sampling_error<V, kBiased>(p, pd, noise);
				for (uint32_t s = 0; s != S; ++s) {
					f3 dir = sample_psa<V, kBiased>(pd, next_noise_2(p, noise));
					float density = divide(dir.z, pd.total);
					dir = mul_transposed(world_to_shading, dir);
					add_light_mis_estimate<STRATEGY, RAYS, kTextured>(ctx, result, dir, density, sd, light);
				}
				density_factor = rcp(pd.total);
			}
		}
		else {
			// both techniques are prepared by the same code path (:506-547)
			if constexpr (ERROR == kErrorDiffuse || ERROR == kErrorSpecular) {
				psa_polygon<V> pd, ps;
				ps.total = 0.0f;
				ps.vertex_count = 0;
				ps.inner_ellipse_0 = mk2(0.0f, 0.0f);
				bool specular_culled = false;
#pragma unroll
				for (int t = 0; t != 2; ++t) {
					const m43& to_local = (t == 0) ? world_to_shading : world_to_cosine;
					if (t > 0) pd = ps;
					f3 vl[V];
#pragma unroll
					for (int j = 0; j < V - 1; ++j) vl[j] = mul_point(to_local, light_vertex(light, min((uint32_t) j, p.max_light_vertex_count - 1)));
					vl[V - 1] = zero;
					uint32_t clipped = clip_polygon<V>(count, vl);
					if (clipped == 0 && t == 0) return zero;
					else if (clipped == 0) { specular_culled = true; break; }
					prepare_psa<V, kBiased>(ps, clipped, vl);
				}
				if (specular_culled) ps.total = 0.0f;
				if (pd.total == 0.0f) return zero;
				if constexpr (ERROR == kErrorDiffuse) return display_sampling_error<V, kBiased>(p, pd, noise);
				if (ps.total > 0.0f) return display_sampling_error<V, kBiased>(p, ps, noise);
				return zero;
			}
			// The prepared polygons live in LDS (psa_compact, polygon_sampling.h): the diffuse one
			// first, then the specular one
			psa_compact<V> pd, ps;
			ps.total = 0.0f;
			ps.vertex_count = 0;
			ps.inner_slots = 0; ps.outer_slots = 0;
#pragma unroll
			for (int i = 0; i < V; ++i) ps.sector[i] = 0.0f;
			float2* const tables_d = ctx.psa_tables;
			// ... unless there is room for one table only (psa_table_in_memory(): V >= 6): then the specular polygon's table
			// is this thread's column of a table in device memory
			float2* const tables_s = psa_table_in_memory(V) ? ctx.psa_table_memory : ctx.psa_tables + kPsaTableSlots(V) * kPsaTableStride;
#pragma unroll
			for (int t = 0; t != 2; ++t) {
				const m43& to_local = (t == 0) ? world_to_shading : world_to_cosine;
				f3 vl[V];
#pragma unroll
				for (int j = 0; j < V - 1; ++j) vl[j] = mul_point(to_local, light_vertex(light, min((uint32_t) j, p.max_light_vertex_count - 1)));
				vl[V - 1] = zero;
				uint32_t clipped = clip_polygon<V>(count, vl);
				if (clipped == 0 && t == 0) return zero;
				else if (clipped == 0) break;  // the specular polygon is below the horizon: ps.total stays 0
				psa_polygon<V> prepared;
				prepare_psa<V, kBiased>(prepared, clipped, vl);
				if (t == 0) store_psa_tables<V>(pd, tables_d, prepared);
				else store_psa_tables<V>(ps, tables_s, prepared);
			}
			if (pd.total == 0.0f) return zero;
			float specular_albedo = ltc_in.albedo;
			float specular_weight = specular_albedo * ps.total;
			if constexpr (STRATEGY == kStrategySeparately) {
				for (uint32_t s = 0; s != S; ++s) {
					f3 dd = sample_psa_tables<V, kBiased>(pd, tables_d, next_noise_2(p, noise));
					dd = mul_transposed(world_to_shading, dd);
					settle_noise(noise);
					float lambert;
					bool candidate;
					f3 rb = radiance_brdf<true, false, kTextured>(p, lambert, candidate, dd, sd, light);
					accumulate<RAYS>(ctx, result, candidate, rb * pd.total, zero * pd.total, dd, sd, light);
					if (ps.total > 0.0f) {
						f3 dc = sample_psa_tables<V, kBiased>(ps, tables_s, next_noise_2(p, noise));
						f3 ds = normalize(cosine_to_shading(ltc_in, dc));
						float ltc_density = evaluate_ltc_density(ltc_in, ds, 1.0f);
						f3 dw = mul_transposed(world_to_shading, ds);
						f3 rb2 = radiance_brdf<false, true, kTextured>(p, lambert, candidate, dw, sd, light);
						if (!(ds.z <= 0.0f || dc.z <= 0.0f)) {
							f3 t = (rb2 * ds.z) * ps.total;
							f3 t0 = (zero * ds.z) * ps.total;
							accumulate<RAYS>(ctx, result, candidate,
								mk3(divide(t.x, ltc_density), divide(t.y, ltc_density), divide(t.z, ltc_density)),
								mk3(divide(t0.x, ltc_density), divide(t0.y, ltc_density), divide(t0.z, ltc_density)), dw, sd, light);
						}
						else if (RAYS == kRaysInline && candidate) {
							// the reference traces this ray even though the sample is discarded (:583-584)
							accumulate<RAYS>(ctx, result, candidate, zero, zero, dw, sd, light);
						}
					}
				}
			}
			else if constexpr (STRATEGY == kStrategyMis) {
				const int heuristic = p.mis_heuristic;
				const float visibility_estimate = load_f(p.constants, 156);
				f3 albedo = mk3(gmax(sd.diffuse_albedo.x, 0.01f), gmax(sd.diffuse_albedo.y, 0.01f), gmax(sd.diffuse_albedo.z, 0.01f));
				f3 diffuse_weight = albedo * pd.total;
				uint32_t technique_count = (ps.total > 0.0f) ? 2 : 1;
				float rcp_d = rcp(pd.total);
				float rcp_s = rcp(ps.total);
				f3 specular_weight_rgb = mk3(specular_weight, specular_weight, specular_weight);
				if (heuristic == kMisOptimal) {
					f3 radiance_over_pi = light_radiance(light) * kInvPi;
					diffuse_weight = diffuse_weight * radiance_over_pi;
					specular_weight_rgb = specular_weight_rgb * radiance_over_pi;
				}
				for (uint32_t s = 0; s != S; ++s) {
					f3 dir_d = sample_psa_tables<V, kBiased>(pd, tables_d, next_noise_2(p, noise));
					f3 dir_s = zero;
					if (ps.total > 0.0f) {
						dir_s = sample_psa_tables<V, kBiased>(ps, tables_s, next_noise_2(p, noise));
						dir_s = normalize(cosine_to_shading(ltc_in, dir_s));
					}
					settle_noise(noise);
					for (uint32_t j = 0; j != technique_count; ++j) {
						f3 ds = (j == 0) ? dir_d : dir_s;
						if (ds.z <= 0.0f) continue;
						float dens_d = ds.z * rcp_d;
						float dens_s = evaluate_ltc_density(ltc_in, ds, rcp_s);
						float lambert;
						bool candidate;
						f3 dw = mul_transposed(world_to_shading, ds);
						f3 rb = radiance_brdf<true, true, kTextured>(p, lambert, candidate, dw, sd, light);
						f3 integrand = rb * ds.z;
						f3 dark = zero * ds.z;
						f3 visible_term, hidden_term;
						if (j == 0 && ps.total <= 0.0f) {
							visible_term = candidate ? integrand * rcp(dens_d) : zero;
							hidden_term = zero;
						}
						else if (j == 0) {
							mis_estimate_pair(heuristic, integrand, dark, diffuse_weight, dens_d, specular_weight_rgb, dens_s, visibility_estimate, visible_term, hidden_term);
						}
						else {
							mis_estimate_pair(heuristic, integrand, dark, specular_weight_rgb, dens_s, diffuse_weight, dens_d, visibility_estimate, visible_term, hidden_term);
						}
						accumulate<RAYS>(ctx, result, candidate, visible_term, hidden_term, dw, sd, light);
					}
				}
			}
			else if constexpr (STRATEGY == kStrategyRandom) {
				float lum = (sd.diffuse_albedo.x * 0.21263901f + sd.diffuse_albedo.y * 0.71516868f) + sd.diffuse_albedo.z * 0.07219232f;
				float diffuse_albedo = gmax(lum, 0.01f);
				float diffuse_weight = diffuse_albedo * pd.total;
				float diffuse_ratio = divide(diffuse_weight, diffuse_weight + specular_weight);
				for (uint32_t s = 0; s != S; ++s) {
					f2 u = next_noise_2(p, noise);
					bool specular_selected = u.x >= diffuse_ratio;
					float offset = specular_selected ? 1.0f : 0.0f;
					u.x = divide(u.x - offset, diffuse_ratio - offset);
					f3 ds = specular_selected ? sample_psa_tables<V, kBiased>(ps, tables_s, u) : sample_psa_tables<V, kBiased>(pd, tables_d, u);
					if (specular_selected) ds = normalize(cosine_to_shading(ltc_in, ds));
					settle_noise(noise);
					float lambert = ds.z;
					float dens_d = lambert * diffuse_albedo;
					float dens_s = evaluate_ltc_density(ltc_in, ds, specular_albedo);
					float density = divide(dens_d + dens_s, diffuse_weight + specular_weight);
					bool candidate;
					f3 dw = mul_transposed(world_to_shading, ds);
					f3 rb = radiance_brdf<true, true, kTextured>(p, lambert, candidate, dw, sd, light);
					if (!(ds.z <= 0.0f)) {
						f3 t = rb * ds.z;
						f3 t0 = zero * ds.z;
						accumulate<RAYS>(ctx, result, candidate, mk3(divide(t.x, density), divide(t.y, density), divide(t.z, density)),
							mk3(divide(t0.x, density), divide(t0.y, density), divide(t0.z, density)), dw, sd, light);
					}
					else if (RAYS == kRaysInline && candidate) accumulate<RAYS>(ctx, result, candidate, zero, zero, dw, sd, light);
				}
			}
		}
	}

	if constexpr (STRATEGY == kStrategyDiffuseGgxMis) {
		// GGX VNDF samples that happen to hit the light (:676-709)
		f3 out_shading = mul_direction(world_to_shading, sd.outgoing);
		out_shading.y = 0.0f;
		for (uint32_t s = 0; s != S; ++s) {
			float ggx_density;
			f3 dg = sample_ggx_reflected(ggx_density, out_shading, sd.roughness, next_noise_2(p, noise));
			f3 dw = mul_transposed(world_to_shading, dg);
			settle_noise(noise);
			if (dg.z > 0.0f && light_ray_intersection(light, p.max_light_vertex_count, sd.position, dw, 0.0f)) {
				float lambert;
				bool candidate;
				f3 rb = radiance_brdf<true, true, kTextured>(p, lambert, candidate, dw, sd, light);
				float polygon_density = kIsPsa ? (lambert * density_factor) : density_factor;
				float weight = mis_weight_over_density(p.mis_heuristic, ggx_density, polygon_density);
				accumulate<RAYS>(ctx, result, candidate, (rb * lambert) * weight, (zero * lambert) * weight, dw, sd, light);
			}
		}
	}
	if constexpr (is_deferred(RAYS)) {
		// close this light's run of terms; the resolve kernel scales by 1 / S and adds it
#if VKR_SUM_FINAL_TERMS
		if (ctx.light_clear && (ctx.final_state & 1u)) flush_final_sum(ctx);
		ctx.final_state = 2u;
#endif
		if (ctx.light_has_terms && ctx.code_cursor + 1 < p.max_codes) {
			if (ctx.noise) settle_noise(*ctx.noise);
			p.codes[code_slot(p.thread_count, ctx.code_cursor, ctx.tid)] = (uint8_t) kCodeEndOfLight;
			++ctx.code_cursor;
			ctx.light_has_terms = false;
		}
	}
	return result * (1.0f / (float) S);
}

// Maps a thread of the frame to a pixel and its output slot.  `block` counts the 16x16 pixel
// blocks of the tiles this rank owns, `thread` the 256 pixels of a block; a wave is an 8x8 patch.
// Thread number block * 256 + thread is what the per-thread wavefront buffers are indexed by.
VKR_DEV bool locate_pixel(const shade_params& p, uint32_t block, uint32_t thread, uint32_t& px, uint32_t& py, size_t& out_index) {
	uint32_t blocks_per_side = p.tile_size >> 4;
	uint32_t blocks_per_tile = blocks_per_side * blocks_per_side;
	uint32_t local_tile = block / blocks_per_tile;
	uint32_t block_in_tile = block - local_tile * blocks_per_tile;
	uint32_t tile = local_tile * p.rank_count + p.rank;
	if (tile >= p.tile_count) return false;
	uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
	uint32_t by = block_in_tile / blocks_per_side, bx = block_in_tile - by * blocks_per_side;
	uint32_t wave = thread >> 6, lane = thread & 63;
	uint32_t ix = (bx << 4) + ((wave & 1) << 3) + (lane & 7);
	uint32_t iy = (by << 4) + ((wave >> 1) << 3) + (lane >> 3);
	px = tx * p.tile_size + ix;
	py = ty * p.tile_size + iy;
	if (!p.slab_layout) out_index = (size_t) py * p.width + px;
	else out_index = (size_t) local_tile * p.tile_size * p.tile_size + (size_t) iy * p.tile_size + ix;
	return px < p.width && py < p.height;
}

VKR_DEV void store_final_color(const shade_params& p, size_t out_index, f3 color) {
	float exposure = load_f(p.constants, 176);
	bool broken = !(fabsf(color.x) < __builtin_inff()) || !(fabsf(color.y) < __builtin_inff()) || !(fabsf(color.z) < __builtin_inff());
	if (broken) color = mk3(divide(1.0f, exposure), divide(0.0f, exposure), divide(0.8f, exposure));
	p.out_radiance[out_index] = make_float4(color.x * exposure, color.y * exposure, color.z * exposure, 1.0f);
}

// main, shading_pass.frag.glsl:824-866
// The kernel lives in a namespace per arithmetic mode: the exact and the fast translation
// units instantiate the same template arguments, and without distinct names their host
// stubs would be merged by the linker (one mode would silently run for both).
#if VKR_FAST_MATH
#define VKR_MODE_NAMESPACE fast_math
#elif VKR_MATH_MODE == 2
#define VKR_MODE_NAMESPACE libm_math
#else
#define VKR_MODE_NAMESPACE exact_math
#endif
inline namespace VKR_MODE_NAMESPACE {
// Occupancy.  With the polygon tables out of the registers (two-technique strategies) the kernel needs 163 - 171
// VGPRs up to V = 7, the one-technique variants 160 - 175; asking for three waves per SIMD makes
// the register allocator stop at 168 (no scratch at V = 5, 15 dwords per lane at V = 7; checked for every variant by
// profiles/tools/kernel_resources.sh).  LDS comes in granules of 1 280 bytes: both tables of V <= 5 fit twelve times
// into the 160 KB of a CU; from V = 6 on one table is in LDS and the other in device memory (psa_table_in_memory above),
// and LDS limits no variant below the twelve waves that its registers allow; V = 8 would spill.
constexpr uint32_t kShadeThreads = 64;
// Workgroups to launch for `blocks` 16x16 pixel blocks (whole groups of 8 blocks x 4 patches)
inline uint32_t shade_grid_size(uint32_t blocks) { return ((blocks + 7u) / 8u) * 32u; }
constexpr bool has_psa_tables(int strategy, int technique, int error) {
	return strategy >= kStrategySeparately && (technique == kTechniquePsa || technique == kTechniquePsaBiased) && error != kErrorDiffuse && error != kErrorSpecular;
}
// (Rays traced inside the kernel bring the traversal's registers with them: those variants would spill.)
// (waves per SIMD asked of the register allocator; VKR_SHADE_TABLE_WAVES for the kernels with polygon tables: the experiment above)
#ifndef VKR_SHADE_TABLE_WAVES
#define VKR_SHADE_TABLE_WAVES 3
#endif
constexpr int shade_min_workgroups(int strategy, int technique, int v, int rays, int error) {
	return ((technique == kTechniquePsa || technique == kTechniquePsaBiased) && error != kErrorDiffuse && error != kErrorSpecular
		&& v <= (has_psa_tables(strategy, technique, error) ? 7 : 6) && rays != kRaysInline) ? (has_psa_tables(strategy, technique, error) ? VKR_SHADE_TABLE_WAVES : 3) : 1;
}
// bytes of dynamic LDS of a shading workgroup: the polygon tables
constexpr uint32_t shade_lds_bytes(int strategy, int technique, int v, int error) {
	return has_psa_tables(strategy, technique, error) ? (psa_table_in_memory(v) ? 1u : 2u) * (2u * (uint32_t) v + 1u) * kShadeThreads * 8u : 0u;
}
template <int STRATEGY, int TECHNIQUE, int V, int RAYS, int ERROR = kErrorNone>
__global__ void __launch_bounds__(kShadeThreads, shade_min_workgroups(STRATEGY, TECHNIQUE, V, RAYS, ERROR)) shade_pixels(const shade_params p) {
	// A workgroup is ONE wave: the four 8x8 patches of a 16x16 block differ a lot in cost (background,
	// culled lights), and a wave that is done early gives its registers and LDS back at once instead
	// of waiting for its block (mean resident waves per SIMD 2.3 -> see profiles/).  Workgroup b runs
	// on XCD b % 8; the four patches of a block are the workgroups b, b + 8, b + 16, b + 24 of a
	// group of 32, so they share that XCD's L2.
	fill_atan_rows();  // (nothing unless VKR_ATAN_TABLE, device_math.h)
	const uint32_t b = blockIdx.x;
	const uint32_t local_block = ((b >> 5) << 3) | (b & 7u);
	const uint32_t block = p.first_block + local_block;
	const uint32_t thread = (((b >> 3) & 3u) << 6) | threadIdx.x;
	uint32_t px, py;
	size_t out_index;
	bool inside = local_block < p.block_count && locate_pixel(p, block, thread, px, py, out_index);
	// ray queue of this wave: one of the 64 queues of its XCD, so a queue counter's cache line is
	// only ever touched from one L2
	uint32_t queue = (b & 7u) * 64u + ((b >> 3) & 63u);
	// the polygon tables exist for the techniques that prepare two polygons per light in registers
	constexpr bool kTables = has_psa_tables(STRATEGY, TECHNIQUE, ERROR);
	// (dynamic LDS, shade_lds_bytes() at the launch: a static array would let the compiler conclude from
	// its size that a third wave cannot fit and drop the register limit that goes with three - at V = 7 ten
	// waves fit a CU, i.e. three on two of the four SIMDs)
	extern __shared__ float2 psa_tables[];
	// (the wavefront buffers are indexed by the thread's number within this launch)
	pixel_context ctx = {p, 0, local_block * 256u + thread, 0, 0, false, false, 0u, 0u, nullptr, queue, mk3(0.0f, 0.0f, 0.0f), 2u, kTables ? psa_tables + threadIdx.x : nullptr,
		(kTables && psa_table_in_memory(V) && p.psa_table_memory) ? p.psa_table_memory + (size_t) (p.psa_table_by_wave_slot ? hardware_wave_slot() : b) * (kPsaTableSlots(V) * kPsaTableStride) + threadIdx.x : nullptr, nullptr};
	if constexpr (RAYS == kRaysDeferredBlocks) {
		// (the waves of a workgroup never touch each other's entry: no barrier)
		lds_state_word* state = ray_block_state();
		if ((threadIdx.x & 63u) == 0) { state[0] = 0; state[1] = 0; state[2] = 0; }
	}
	if (inside) {
		const uint8_t* c = p.constants;
		uint32_t primitive = p.visibility[(size_t) py * p.width + px];
		f3 color = mk3(0.0f, 0.0f, 0.0f);
		float fx = (float) (int32_t) px, fy = (float) (int32_t) py;
		f3 ray = mk3(
			(load_f(c, 96) * fx + load_f(c, 100) * fy) + load_f(c, 104) * 1.0f,
			(load_f(c, 112) * fx + load_f(c, 116) * fy) + load_f(c, 120) * 1.0f,
			(load_f(c, 128) * fx + load_f(c, 132) * fy) + load_f(c, 136) * 1.0f);
		shading_data sd;
		f3 end_xyz = ray;
		float end_w = 0.0f;
		if (primitive != 0xFFFFFFFFu) {
			sd = get_shading_data(p, primitive, ray, (size_t) py * p.width + px);
			end_xyz = sd.position;
			end_w = 1.0f;
			// every shadow ray of this pixel starts here
			if constexpr (is_deferred(RAYS)) p.ray_origins[ctx.tid] = make_float4(sd.position.x, sd.position.y, sd.position.z, 0.0f);
		}
		if (p.show_polygonal_lights) {
			f3 camera = load_f3(c, 144);
			f3 view_dir = normalize(ray);
			for (uint32_t i = 0; i != p.light_count; ++i) {
				light_ref light = get_light(p, i);
				if (light_ray_intersection(light, p.max_light_vertex_count, camera, end_xyz, end_w))
					color = color + polygon_radiance<ERROR != kErrorNone>(p, view_dir, camera, light);
			}
		}
		if (primitive != 0xFFFFFFFFu) {
			float fresnel_luminance = (sd.fresnel_0.x * 0.2126f + sd.fresnel_0.y * 0.7152f) + sd.fresnel_0.z * 0.0722f;
			ltc_coefficients ltc = get_ltc_coefficients(p, fresnel_luminance, sd.roughness, sd.position, sd.normal, sd.outgoing);
			noise_accessor noise = make_noise_accessor(p, px, py);
			ctx.noise = &noise;
			for (uint32_t i = 0; i != p.light_count; ++i) {
				light_ref light = get_light(p, i);
				if constexpr (is_deferred(RAYS)) {
					// the verdict of the shaft walk (light_shafts.h): 1 clear, 2 | n << 8 a list of n triangles, else trace
					uint32_t verdict = p.shaft_clear != nullptr ? *(constant_uint_pointer) (uintptr_t) (p.shaft_clear + ((size_t) b * p.light_count + i)) : 0u;
					bool listed = kUseShaftLists && (verdict & 0xFFu) == 2u && p.shaft_lists != nullptr;
					ctx.light_clear = verdict == 1u || listed;
					ctx.list_count = listed ? ((verdict >> 8) & 0x1Fu) : 0u;
					ctx.list = p.shaft_lists + ((size_t) b * p.light_count + i) * (kShaftListMax * kShaftListEntry);
					ctx.light_index = i;
				}
				color = color + evaluate_light<STRATEGY, TECHNIQUE, V, RAYS, ERROR>(ctx, sd, ltc, light, noise);
			}
		}
		if constexpr (is_deferred(RAYS)) {
			// hand over to trace_shadow_rays / resolve_shadow_terms: the colour so far
			// (light display) and the terminated term stream
			// (without the light display the colour so far is +0, and the resolve kernel knows)
			if (p.show_polygonal_lights) p.base_color[ctx.tid] = make_float4(color.x, color.y, color.z, 0.0f);
			p.codes[code_slot(p.thread_count, ctx.code_cursor, ctx.tid)] = (uint8_t) kCodeEnd;
		}
		else
			store_final_color(p, out_index, color);
	}
	if constexpr (RAYS == kRaysDeferredBlocks) close_ray_block(p, queue);
	if (RAYS == kRaysInline && p.ray_counter) {
		// one atomic per wave
		uint32_t rays = ctx.rays;
#pragma unroll
		for (int offset = 32; offset > 0; offset >>= 1) rays += __shfl_xor(rays, offset);
		if ((threadIdx.x & 63) == 0 && rays) atomicAdd(p.ray_counter, (unsigned long long) rays);
	}
}

}  // inline namespace VKR_MODE_NAMESPACE

}  // namespace vkr
