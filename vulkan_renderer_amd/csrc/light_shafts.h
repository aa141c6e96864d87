// Light shafts: which (8x8 pixel patch, polygonal light) pairs need no shadow rays at all.
//
// The reference traces one ray query per sample (src/shaders/shading_pass.frag.glsl:120-138).  Most of those
// rays cross empty space: every sample of a patch of neighbouring pixels toward one light lies inside the convex
// hull of the patch's shading positions and the light polygon - the shaft of Haines & Wallace, "Shaft Culling for
// Efficient Ray-Cast Radiosity" (1991) - and if no triangle of the scene reaches into that hull, every one of
// those rays arrives, whatever its direction.  k_light_shafts decides that per patch and light with ONE
// conservative walk of the four-wide BVH by the whole wave (64 boxes or 64 triangles per step, one per lane,
// the shaft's planes wave-uniform); the shading kernel then writes the terms of such a light as final ones
// instead of queueing their rays (accumulate(), shading_kernel.h).  The result of every ray query is unchanged:
// "clear" is only ever claimed when no triangle can be hit, with margins that cover the rounding of the kernels
// that would have traced the rays, so frames stay bit-identical (tests/test_gpu_light_shafts.py renders them
// with and without).  What it buys depends on the scene: at config 3 the benchmark scene is left with 1.6 M of its
// 28.8 M rays to trace, the large scene - slats and fences between every patch and every light - with 21.8 of 22.2 M.
//
// Which rays a shaft holds.  Samples aim at the polygon, but where the sampling breaks down numerically (a polygon that
// is a sliver in the space it is sampled in: seed 6 of the random sweep has a specular sample 2.5 degrees off a light
// seen edge-on) the reference's directions leave it, and the ray query toward the light's PLANE still decides the term.
// So the shaft is built around a rectangle R in the light's plane space that contains the polygon with a margin of
// 1/32 of its size, and the shading kernel lets a ray skip the tracing only if it meets the light's plane inside a
// slightly smaller rectangle (shaft_holds_ray, shading_kernel.h; out_rectangles carries it); every other ray is traced.
//
// Conservative by construction.  With B the bounding box of the patch's shading positions (plus margin), c its
// centre, support(n) = |n.x| h.x + |n.y| h.y + |n.z| h.z the reach of B along n, L'_i the four corners of R in world
// space, n_L the light's plane:
//   side plane k     through c, L'_k, L'_k+1, pushed outwards by support(n)   contains the box and the polygon
//   far cap          n_L . x <= n_L . L_0 + margin                              rays end on the light's plane
//   near cap         a . x >= a . c - support(a), a towards the centroid      only if every L'_i lies in front of it
// (A box, not a sphere: a patch on a floor seen at a grazing angle is long and flat, and a sphere around it would
// make the shaft swallow a strip of the floor as wide as the patch is long.)
// Their intersection contains every ray of the patch toward the light.  A box is skipped when it lies outside
// one plane (or outside the bounding box of sphere and polygon); a triangle is harmless when
//   (i)  its three vertices lie outside one of those planes, or
//   (ii) its own plane separates it from the rays: the light's vertices lie on one side of it by a margin and
//        every ray's first point o + t_min u (t_min = 1e-3, the ray query's) does too - which is what lets a
//        patch see past the very surface it lies on.
// Anything else - also a walk that gets long or a queue that fills up - means "trace the rays".
//
// Occluder lists.  A walk that meets a triangle it cannot rule out goes on and collects such triangles, up to kShaftListMax.
// If it reaches its end, the list is complete: every triangle of the scene is either on it or cannot be hit by a ray that
// the shaft holds.  The shading kernel then decides each such ray itself - ray_triangle_edges() of lbvh.h, the function
// the tracing kernels call at their leaves, with the same origin, direction and interval, on the vertex and the two
// edges that the list stores (computed here as ray_triangle computes them) - and the ray query's answer "some triangle is
// hit" is the OR over the list.  (Both directions rest on margins that cover rounding: the walk must not drop a triangle that
// the tracing kernel would report - the margins of the tests above -, and the tracing kernel must not cull a triangle
// that its own triangle test accepts - its boxes are rounded outwards, lbvh.h kGridMargin.)  Around occluders most pairs
// end this way: config 3 of the benchmark scene 85 % of the pairs that are not clear, 4.3 triangles on average.
#pragma once
#include "shading_kernel.h"

namespace vkr {

constexpr uint32_t kShaftMaxVertices = 4;                     // the shaft is built around a rectangle in the light's plane
constexpr uint32_t kShaftMaxPlanes = kShaftMaxVertices + 3;   // sides, far cap, near cap, the patch's own plane
constexpr uint32_t kShaftLights = 8;                          // lights whose shafts are walked together
constexpr uint32_t kShaftLightShift = 27;                     // an entry of the queues: node or triangle | light << 27
constexpr uint32_t kShaftFrontier = 640;                      // inner nodes waiting (LDS); more -> not clear
constexpr uint32_t kShaftLeaves = 320;                        // triangles waiting
// Measured (profiles/r05n, r05o, r05p; config 3 / config 4, shaft kernel alone): test (iii) with all 64 shading positions
// instead of the corners of their bounding box finds 0.2 % more clear pairs and costs 0.288 instead of 0.173 ms / 0.90
// instead of 0.80 ms (VKR_SHAFT_ORIGIN_LOOP=1); batches of 16 or 8 triangles instead of 32 cost 0.35 / 0.39 ms, 64 the
// same as 32; the whole per-light set-up computed by every lane alike instead of four lanes and a dozen wave-wide
// reductions saves 4 % of the kernel alone but takes 126 instead of 79 registers, and the frame with three of them in
// flight gets slower (1.441 vs 1.428 ms).  With the occluder lists (walks go on where they used to end; profiles/r06d):
// the reductions over the rectangle's corners within a quad instead of the wave 0.241 -> 0.224 ms, and batches of 48 /
// 64 triangles 0.212 / 0.210 ms.
#ifndef VKR_SHAFT_ORIGIN_LOOP
#define VKR_SHAFT_ORIGIN_LOOP 0
#endif
#ifndef VKR_SHAFT_LEAF_BATCH
#define VKR_SHAFT_LEAF_BATCH 64
#endif
constexpr uint32_t kShaftLeafBatch = VKR_SHAFT_LEAF_BATCH;         // triangles that must wait before a batch of them is tested
constexpr uint32_t kShaftMaxSteps = 40;                       // steps of 16 nodes (plus a fifth of it per light); more -> not clear (run-time knob VKR_SHAFT_MAX_STEPS)
constexpr uint32_t kShaftSmallLaunchSteps = 12;               // the same for launches of less than 12 288 patches (shading_pass.hip)
constexpr float kShaftDilation = 1.0f / 32.0f;
// Measured and not adopted (profiles/r05h/): cutting every candidate triangle by all planes of the shaft (test (iii) below)
// finds 7 % more clear pairs at config 3 (25.0 instead of 23.3 % of all pairs) but costs the kernel 17 registers and 300
// bytes of scratch per lane: 1.459 instead of 1.414 ms per frame.
#ifndef VKR_SHAFT_CLIPPING
#define VKR_SHAFT_CLIPPING 0
#endif
// Occluder lists (above).  Measured (profiles/r05zg, frame period of config 3 / config 4 with kShaftListMax = 4, 6, 8, 12,
// 16; none: 1.400 / 22.6 ms): 1.304 / 21.06, 1.240 / 20.15, 1.217 / 19.41, 1.187 / 19.01, 1.192 / 18.87 ms - a triangle
// test costs a twelfth of what tracing the ray costs, and the walks that give up do so a little later.
// (kShaftListMax, VKR_SHAFT_LIST: shading_kernel.h)

// what the walk needs to know about one (patch, light), wave-uniform, in LDS
struct shaft_state {
	float plane[kShaftMaxPlanes][4];   // n . x <= d  (n unit length)
	float box_lo[3], box_hi[3];        // bounding box of sphere and dilated polygon
	float light[kShaftMaxVertices][3]; // dilated vertices
	uint32_t plane_count, vertex_count;
	// rays toward the light climb at least this steeply out of a plane that the light lies `h` above:
	// (h - g) / reach for an origin at height g
	float reach;                        // largest distance from an origin to a light vertex
	// flat patches: how far above the patch's own plane the first point of every ray lies at least (the light on one
	// side of that plane, else negative), measured along the plane's normal towards the light (flat_sign x the normal)
	float flat_first, flat_sign;
};

// why a pair is not clear (the word that the shading kernel reads is 1 for clear pairs and one of these otherwise)
// (kShaftList: bits 8 ... 12 hold the number of triangles on the pair's list)
// kShaftResting: the pair's last walk ended at a triangle too many and this frame did not repeat it (below); bits 8 ... 15
// count the frames since that walk
enum { kShaftClear = 1, kShaftList = 2, kShaftNoPixels = 16, kShaftGeometry = 17, kShaftTooLong = 18, kShaftQueueFull = 19, kShaftTriangle = 20, kShaftResting = 21 };
// Walks that find nothing (round 5).  Behind fences and louvres - the large scene - 64 % of the walks end with "more
// triangles in the way than a list holds": they cost what a successful walk costs and their rays are traced all the same
// (5 % of that scene's frame).  A verdict is only ever a HINT - a pair without one has its rays traced, and every term
// comes out the same either way - so it may lean on the past: the table of a frame context keeps the verdicts of the
// frame it rendered before (three frames back with three frames in flight), and a pair whose walk failed that way then
// is not walked again for `rest_frames` of its context's frames.  After that the walk is repeated, so a pair that a moving
// camera or light has freed gets its shaft back a few frames late, nothing more.  What the table holds before the first
// frame is zeros (the host clears it), i.e. "walk".  VKR_SHAFT_REST=0 walks every pair in every frame.
constexpr uint32_t kShaftRestFrames = 7;

struct shaft_patch {
#if VKR_SHAFT_ORIGIN_LOOP
	float origin[64][3];
#endif
	uint64_t valid;                     // lanes with a shading position
	float centre[3], half[3], radius;   // bounding box of the positions (with margin), length of its half diagonal
	// the patch's own plane if all its positions lie on one (within flat_tolerance): n unit, n . x = d
	float flat_normal[3], flat_d;
	uint32_t flat;
};

VKR_DEV float wave_min(float v) {
#pragma unroll
	for (int offset = 32; offset > 0; offset >>= 1) v = fminf(v, __shfl_xor(v, offset));
	return v;
}
VKR_DEV float wave_max(float v) {
#pragma unroll
	for (int offset = 32; offset > 0; offset >>= 1) v = fmaxf(v, __shfl_xor(v, offset));
	return v;
}

// (over the four lanes of a quad: every quad of the wave holds the four corners of the light's rectangle)
VKR_DEV float quad_min(float v) {
	v = fminf(v, __shfl_xor(v, 1));
	return fminf(v, __shfl_xor(v, 2));
}
VKR_DEV float quad_max(float v) {
	v = fmaxf(v, __shfl_xor(v, 1));
	return fmaxf(v, __shfl_xor(v, 2));
}

// One box or triangle per lane against the planes of the shaft.  true: the box may reach into the shaft.
VKR_DEV bool shaft_box_inside(const shaft_state& s, f3 lo, f3 hi) {
	if (hi.x < s.box_lo[0] || lo.x > s.box_hi[0] || hi.y < s.box_lo[1] || lo.y > s.box_hi[1] || hi.z < s.box_lo[2] || lo.z > s.box_hi[2]) return false;
	f3 centre = mk3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
	f3 half = mk3(0.5f * (hi.x - lo.x), 0.5f * (hi.y - lo.y), 0.5f * (hi.z - lo.z));
	bool inside = true;
	for (uint32_t k = 0; k != s.plane_count; ++k) {
		float nx = s.plane[k][0], ny = s.plane[k][1], nz = s.plane[k][2];
		// the corner of the box that is deepest inside the half space
		float nearest = fmaf(nx, centre.x, fmaf(ny, centre.y, nz * centre.z)) - fmaf(fabsf(nx), half.x, fmaf(fabsf(ny), half.y, fabsf(nz) * half.z));
		inside = inside && nearest <= s.plane[k][3];
	}
	return inside;
}

// true: the triangle cannot be hit by any ray of the patch toward the light
VKR_DEV bool shaft_triangle_harmless(const shaft_state& s, const shaft_patch& patch, f3 a, f3 b, f3 c, float margin) {
	// (i) outside one plane of the shaft
	for (uint32_t k = 0; k != s.plane_count; ++k) {
		float nx = s.plane[k][0], ny = s.plane[k][1], nz = s.plane[k][2], d = s.plane[k][3];
		float da = fmaf(nx, a.x, fmaf(ny, a.y, nz * a.z)), db = fmaf(nx, b.x, fmaf(ny, b.y, nz * b.z)), dc = fmaf(nx, c.x, fmaf(ny, c.y, nz * c.z));
		if (fminf(da, fminf(db, dc)) > d) return true;
	}
	// (ii) for a flat patch with the light on one side of its plane: every ray starts on that plane (within `margin`),
	// its first point lies flat_first above it and it climbs from there - a triangle that stays on or below the plane
	// (the surface the patch lies on, the other faces of the same box, the floor under it) cannot be reached
	if (patch.flat && s.flat_first > 3.0f * margin) {
		float ha = s.flat_sign * (patch.flat_normal[0] * a.x + patch.flat_normal[1] * a.y + patch.flat_normal[2] * a.z - patch.flat_d);
		float hb = s.flat_sign * (patch.flat_normal[0] * b.x + patch.flat_normal[1] * b.y + patch.flat_normal[2] * b.z - patch.flat_d);
		float hc = s.flat_sign * (patch.flat_normal[0] * c.x + patch.flat_normal[1] * c.y + patch.flat_normal[2] * c.z - patch.flat_d);
		if (fmaxf(ha, fmaxf(hb, hc)) <= margin) return true;
	}
#if VKR_SHAFT_CLIPPING
	// (iii) what is left of the triangle inside ALL planes at once: nothing?  (A large triangle next to the narrow end
	// of the shaft - the top of a box beside a patch on the floor - lies outside the shaft without lying outside any one
	// of its planes.)  The triangle is cut by one plane after the other; planes are moved outwards by `margin` first.
	{
		constexpr int kMost = 12;
		float x[2][kMost], y[2][kMost], z[2][kMost];
		x[0][0] = a.x; y[0][0] = a.y; z[0][0] = a.z;
		x[0][1] = b.x; y[0][1] = b.y; z[0][1] = b.z;
		x[0][2] = c.x; y[0][2] = c.y; z[0][2] = c.z;
		int count = 3, from = 0;
		for (uint32_t k = 0; k != s.plane_count && count != 0 && count <= kMost - 2; ++k) {
			float nx = s.plane[k][0], ny = s.plane[k][1], nz = s.plane[k][2], d = s.plane[k][3] + margin;
			int to = from ^ 1, kept = 0;
			float previous_x = x[from][count - 1], previous_y = y[from][count - 1], previous_z = z[from][count - 1];
			float previous_distance = fmaf(nx, previous_x, fmaf(ny, previous_y, nz * previous_z)) - d;
			for (int i = 0; i != count; ++i) {
				float cx = x[from][i], cy = y[from][i], cz = z[from][i];
				float distance = fmaf(nx, cx, fmaf(ny, cy, nz * cz)) - d;
				if ((distance <= 0.0f) != (previous_distance <= 0.0f)) {
					// the edge crosses the plane
					float t = previous_distance / (previous_distance - distance);
					x[to][kept] = fmaf(t, cx - previous_x, previous_x); y[to][kept] = fmaf(t, cy - previous_y, previous_y); z[to][kept] = fmaf(t, cz - previous_z, previous_z);
					++kept;
				}
				if (distance <= 0.0f) { x[to][kept] = cx; y[to][kept] = cy; z[to][kept] = cz; ++kept; }
				previous_x = cx; previous_y = cy; previous_z = cz; previous_distance = distance;
			}
			count = kept;
			from = to;
		}
		if (count == 0) return true;
	}
#endif
	// (ii) in general: the triangle's own plane separates it from the rays - the light's vertices and the first
	// points of all rays on one side
	f3 n = cross(b - a, c - a);
	float length_squared = dot(n, n);
	if (!(length_squared > 1.0e-30f)) return false;  // (no plane to speak of: whether rounding lets a ray "hit" a sliver is the tracing kernel's business)
	float scale = __builtin_amdgcn_rsqf(length_squared);
	n = n * scale;
	float h_min = 3.0e38f, h_max = -3.0e38f;
	for (uint32_t i = 0; i != s.vertex_count; ++i) {
		float h = n.x * (s.light[i][0] - a.x) + n.y * (s.light[i][1] - a.y) + n.z * (s.light[i][2] - a.z);
		h_min = fminf(h_min, h); h_max = fmaxf(h_max, h);
	}
	// look at the triangle from the light's side
	if (h_max < 0.0f) { n = -n; float t = h_min; h_min = -h_max; h_max = -t; }
	if (!(h_min > 4.0f * margin)) return false;
	// lowest origin above the plane
#if VKR_SHAFT_ORIGIN_LOOP
	float g_min = 3.0e38f;
	uint64_t lanes = patch.valid;
	while (lanes) {
		int j = __builtin_ctzll(lanes);
		lanes &= lanes - 1;
		float g = n.x * (patch.origin[j][0] - a.x) + n.y * (patch.origin[j][1] - a.y) + n.z * (patch.origin[j][2] - a.z);
		g_min = fminf(g_min, g);
	}
	g_min -= margin;
#else
	// (the lowest corner of the positions' bounding box: never higher than the lowest position)
	float g_min = n.x * (patch.centre[0] - a.x) + n.y * (patch.centre[1] - a.y) + n.z * (patch.centre[2] - a.z)
		- (fabsf(n.x) * patch.half[0] + fabsf(n.y) * patch.half[1] + fabsf(n.z) * patch.half[2]) - margin;
#endif
	// the first point of a ray from height g: g + t_min n . u with n . u >= (h_min - g) / reach; grows with g
	float first = g_min + 1.0e-3f * (h_min - g_min) / s.reach;
	return g_min <= h_min && first > 2.0f * margin;
}

// The rectangle of a light in its plane space (u_min, v_min, u_max, v_max): the bounding box of the polygon's plane-space
// vertices (reference polygonal_light.h: vertices_plane_space) grown on every side by `room` x (1/32 of its size + eight
// margins in plane units).
VKR_DEV float4 shaft_rectangle(const light_ref& light, float margin, float room) {
	uint32_t count = light_vertex_count(light);
	float u_min = 3.0e38f, v_min = 3.0e38f, u_max = -3.0e38f, v_max = -3.0e38f;
	for (uint32_t i = 0; i < count; ++i) {
		float u = load_f(light.base, 160 + 16 * i), v = load_f(light.base, 164 + 16 * i);
		u_min = fminf(u_min, u); u_max = fmaxf(u_max, u);
		v_min = fminf(v_min, v); v_max = fmaxf(v_max, v);
	}
	float grow_u = room * (kShaftDilation * (u_max - u_min) + 8.0f * margin * fabsf(load_f(light.base, 44)));
	float grow_v = room * (kShaftDilation * (v_max - v_min) + 8.0f * margin * fabsf(load_f(light.base, 60)));
	return make_float4(u_min - grow_u, v_min - grow_v, u_max + grow_u, v_max + grow_v);
}
// corner 0 ... 3 (counter-clockwise in plane space) of such a rectangle in world space: T + R (u s_x, v s_y, 0)
VKR_DEV f3 shaft_rectangle_corner(const light_ref& light, float4 rectangle, uint32_t corner) {
	float u = (corner == 0u || corner == 3u) ? rectangle.x : rectangle.z;
	float v = (corner < 2u) ? rectangle.y : rectangle.w;
	u *= load_f(light.base, 12);
	v *= load_f(light.base, 28);
	return light_translation(light) + light_rotation_column(light, 0) * u + light_rotation_column(light, 1) * v;
}

// `b` numbers the workgroups like shade_pixels does (one 8x8 patch each); out_clear[b * light_count + i] = 1 when no
// ray of that patch toward light i can be blocked.  extent: largest coordinate difference of the scene (margins).
// work_counters (diagnostics, may be NULL): [0] steps of the walks, [1] batches of triangles, [2] walks
// out_lists (may be NULL: then a walk ends at the first triangle in the way): kShaftListMax entries of kShaftListEntry
// floats per (patch, light), in the order of out_clear - a vertex of the triangle and the two edges that leave it.
// out_clear is read before it is written: the verdict that this patch and light got when the table was last used (see kShaftResting)
__global__ void __launch_bounds__(64) k_light_shafts(const shade_params p, const uint4* __restrict__ wide_nodes, uint32_t* out_clear, float4* __restrict__ out_rectangles, float* __restrict__ out_lists, float extent, unsigned long long* work_counters, uint32_t rest_frames, uint32_t max_steps) {
	__shared__ shaft_patch patch;
	__shared__ shaft_state shafts[kShaftLights];
	__shared__ uint32_t frontier[kShaftFrontier];
	__shared__ uint32_t leaves[kShaftLeaves];
	__shared__ uint32_t failed_lights, failed_triangle[kShaftLights];
	__shared__ uint32_t listed_count[kShaftLights];
	__shared__ uint32_t list_count[kShaftLights], list_slot[kShaftLights][kShaftListMax ? kShaftListMax : 1], list_triangle[kShaftLights][kShaftListMax ? kShaftListMax : 1];
	const uint32_t list_capacity = out_lists ? kShaftListMax : 0u;
	const uint32_t lane = threadIdx.x;
	const uint32_t b = blockIdx.x;
	const uint32_t local_block = ((b >> 5) << 3) | (b & 7u);
	const uint32_t thread = (((b >> 3) & 3u) << 6) | lane;
	uint32_t px, py;
	size_t out_index;
	bool inside = local_block < p.block_count && locate_pixel(p, p.first_block + local_block, thread, px, py, out_index);
	uint32_t primitive = inside ? p.visibility[(size_t) py * p.width + px] : 0xFFFFFFFFu;
	bool shaded = primitive != 0xFFFFFFFFu;
	uint32_t* clear = out_clear + (size_t) b * p.light_count;
	const uint64_t valid = __ballot(shaded);
	// What the tests below allow for: shading positions and triangle vertices are a few units in the last place of the
	// scene's coordinates off the planes they lie on, and so is what the tracing kernels compute with them - 5e-7 of
	// the extent is ten units in the last place of the largest coordinate.
	const float margin = 5.0e-7f * extent;
	if (b == 0) {
		// the rectangles that the shading kernel tests its rays against (the same for every patch: one workgroup writes them)
		for (uint32_t i = lane; i < p.light_count; i += 64u) out_rectangles[i] = shaft_rectangle(get_light(p, i), margin, 1.0f);
	}
	if (valid == 0) {
		for (uint32_t i = lane; i < p.light_count; i += 64u) clear[i] = kShaftNoPixels;
		return;
	}
	// ---- the patch: shading positions as the shading kernel computes them (get_shading_data) ---------------
	f3 position = mk3(0.0f, 0.0f, 0.0f), face_normal = position;
	float face_d = 0.0f;
	if (shaded) {
		const uint8_t* c = p.constants;
		float fx = (float) (int32_t) px, fy = (float) (int32_t) py;
		f3 ray = mk3(
			(load_f(c, 96) * fx + load_f(c, 100) * fy) + load_f(c, 104) * 1.0f,
			(load_f(c, 112) * fx + load_f(c, 116) * fy) + load_f(c, 120) * 1.0f,
			(load_f(c, 128) * fx + load_f(c, 132) * fy) + load_f(c, 136) * 1.0f);
		triangle_hit t = intersect_primitive(p, primitive, ray);
		position = fma3(t.b0, t.pos[0], fma3(t.b1, t.pos[1], t.pos[2] * t.b2));
		face_normal = cross(t.e0, t.e1);
		face_normal = face_normal * __builtin_amdgcn_rsqf(fmaxf(dot(face_normal, face_normal), 1.0e-38f));
		face_d = dot(face_normal, t.pos[0]);
#if VKR_SHAFT_ORIGIN_LOOP
		patch.origin[lane][0] = position.x; patch.origin[lane][1] = position.y; patch.origin[lane][2] = position.z;
#endif
	}
	const float big = 3.0e38f;
	f3 lo = mk3(wave_min(shaded ? position.x : big), wave_min(shaded ? position.y : big), wave_min(shaded ? position.z : big));
	f3 hi = mk3(wave_max(shaded ? position.x : -big), wave_max(shaded ? position.y : -big), wave_max(shaded ? position.z : -big));
	f3 centre = mk3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
	f3 diagonal = hi - lo;
	const f3 half = mk3(0.5f * diagonal.x * 1.0001f + 2.0f * margin, 0.5f * diagonal.y * 1.0001f + 2.0f * margin, 0.5f * diagonal.z * 1.0001f + 2.0f * margin);
	float radius = __builtin_sqrtf(dot(half, half)) * 1.0001f;
	// is the patch flat?  the plane of the first valid lane's triangle, and every position on it
	const int first_lane = __builtin_ctzll(valid);
	f3 plane_n = mk3(__shfl(face_normal.x, first_lane), __shfl(face_normal.y, first_lane), __shfl(face_normal.z, first_lane));
	float plane_d = __shfl(face_d, first_lane);
	float off_plane = wave_max(shaded ? fabsf(dot(plane_n, position) - plane_d) : 0.0f);
	if (lane == 0) {
		patch.valid = valid;
		patch.centre[0] = centre.x; patch.centre[1] = centre.y; patch.centre[2] = centre.z;
		patch.half[0] = half.x; patch.half[1] = half.y; patch.half[2] = half.z;
		patch.radius = radius;
		patch.flat_normal[0] = plane_n.x; patch.flat_normal[1] = plane_n.y; patch.flat_normal[2] = plane_n.z;
		patch.flat_d = plane_d;
		patch.flat = (off_plane <= margin && dot(plane_n, plane_n) > 0.5f) ? 1u : 0u;
	}
	__syncthreads();
	const f3 grid_origin = p.bvh.grid_origin;
	const f3 cell = mk3(1.0f / p.bvh.grid_inverse_cell.x, 1.0f / p.bvh.grid_inverse_cell.y, 1.0f / p.bvh.grid_inverse_cell.z);
	// The lights are walked TOGETHER, kShaftLights at a time: a step of the walk is one round trip to memory (the
	// nodes), and a shaft through open space needs about one step per level of the tree - walked one after the
	// other, a patch with four lights spent forty round trips where ten do (profiles/r05f/).  An entry of the frontier
	// is a node and the light whose shaft reached it.
	for (uint32_t chunk = 0; chunk < p.light_count; chunk += kShaftLights) {
		const uint32_t chunk_lights = min(kShaftLights, p.light_count - chunk);
		uint32_t alive = 0;  // lights of the chunk whose shafts are still being walked (bit k: light chunk + k)
		__syncthreads();    // (the previous chunk's walk is over: the shared state may change)
		for (uint32_t k = 0; k != chunk_lights; ++k) {
			shaft_state& s = shafts[k];
			light_ref light = get_light(p, chunk + k);
			constexpr uint32_t vertex_count = 4;
			// ---- the shaft of this light (lanes build one plane each) -------------------------------------------
			// the rectangle the shaft is built around: the one the rays are tested against, and half as much room again
			const float4 rectangle = shaft_rectangle(light, margin, 1.5f);
			bool possible = light_vertex_count(light) >= 3u && rectangle.z > rectangle.x && rectangle.w > rectangle.y;
			// a pair whose last walk met too many triangles rests for a few frames (the same word for every lane)
			uint32_t resting = 0u;
			if (rest_frames != 0u) {
				const uint32_t before = clear[chunk + k], kind = before & 0xFFu, age = (before >> 8) & 0xFFu;
				if (kind == kShaftTriangle) resting = kShaftResting | (1u << 8);
				else if (kind == kShaftResting && age < rest_frames) resting = kShaftResting | ((age + 1u) << 8);
			}
			possible = possible && resting == 0u;
			// every position on the same side of the light's plane, away from it
			float side = shaded ? plane_distance(light, position) : 0.0f;
			float side_min = wave_min(shaded ? side : big), side_max = wave_max(shaded ? side : -big);
			possible = possible && (side_min > 8.0f * margin || side_max < -8.0f * margin);
			if (possible) {
				// corner of this lane and the next one around the rectangle (lanes beyond the fourth repeat; unused)
				uint32_t v = lane & 3u, v_next = (v + 1u) & 3u;
				f3 v0 = shaft_rectangle_corner(light, rectangle, v), v1 = shaft_rectangle_corner(light, rectangle, v_next);
				f3 centroid = (shaft_rectangle_corner(light, rectangle, 0u) + shaft_rectangle_corner(light, rectangle, 2u)) * 0.5f;
				f3 axis = centroid - centre;
				float axis_length = __builtin_sqrtf(dot(axis, axis));
				axis = axis * (1.0f / fmaxf(axis_length, 1.0e-30f));
				float along = dot(v0 - centre, axis);
				float reach = __builtin_sqrtf(dot(v0 - centre, v0 - centre));
				float along_min = quad_min(along);
				float reach_max = quad_max(reach);
				// heights of the light above the patch's own plane (flat patches)
				float above = dot(plane_n, v0) - plane_d;
				float above_min = quad_min(above), above_max = quad_max(above);
				float above_sign = 1.0f;
				// (a flat patch is seen from the front of its plane: a light that lies behind that plane altogether sends it no
				// light - whatever rays rounding or a bent shading normal may still produce are traced as before, without a walk)
				const bool light_behind_patch = (off_plane <= margin) && above_max <= margin;
				if (above_max < 0.0f) { float t = above_min; above_min = -above_max; above_max = -t; above_sign = -1.0f; }
				// side plane v: through the centre and the edge, the rest of the light behind it
				f3 n = cross(v0 - centre, v1 - centre);
				float n_length_squared = dot(n, n);
				bool degenerate = !(n_length_squared > 1.0e-30f);
				n = n * __builtin_amdgcn_rsqf(fmaxf(n_length_squared, 1.0e-30f));
				if (dot(n, centroid - centre) > 0.0f) n = -n;
				// (a polygon seen edge-on or a centre inside it leaves no pyramid: every plane must keep the centroid well inside)
				bool bad = lane < vertex_count && (degenerate || !(dot(n, centroid - centre) < -1.0e-3f * axis_length));
				possible = __ballot(bad) == 0 && axis_length > 4.0f * radius && !light_behind_patch;
				if (lane < vertex_count) {
					s.plane[lane][0] = n.x; s.plane[lane][1] = n.y; s.plane[lane][2] = n.z;
					s.plane[lane][3] = dot(n, centre) + (fabsf(n.x) * half.x + fabsf(n.y) * half.y + fabsf(n.z) * half.z);
					s.light[lane][0] = v0.x; s.light[lane][1] = v0.y; s.light[lane][2] = v0.z;
				}
				// bounding box of patch and polygon
				f3 box_lo = mk3(quad_min(v0.x), quad_min(v0.y), quad_min(v0.z));
				f3 box_hi = mk3(quad_max(v0.x), quad_max(v0.y), quad_max(v0.z));
				if (lane == 0) {
					// far cap: nothing behind the light's plane matters
					f3 nl = plane_normal(light);
					float nl_length = __builtin_sqrtf(dot(nl, nl));
					float sign = (side_min > 0.0f) ? -1.0f : 1.0f;  // positions on the positive side: the far side is the negative one
					f3 far_n = nl * (sign / fmaxf(nl_length, 1.0e-30f));
					float light_offset = dot(far_n, light_vertex(light, 0));
					s.plane[vertex_count][0] = far_n.x; s.plane[vertex_count][1] = far_n.y; s.plane[vertex_count][2] = far_n.z;
					s.plane[vertex_count][3] = light_offset + 8.0f * margin;
					uint32_t planes = vertex_count + 1u;
					// near cap: only if the whole light lies in front of it
					if (along_min > 0.0f) {
						s.plane[planes][0] = -axis.x; s.plane[planes][1] = -axis.y; s.plane[planes][2] = -axis.z;
						s.plane[planes][3] = -(dot(axis, centre) - (fabsf(axis.x) * half.x + fabsf(axis.y) * half.y + fabsf(axis.z) * half.z));
						++planes;
					}
					s.vertex_count = vertex_count;
					s.reach = reach_max + radius;
					// (positions within `margin` of the plane; a ray toward a point `h` above it climbs at least (h - margin) / reach)
					const bool flat = off_plane <= margin && dot(plane_n, plane_n) > 0.5f;
					s.flat_first = (flat && above_min > 2.0f * margin) ? -margin + 1.0e-3f * (above_min - margin) / s.reach : -1.0f;
					s.flat_sign = above_sign;
					// ... and nothing below the plane of a flat patch is inside the shaft (boxes: two margins of slack)
					if (s.flat_first > 3.0f * margin) {
						s.plane[planes][0] = -above_sign * plane_n.x; s.plane[planes][1] = -above_sign * plane_n.y; s.plane[planes][2] = -above_sign * plane_n.z;
						s.plane[planes][3] = -above_sign * plane_d + 2.0f * margin;
						++planes;
					}
					s.plane_count = planes;
					s.box_lo[0] = fminf(box_lo.x, centre.x - half.x); s.box_lo[1] = fminf(box_lo.y, centre.y - half.y); s.box_lo[2] = fminf(box_lo.z, centre.z - half.z);
					s.box_hi[0] = fmaxf(box_hi.x, centre.x + half.x); s.box_hi[1] = fmaxf(box_hi.y, centre.y + half.y); s.box_hi[2] = fmaxf(box_hi.z, centre.z + half.z);
				}
			}
			if (possible) alive |= 1u << k;
			else if (lane == 0) clear[chunk + k] = resting ? resting : kShaftGeometry;
		}
		uint32_t waiting = 0, leaf_count = 0, steps = 0, batches = 0;
		const uint32_t walked = alive;
		if (lane == 0) {
			// every shaft starts at the root
			uint32_t lights = alive;
			while (lights) {
				uint32_t k = (uint32_t) __builtin_ctz(lights);
				lights &= lights - 1u;
				frontier[waiting++] = k << kShaftLightShift;
			}
			failed_lights = 0u;
		}
		if (lane < kShaftLights) list_count[lane] = 0u;
		waiting = (uint32_t) __popc(alive);
		__syncthreads();
		// ---- the walk ----------------------------------------------------------------------------------------------
		const uint32_t step_limit = max_steps + (max_steps / 5u) * waiting;
		uint32_t too_long = 0, queue_full = 0;
		while (alive != 0 && (waiting != 0 || leaf_count != 0)) {
			if (++steps > step_limit) { too_long = alive; alive = 0; break; }
			if (waiting != 0) {
				// the last (up to) sixteen entries of the frontier, four lanes each
				uint32_t take = waiting < 16u ? waiting : 16u;
				bool active = (lane >> 2) < take;
				uint32_t link = kWideEmpty, k = 0;
				bool hit = false;
				if (active) {
					uint32_t entry = frontier[waiting - take + (lane >> 2)];
					k = entry >> kShaftLightShift;
					uint32_t node = entry & ((1u << kShaftLightShift) - 1u);
					// (a light that has failed in the meantime: its entries are dropped as they come up)
					if ((alive >> k) & 1u) {
						const uint32_t* words = (const uint32_t*) (wide_nodes + 4 * (size_t) node);
						uint32_t child = lane & 3u;
						uint32_t qx = words[child], qy = words[4 + child], qz = words[8 + child];
						link = words[12 + child];
						if (link != kWideEmpty) {
							f3 box_lo = mk3(fmaf((float) (qx & 0xFFFFu), cell.x, grid_origin.x), fmaf((float) (qy & 0xFFFFu), cell.y, grid_origin.y), fmaf((float) (qz & 0xFFFFu), cell.z, grid_origin.z));
							f3 box_hi = mk3(fmaf((float) (qx >> 16), cell.x, grid_origin.x), fmaf((float) (qy >> 16), cell.y, grid_origin.y), fmaf((float) (qz >> 16), cell.z, grid_origin.z));
							box_lo = box_lo - mk3(margin, margin, margin);
							box_hi = box_hi + mk3(margin, margin, margin);
							hit = shaft_box_inside(shafts[k], box_lo, box_hi);
						}
					}
				}
				__syncthreads();  // (everyone has read its frontier entry before the entries are overwritten)
				waiting -= take;
				bool to_frontier = hit && !(link & kLeafBit), to_leaves = hit && (link & kLeafBit) != 0;
				uint64_t inner_mask = __ballot(to_frontier), leaf_mask = __ballot(to_leaves);
				uint32_t inner_new = (uint32_t) __popcll((unsigned long long) inner_mask), leaf_new = (uint32_t) __popcll((unsigned long long) leaf_mask);
				if (waiting + inner_new > kShaftFrontier || leaf_count + leaf_new > kShaftLeaves) { queue_full = alive; alive = 0; break; }
				if (to_frontier) frontier[waiting + __builtin_amdgcn_mbcnt_hi((uint32_t) (inner_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) inner_mask, 0u))] = link | (k << kShaftLightShift);
				if (to_leaves) leaves[leaf_count + __builtin_amdgcn_mbcnt_hi((uint32_t) (leaf_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) leaf_mask, 0u))] = (link & ~kLeafBit) | (k << kShaftLightShift);
				waiting += inner_new;
				leaf_count += leaf_new;
				__syncthreads();
			}
			// triangles: as soon as a good part of the wave has one, or nothing else is left
			if (leaf_count >= kShaftLeafBatch || (waiting == 0 && leaf_count != 0)) {
				uint32_t take = leaf_count < 64u ? leaf_count : 64u;
				++batches;
				if (lane < take) {
					uint32_t entry = leaves[leaf_count - take + lane];
					uint32_t k = entry >> kShaftLightShift;
					if ((alive >> k) & 1u) {
						const float4* t = p.bvh.triangles + 3 * (size_t) (entry & ((1u << kShaftLightShift) - 1u));
						float4 a = t[0], bq = t[1], cq = t[2];
						if (!shaft_triangle_harmless(shafts[k], patch, mk3(a.x, a.y, a.z), mk3(bq.x, bq.y, bq.z), mk3(cq.x, cq.y, cq.z), margin)) {
							uint32_t which = __float_as_uint(a.w);
							// onto the pair's list, if there is room left (a triangle cut into several leaves comes up several times:
							// sorted out at the end)
							uint32_t position = list_capacity ? atomicAdd(&list_count[k], 1u) : 0u;
							if (position < list_capacity) {
								list_slot[k][position] = entry & ((1u << kShaftLightShift) - 1u);
								list_triangle[k][position] = which;
							}
							else {
								atomicOr(&failed_lights, 1u << k);
								// (diagnostics: bits 8 ... 31 of the verdict name one triangle that is in the way, if its index fits)
								failed_triangle[k] = which < (1u << 24) ? which << 8 : 0u;
							}
						}
					}
				}
				__syncthreads();
				leaf_count -= take;
				alive &= ~failed_lights;
			}
		}
		__syncthreads();
		// ---- the lists, one lane per entry (a triangle that was cut into several leaves came up
		// several times: the first one counts) -------------------------------------------------------------------------
		static_assert(kShaftListMax <= 16, "the verdict has five bits for the length of a list");
		if (list_capacity) {
			const uint32_t given_up = failed_lights | too_long | queue_full;
			if (lane < kShaftLights) listed_count[lane] = 0u;
			__syncthreads();
			for (uint32_t entry = lane; entry < kShaftLights * kShaftListMax; entry += 64u) {
				const uint32_t k = entry / (kShaftListMax ? kShaftListMax : 1u), j = entry % (kShaftListMax ? kShaftListMax : 1u);
				bool mine = k < chunk_lights && ((walked >> k) & 1u) && !((given_up >> k) & 1u) && j < list_count[k];
				uint32_t before = 0;
				if (mine) {
					for (uint32_t i = 0; i < j; ++i) {
						bool first = true;
						for (uint32_t h = 0; h < i; ++h) first = first && list_triangle[k][h] != list_triangle[k][i];
						before += first ? 1u : 0u;
						mine = mine && list_triangle[k][i] != list_triangle[k][j];
					}
				}
				if (mine) {
					const float4* t = p.bvh.triangles + 3 * (size_t) list_slot[k][j];
					float4 a = t[0], bq = t[1], cq = t[2];
					float4* out = (float4*) (out_lists + (((size_t) b * p.light_count + chunk + k) * kShaftListMax + before) * kShaftListEntry);
					// (the edges as ray_triangle computes them, lbvh.h)
					out[0] = make_float4(a.x, a.y, a.z, a.w);
					out[1] = make_float4(bq.x - a.x, bq.y - a.y, bq.z - a.z, 0.0f);
					out[2] = make_float4(cq.x - a.x, cq.y - a.y, cq.z - a.z, 0.0f);
					atomicAdd(&listed_count[k], 1u);
				}
			}
			__syncthreads();
		}
		if (lane < chunk_lights && ((walked >> lane) & 1u)) {
			uint32_t verdict = kShaftClear;
			uint32_t listed = list_capacity ? listed_count[lane] : 0u;
			if ((failed_lights >> lane) & 1u) verdict = kShaftTriangle | failed_triangle[lane];
			else if ((too_long >> lane) & 1u) verdict = kShaftTooLong;
			else if ((queue_full >> lane) & 1u) verdict = kShaftQueueFull;
			else if (listed != 0u) verdict = kShaftList | (listed << 8);
			clear[chunk + lane] = verdict;
		}
		if (lane == 0 && work_counters && walked) {
			atomicAdd(work_counters, (unsigned long long) steps);
			atomicAdd(work_counters + 1, (unsigned long long) batches);
			atomicAdd(work_counters + 2, (unsigned long long) __popc(walked));
		}
	}
}

}  // namespace vkr
