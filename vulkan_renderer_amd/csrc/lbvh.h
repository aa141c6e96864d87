// LBVH node layout and ray traversal (device code, gfx950).
//
// The reference has no BVH code: shadow rays go through VK_KHR_ray_query against
// a driver-built acceleration structure (reference src/scene.c:142-406,
// src/shaders/shading_pass.frag.glsl:120-138).  The contract kept here is the
// ray-query semantics: opaque geometry, first hit terminates, no face culling,
// t in [t_min, t_max], triangle soup de-quantised like scene.c:176-187.
//
// Layout in HBM: one 64-byte node per inner node holding BOTH child boxes, so a
// visit is four coalescable 16-byte loads; leaves are single triangles stored as
// three float4 in Morton order (w of vertex 0 carries the original primitive
// index).  Child links with bit 31 set point at triangle slots.
#pragma once
#include "device_math.h"

namespace vkr {

struct alignas(16) bvh_node {
	// a = (lo0.xyz, hi0.x)  b = (hi0.yz, lo1.xy)  c = (lo1.z, hi1.xyz)
	float4 a, b, c;
	// x = child 0, y = child 1 (bit 31: leaf), z = parent, w = unused
	uint4 links;
};

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr int kTraversalStack = 64;

struct bvh_view {
	const bvh_node* nodes;
	const float4* triangles;  // 3 per leaf slot
	uint32_t root;            // inner node index, or kLeafBit | slot for 1 triangle
};

// Moeller-Trumbore, fp32, same operation order as oracle/oracle_bvh.c ray_triangle.
// CULL_BACK additionally rejects triangles whose normal (v1-v0)x(v2-v0) points
// along the ray (det <= 0).  Returns the distance through `dist`.
template <bool CULL_BACK>
VKR_DEV bool ray_triangle(float4 p0, float4 p1, float4 p2, f3 o, f3 d, float t_min, float t_max, float& dist) {
	f3 e1 = mk3(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
	f3 e2 = mk3(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z);
	f3 p = cross(d, e2);
	float det = dot(e1, p);
	if (CULL_BACK ? !(det > 0.0f) : !(det != 0.0f)) return false;
	float inv = rcp(det);
	f3 s = mk3(o.x - p0.x, o.y - p0.y, o.z - p0.z);
	float u = dot(s, p) * inv;
	if (!(u >= 0.0f && u <= 1.0f)) return false;
	f3 q = cross(s, e1);
	float v = dot(d, q) * inv;
	if (!(v >= 0.0f && u + v <= 1.0f)) return false;
	dist = dot(e2, q) * inv;
	return dist >= t_min && dist <= t_max;
}

// Conservative slab test; boxes are padded at build time, so approximate
// reciprocals are fine here in every arithmetic mode.
VKR_DEV bool ray_box(f3 lo, f3 hi, f3 o, f3 inv, float t_min, float t_max) {
	float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
	float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
	float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
	float near = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
	float far = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
	return near <= far * 1.0000004f;
}

// Any-hit query (shadow rays).  Returns true iff some triangle intersects the
// ray within [t_min, t_max].
VKR_DEV bool any_hit(const bvh_view& bvh, f3 o, f3 d, float t_min, float t_max) {
	if (!(t_max >= t_min)) return false;
	f3 inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
	uint32_t stack[kTraversalStack];
	int top = 0;
	uint32_t current = bvh.root;
	float dist;
	while (true) {
		if (current & kLeafBit) {
			const float4* t = bvh.triangles + 3 * (size_t) (current & ~kLeafBit);
			if (ray_triangle<false>(t[0], t[1], t[2], o, d, t_min, t_max, dist)) return true;
			if (top == 0) return false;
			current = stack[--top];
			continue;
		}
		const bvh_node& n = bvh.nodes[current];
		float4 a = n.a, b = n.b, c = n.c;
		uint4 links = n.links;
		bool hit0 = ray_box(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), o, inv, t_min, t_max);
		bool hit1 = ray_box(mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), o, inv, t_min, t_max);
		if (hit0 && hit1) {
			if (top < kTraversalStack) stack[top++] = links.y;
			current = links.x;
		}
		else if (hit0) current = links.x;
		else if (hit1) current = links.y;
		else {
			if (top == 0) return false;
			current = stack[--top];
		}
	}
}

// Closest hit with back-face culling (primary visibility).  Returns the original
// primitive index or 0xFFFFFFFF.
VKR_DEV uint32_t closest_front_hit(const bvh_view& bvh, f3 o, f3 d, float t_min, float t_max) {
	f3 inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
	uint32_t stack[kTraversalStack];
	int top = 0;
	uint32_t current = bvh.root;
	uint32_t best = 0xFFFFFFFFu;
	float dist;
	while (true) {
		if (current & kLeafBit) {
			const float4* t = bvh.triangles + 3 * (size_t) (current & ~kLeafBit);
			float4 p0 = t[0];
			if (ray_triangle<true>(p0, t[1], t[2], o, d, t_min, t_max, dist)) {
				uint32_t primitive = __float_as_uint(p0.w);
				// depth test LESS; ties go to the smaller primitive index for determinism
				if (dist < t_max || primitive < best) { t_max = dist; best = primitive; }
			}
			if (top == 0) return best;
			current = stack[--top];
			continue;
		}
		const bvh_node& n = bvh.nodes[current];
		float4 a = n.a, b = n.b, c = n.c;
		uint4 links = n.links;
		bool hit0 = ray_box(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), o, inv, t_min, t_max);
		bool hit1 = ray_box(mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), o, inv, t_min, t_max);
		if (hit0 && hit1) {
			if (top < kTraversalStack) stack[top++] = links.y;
			current = links.x;
		}
		else if (hit0) current = links.x;
		else if (hit1) current = links.y;
		else {
			if (top == 0) return best;
			current = stack[--top];
		}
	}
}

}  // namespace vkr
