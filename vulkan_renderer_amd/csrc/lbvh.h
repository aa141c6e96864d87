// LBVH layout and stackless ray traversal (device code, gfx950).
//
// The reference has no BVH code: shadow rays go through VK_KHR_ray_query against
// a driver-built acceleration structure (reference src/scene.c:142-406,
// src/shaders/shading_pass.frag.glsl:120-138).  The contract kept here is the
// ray-query semantics: opaque geometry, first hit terminates, no face culling,
// t in [t_min, t_max], triangle soup de-quantised like scene.c:176-187.
//
// Two layouts are built from the same binary tree.  The wavefront shadow-ray kernel walks the
// four-wide one (further down: "wide BVH"), everything else (primary visibility, rays traced
// inside the shading kernel) the binary one:
//
// Layout in HBM ("threaded" BVH): all 2n-1 nodes of the binary tree, inner nodes and
// leaves alike, stored in depth-first order as 16 bytes each:
//     uint4 = (lo.x | hi.x << 16, lo.y | hi.y << 16, lo.z | hi.z << 16, link)
// Box coordinates are 15-bit positions (kGridMax, see "wide BVH") on a uniform grid over the
// (padded) scene box, rounded outwards, so the quantised box contains the fp32 box.  The left child of an
// inner node is the next node; for an inner node `link` is the index of the node that
// follows its whole subtree, for a leaf it is 0x80000000 | triangle slot (the node after
// a leaf is always the next one).  A ray walks the array with a single cursor: hit ->
// next node (testing the triangle first if it is a leaf), miss -> link.  No stack, hence
// no scratch memory and no LDS: the only per-ray state is the cursor.  One 16-byte load
// per visit matters: the traversal is bound by the L1/TA data path (profiles/).
// The ray is moved into grid space once ((o - origin) * inverse_cell, d * inverse_cell),
// which leaves the ray parameter t unchanged.  Triangles are three float4 per slot in
// leaf order, w of vertex 0 carries the original primitive index.
#pragma once
#include "device_math.h"

namespace vkr {

// build-time node of the binary radix tree (Karras 2012); not used for traversal
struct alignas(16) bvh_build_node {
	// x = child 0, y = child 1 (bit 31: leaf slot), z = parent, w = first leaf of the range
	uint4 links;
	// x = last leaf of the range, y = number of leaves under child 0
	uint2 range;
	uint2 pad;
};

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoLeaf = 0xFFFFFFFFu;
// outward rounding margin of the quantised boxes in cells: covers the rounding of the
// grid transform of the ray and of the slab arithmetic (a few 1e-3 cells each)
constexpr float kGridMargin = 0.05f;

struct bvh_view {
	const uint4* nodes;       // 16 bytes per node, depth-first order
	const float4* triangles;  // 3 float4 per leaf slot
	uint32_t node_count;      // 2 * triangle_count - 1
	f3 grid_origin, grid_inverse_cell;
};

#if VKR_FAST_MATH
// the triangle test below with every product rounded before it is added (see ray_triangle_edges)
VKR_DEV float unfused_dot(f3 a, f3 b) { return (opaque(a.x * b.x) + opaque(a.y * b.y)) + opaque(a.z * b.z); }
VKR_DEV f3 unfused_cross(f3 a, f3 b) { return mk3(opaque(a.y * b.z) - opaque(a.z * b.y), opaque(a.z * b.x) - opaque(a.x * b.z), opaque(a.x * b.y) - opaque(a.y * b.x)); }
template <bool CULL_BACK>
VKR_DEV bool ray_triangle_edges_unfused(f3 p0, f3 e1, f3 e2, f3 o, f3 d, float t_min, float t_max, float& dist) {
	f3 p = unfused_cross(d, e2);
	float det = unfused_dot(e1, p);
	if (CULL_BACK ? !(det > 0.0f) : !(det != 0.0f)) return false;
	float sign = (det < 0.0f) ? -1.0f : 1.0f;
	float adet = det * sign;
	f3 s = mk3(o.x - p0.x, o.y - p0.y, o.z - p0.z);
	float U = unfused_dot(s, p) * sign;
	if (!(U >= 0.0f && U <= adet)) return false;
	f3 q = unfused_cross(s, e1);
	float V = unfused_dot(d, q) * sign;
	if (!(V >= 0.0f && opaque(U) + V <= adet)) return false;
	float T = unfused_dot(e2, q) * sign;
	if (!(T >= opaque(t_min * adet) && T <= opaque(t_max * adet))) return false;
	if (CULL_BACK) dist = T * __builtin_amdgcn_rcpf(adet);
	return true;
}
#endif

// Moeller-Trumbore without the division, fp32, same operation order as
// oracle/oracle_bvh.c ray_triangle: with det = e1 . (d x e2) the barycentrics u, v and
// the distance t are compared in their det-scaled form (U = u det, V = v det, T = t det)
// and all comparisons are mirrored for det < 0.  No face culling unless CULL_BACK
// (then triangles whose normal (v1-v0)x(v2-v0) points along the ray are rejected).
// `dist` (only needed by the closest-hit query) is T / det.
// (the test proper, on a vertex and the two edges that leave it - what an occluder list of light_shafts.h stores)
// (never contracted, whatever the translation unit's flags say: the shading kernels decide the rays of an occluder list with
// this test, and their verdict must be the tracing kernel's in every arithmetic mode - the products are made opaque, because
// -ffp-contract=fast fuses in the back end whatever a pragma says, like difference_of_products in polygon_sampling.h)
template <bool CULL_BACK>
VKR_DEV bool ray_triangle_edges(f3 p0, f3 e1, f3 e2, f3 o, f3 d, float t_min, float t_max, float& dist) {
#if VKR_FAST_MATH
	return ray_triangle_edges_unfused<CULL_BACK>(p0, e1, e2, o, d, t_min, t_max, dist);
#endif
	f3 p = cross(d, e2);
	float det = dot(e1, p);
	if (CULL_BACK ? !(det > 0.0f) : !(det != 0.0f)) return false;
	float sign = (det < 0.0f) ? -1.0f : 1.0f;
	float adet = det * sign;
	f3 s = mk3(o.x - p0.x, o.y - p0.y, o.z - p0.z);
	float U = dot(s, p) * sign;
	if (!(U >= 0.0f && U <= adet)) return false;
	f3 q = cross(s, e1);
	float V = dot(d, q) * sign;
	if (!(V >= 0.0f && U + V <= adet)) return false;
	float T = dot(e2, q) * sign;
	if (!(T >= t_min * adet && T <= t_max * adet)) return false;
	if (CULL_BACK) dist = T / adet;
	return true;
}
template <bool CULL_BACK>
VKR_DEV bool ray_triangle(float4 p0, float4 p1, float4 p2, f3 o, f3 d, float t_min, float t_max, float& dist) {
	f3 e1 = mk3(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
	f3 e2 = mk3(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z);
	return ray_triangle_edges<CULL_BACK>(mk3(p0.x, p0.y, p0.z), e1, e2, o, d, t_min, t_max, dist);
}

// A ray in the grid space of the quantised boxes: t = plane * inv + shift
struct grid_ray {
	f3 inv, shift;
};

VKR_DEV grid_ray make_grid_ray(const bvh_view& bvh, f3 o, f3 d) {
	f3 og = mk3((o.x - bvh.grid_origin.x) * bvh.grid_inverse_cell.x, (o.y - bvh.grid_origin.y) * bvh.grid_inverse_cell.y, (o.z - bvh.grid_origin.z) * bvh.grid_inverse_cell.z);
	f3 dg = mk3(d.x * bvh.grid_inverse_cell.x, d.y * bvh.grid_inverse_cell.y, d.z * bvh.grid_inverse_cell.z);
	grid_ray r;
	// approximate reciprocals are fine: the boxes are rounded outwards by kGridMargin
	r.inv = mk3(__builtin_amdgcn_rcpf(dg.x), __builtin_amdgcn_rcpf(dg.y), __builtin_amdgcn_rcpf(dg.z));
	r.shift = mk3(-og.x * r.inv.x, -og.y * r.inv.y, -og.z * r.inv.z);
	return r;
}

// Conservative slab test against a quantised node.  (For axis-parallel rays inf - inf
// gives NaN, which min/max ignore: that slab then never culls, which is conservative.)
VKR_DEV bool ray_box(uint4 n, const grid_ray& r, float t_min, float t_max) {
	float x0 = fmaf((float) (n.x & 0xFFFFu), r.inv.x, r.shift.x), x1 = fmaf((float) (n.x >> 16), r.inv.x, r.shift.x);
	float y0 = fmaf((float) (n.y & 0xFFFFu), r.inv.y, r.shift.y), y1 = fmaf((float) (n.y >> 16), r.inv.y, r.shift.y);
	float z0 = fmaf((float) (n.z & 0xFFFFu), r.inv.z, r.shift.z), z1 = fmaf((float) (n.z >> 16), r.inv.z, r.shift.z);
	float near = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
	float far = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
	return near <= far * 1.0000004f;
}

// Any-hit query (shadow rays).  Returns true iff some triangle intersects the
// ray within [t_min, t_max].
VKR_DEV bool any_hit(const bvh_view& bvh, f3 o, f3 d, float t_min, float t_max) {
	if (!(t_max >= t_min)) return false;
	grid_ray r = make_grid_ray(bvh, o, d);
	uint32_t node = 0;
	const uint32_t end = bvh.node_count;
	float dist;
	while (node < end) {
		uint4 n = bvh.nodes[node];
		bool is_leaf = (n.w & kLeafBit) != 0;
		bool hit = ray_box(n, r, t_min, t_max);
		if (hit && is_leaf) {
			const float4* t = bvh.triangles + 3 * (size_t) (n.w & ~kLeafBit);
			if (ray_triangle<false>(t[0], t[1], t[2], o, d, t_min, t_max, dist)) return true;
		}
		// inner node that was hit: descend (the left child is the next node); a leaf is
		// followed by the next node as well; otherwise leave the subtree
		node = (hit || is_leaf) ? node + 1 : n.w;
	}
	return false;
}

// ---- wide BVH -------------------------------------------------------------------------
// The binary tree collapsed to four children per node (lbvh_build.hip k_collapse_level: a node's
// two children, then twice the child with the largest box replaced by its own children).  One node
// is 64 bytes = one half cache line, fetched with four dwordx4 loads:
//     uint4 x = (lo.x | hi.x << 16) of children 0..3      (same grid as the binary nodes)
//     uint4 y, uint4 z likewise
//     uint4 link: child is an inner node -> its index; a triangle -> kLeafBit | triangle slot;
//                 absent -> kWideEmpty (and the box kWideEmptyBox, which no ray hits)
// The children of a node are stored next to each other and levels one after the other, so the
// first nodes of the array are the top of the tree.  A visit tests four boxes with one dependent
// fetch, which shortens the chain of dependent fetches per ray four- to fivefold against the
// binary walk (profiles/).  Hit children beyond the first go to a per-lane stack: 16 entries in
// LDS ([entry][thread], conflict-free), deeper ones - the build computes the worst case,
// acceleration_structure_t.wide_stack_need - spill to global memory.  Any-hit queries are
// order-independent, so the result (a boolean) is the same as with any other tree.
constexpr uint32_t kWideEmpty = 0xFFFFFFFFu;
#ifndef VKR_WIDE_STACK_LDS
#define VKR_WIDE_STACK_LDS 16
#endif
constexpr uint32_t kWideStackLds = VKR_WIDE_STACK_LDS;   // stack entries per lane that live in LDS
constexpr uint32_t kWideStackMax = 128;  // deepest stack the kernels are prepared for (else: binary walk)
// Grid coordinates are 15-bit (0 ... kGridMax): a plane q then sits in bits 8 ... 22 of the float
// 2^15 + q, i.e. one v_perm_b32 turns a packed pair (lo | hi << 16) into that float - and the byte
// selector chooses lo or hi per lane, so the near and the far plane of a ray's octant come out of
// the same instruction (wide_ray below).  A 16-bit grid would need a conversion and a min / max
// per plane: 93 instead of 62 clocks of VALU issue per box (profiles/tools/valu_rate.hip).
constexpr float kGridMax = 32767.0f;
// box of an absent child of a wide node: lo > hi on every axis, never hit
constexpr uint32_t kWideEmptyBox = 0x00007FFFu;

// A ray prepared for the four-wide nodes: t = (2^15 + q) * inv + shift with the 2^15 folded into
// shift (2^15 * inv is at most 2^-9 cells off after rounding; the boxes carry kGridMargin = 0.05),
// and per axis the v_perm_b32 selectors of the plane the ray enters through and the one it leaves
// through.  Selector bytes: 0x0C -> 0x00, 4 ... 7 -> bytes of the packed pair, 3 -> 0x47.
struct wide_ray {
	f3 inv, shift;
	uint32_t near_x, near_y, near_z;
};
constexpr uint32_t kPermLo = 0x0305040Cu, kPermHi = 0x0307060Cu, kPermFlip = kPermLo ^ kPermHi, kPermMagic = 0x47000000u;

VKR_DEV wide_ray make_wide_ray(const grid_ray& r) {
	wide_ray w;
	w.inv = r.inv;
	w.shift = mk3(fmaf(-32768.0f, r.inv.x, r.shift.x), fmaf(-32768.0f, r.inv.y, r.shift.y), fmaf(-32768.0f, r.inv.z, r.shift.z));
	// the sign bit decides (inv = -inf for d = -0 as well)
	w.near_x = (__float_as_uint(r.inv.x) >> 31) ? kPermHi : kPermLo;
	w.near_y = (__float_as_uint(r.inv.y) >> 31) ? kPermHi : kPermLo;
	w.near_z = (__float_as_uint(r.inv.z) >> 31) ? kPermHi : kPermLo;
	return w;
}

// Conservative slab test of one child box of a wide node.  A NaN (0 * inf for a ray inside a slab
// it runs parallel to) is ignored by min / max, which is conservative.
VKR_DEV bool wide_ray_box(uint32_t qx, uint32_t qy, uint32_t qz, const wide_ray& r, float t_min, float t_max) {
	float nx = fmaf(__uint_as_float(__builtin_amdgcn_perm(qx, kPermMagic, r.near_x)), r.inv.x, r.shift.x);
	float fx = fmaf(__uint_as_float(__builtin_amdgcn_perm(qx, kPermMagic, r.near_x ^ kPermFlip)), r.inv.x, r.shift.x);
	float ny = fmaf(__uint_as_float(__builtin_amdgcn_perm(qy, kPermMagic, r.near_y)), r.inv.y, r.shift.y);
	float fy = fmaf(__uint_as_float(__builtin_amdgcn_perm(qy, kPermMagic, r.near_y ^ kPermFlip)), r.inv.y, r.shift.y);
	float nz = fmaf(__uint_as_float(__builtin_amdgcn_perm(qz, kPermMagic, r.near_z)), r.inv.z, r.shift.z);
	float fz = fmaf(__uint_as_float(__builtin_amdgcn_perm(qz, kPermMagic, r.near_z ^ kPermFlip)), r.inv.z, r.shift.z);
	float near = fmaxf(fmaxf(nx, ny), fmaxf(nz, t_min));
	float far = fminf(fminf(fx, fy), fminf(fz, t_max));
	return near <= far;
}

// Closest hit with back-face culling (primary visibility).  Returns the original
// primitive index or 0xFFFFFFFF.
VKR_DEV uint32_t closest_front_hit(const bvh_view& bvh, f3 o, f3 d, float t_min, float t_max) {
	grid_ray r = make_grid_ray(bvh, o, d);
	uint32_t node = 0;
	const uint32_t end = bvh.node_count;
	uint32_t best = 0xFFFFFFFFu;
	float dist;
	while (node < end) {
		uint4 n = bvh.nodes[node];
		bool is_leaf = (n.w & kLeafBit) != 0;
		bool hit = ray_box(n, r, t_min, t_max);
		if (hit && is_leaf) {
			const float4* t = bvh.triangles + 3 * (size_t) (n.w & ~kLeafBit);
			float4 p0 = t[0];
			if (ray_triangle<true>(p0, t[1], t[2], o, d, t_min, t_max, dist)) {
				uint32_t primitive = __float_as_uint(p0.w);
				// depth test LESS; ties go to the smaller primitive index for determinism
				if (dist < t_max || primitive < best) { t_max = dist; best = primitive; }
			}
		}
		node = (hit || is_leaf) ? node + 1 : n.w;
	}
	return best;
}

}  // namespace vkr
