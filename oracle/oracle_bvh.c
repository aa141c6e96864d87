/* TEST INFRASTRUCTURE — not part of the product.
 *
 * Any-hit ray queries for the CPU oracle.  The reference delegates shadow rays
 * to VK_KHR_ray_query and a driver-built acceleration structure
 * (src/scene.c:142-406, src/shaders/shading_pass.frag.glsl:120-138), so there is
 * no reference arithmetic to restate: parity is unpinned by the reference and
 * the contract is the ray-query semantics only (opaque, terminate on first hit,
 * no face culling, t in [t_min, t_max]).  What IS fixed here, and mirrored by the
 * HIP traversal so that both sides return the same boolean for every ray:
 *   - geometry = triangle soup de-quantised like scene.c:176-187 (multiply, then
 *     add; two roundings, unlike the fused decode of the shading path),
 *   - the ray/triangle test below (Moeller-Trumbore, fp32, no fusing),
 *   - bounding volumes are padded and tested conservatively, so the tree shape
 *     never changes the answer (checked against brute force in the tests).
 * The tree itself is a plain median split; it shares nothing with the LBVH of
 * the product. */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	float lo[3], hi[3];
	/* leaf: count > 0 and first = first triangle slot; inner: count == 0, children first and first + 1 */
	uint32_t first, count;
} bvh_node_t;

typedef struct {
	float* vertices; /* 9 floats per triangle, in original triangle order */
	uint64_t triangle_count;
	uint32_t* order; /* triangle indices in leaf order */
	bvh_node_t* nodes;
	uint32_t node_count;
	float pad;
} bvh_t;

static const float* g_sort_centroids;
static int g_sort_axis;
static int compare_centroid(const void* a, const void* b) {
	float ca = g_sort_centroids[3 * (size_t) *(const uint32_t*) a + g_sort_axis];
	float cb = g_sort_centroids[3 * (size_t) *(const uint32_t*) b + g_sort_axis];
	return (ca > cb) - (ca < cb);
}

static void bounds_of(const bvh_t* b, uint32_t first, uint32_t count, float lo[3], float hi[3]) {
	for (int j = 0; j != 3; ++j) { lo[j] = INFINITY; hi[j] = -INFINITY; }
	for (uint32_t i = first; i != first + count; ++i) {
		const float* t = b->vertices + 9 * (size_t) b->order[i];
		for (int v = 0; v != 3; ++v)
			for (int j = 0; j != 3; ++j) {
				lo[j] = fminf(lo[j], t[3 * v + j]);
				hi[j] = fmaxf(hi[j], t[3 * v + j]);
			}
	}
	for (int j = 0; j != 3; ++j) { lo[j] -= b->pad; hi[j] += b->pad; }
}

static void build_node(bvh_t* b, const float* centroids, uint32_t node, uint32_t first, uint32_t count) {
	bvh_node_t* n = &b->nodes[node];
	bounds_of(b, first, count, n->lo, n->hi);
	if (count <= 4) { n->first = first; n->count = count; return; }
	int axis = 0;
	float ext[3] = {n->hi[0] - n->lo[0], n->hi[1] - n->lo[1], n->hi[2] - n->lo[2]};
	if (ext[1] > ext[axis]) axis = 1;
	if (ext[2] > ext[axis]) axis = 2;
	g_sort_centroids = centroids;
	g_sort_axis = axis;
	qsort(b->order + first, count, sizeof(uint32_t), compare_centroid);
	uint32_t half = count / 2;
	uint32_t child = b->node_count;
	b->node_count += 2;
	n = &b->nodes[node];
	n->first = child;
	n->count = 0;
	build_node(b, centroids, child, first, half);
	build_node(b, centroids, child + 1, first + half, count - half);
}

void* oracle_bvh_build(const uint32_t* q, uint64_t triangle_count, const float factor[3], const float summand[3]) {
	bvh_t* b = (bvh_t*) calloc(1, sizeof(bvh_t));
	b->triangle_count = triangle_count;
	b->vertices = (float*) malloc(sizeof(float) * 9 * triangle_count);
	float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (uint64_t i = 0; i != triangle_count * 3; ++i) {
		uint32_t q0 = q[2 * i], q1 = q[2 * i + 1];
		float p[3] = {
			(float) (q0 & 0x1FFFFF),
			(float) (((q0 & 0xFFE00000u) >> 21) | ((q1 & 0x3FF) << 11)),
			(float) ((q1 & 0x7FFFFC00u) >> 10)};
		for (int j = 0; j != 3; ++j) {
			float w = p[j] * factor[j] + summand[j];
			b->vertices[3 * i + j] = w;
			lo[j] = fminf(lo[j], w);
			hi[j] = fmaxf(hi[j], w);
		}
	}
	float extent = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
	for (int j = 0; j != 3; ++j) extent = fmaxf(extent, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
	/* sixteen times the rounding error of the slab and triangle tests (about 2^-23 of
	 * the largest coordinate), far below t_min */
	b->pad = 2.0e-6f * extent;
	float* centroids = (float*) malloc(sizeof(float) * 3 * triangle_count);
	b->order = (uint32_t*) malloc(sizeof(uint32_t) * triangle_count);
	for (uint64_t t = 0; t != triangle_count; ++t) {
		b->order[t] = (uint32_t) t;
		for (int j = 0; j != 3; ++j)
			centroids[3 * t + j] = (b->vertices[9 * t + j] + b->vertices[9 * t + 3 + j] + b->vertices[9 * t + 6 + j]) * (1.0f / 3.0f);
	}
	b->nodes = (bvh_node_t*) malloc(sizeof(bvh_node_t) * (2 * triangle_count + 1));
	b->node_count = 1;
	build_node(b, centroids, 0, 0, (uint32_t) triangle_count);
	free(centroids);
	return b;
}

void oracle_bvh_destroy(void* handle) {
	bvh_t* b = (bvh_t*) handle;
	if (!b) return;
	free(b->vertices);
	free(b->order);
	free(b->nodes);
	free(b);
}

/* The shared ray/triangle predicate: Moeller-Trumbore with the division removed
 * (barycentrics and distance are compared in their det-scaled form, mirrored for
 * det < 0).  Comparisons are written so that NaNs miss. */
static int ray_triangle(const float* t, const float o[3], const float d[3], float t_min, float t_max) {
	float e1[3] = {t[3] - t[0], t[4] - t[1], t[5] - t[2]};
	float e2[3] = {t[6] - t[0], t[7] - t[1], t[8] - t[2]};
	float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
	float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
	if (!(det != 0.0f)) return 0;
	float sign = (det < 0.0f) ? -1.0f : 1.0f;
	float adet = det * sign;
	float s[3] = {o[0] - t[0], o[1] - t[1], o[2] - t[2]};
	float U = ((s[0] * p[0] + s[1] * p[1]) + s[2] * p[2]) * sign;
	if (!(U >= 0.0f && U <= adet)) return 0;
	float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
	float V = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * sign;
	if (!(V >= 0.0f && U + V <= adet)) return 0;
	float T = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * sign;
	return T >= t_min * adet && T <= t_max * adet;
}

static int ray_box(const bvh_node_t* n, const float o[3], const float inv[3], float t_min, float t_max) {
	float near = t_min, far = t_max;
	for (int j = 0; j != 3; ++j) {
		float t0 = (n->lo[j] - o[j]) * inv[j], t1 = (n->hi[j] - o[j]) * inv[j];
		near = fmaxf(near, fminf(t0, t1));
		far = fminf(far, fmaxf(t0, t1));
	}
	return near <= far * 1.0000004f;
}

int oracle_bvh_any_hit(const void* handle, const float o[3], const float d[3], float t_min, float t_max, int brute_force) {
	const bvh_t* b = (const bvh_t*) handle;
	if (!b) return 0;
	if (!(t_max >= t_min)) return 0;
	if (brute_force) {
		for (uint64_t t = 0; t != b->triangle_count; ++t)
			if (ray_triangle(b->vertices + 9 * t, o, d, t_min, t_max)) return 1;
		return 0;
	}
	float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
	uint32_t stack[128];
	int top = 0;
	stack[top++] = 0;
	while (top) {
		const bvh_node_t* n = &b->nodes[stack[--top]];
		if (!ray_box(n, o, inv, t_min, t_max)) continue;
		if (n->count) {
			for (uint32_t i = n->first; i != n->first + n->count; ++i)
				if (ray_triangle(b->vertices + 9 * (size_t) b->order[i], o, d, t_min, t_max)) return 1;
		}
		else if (top + 2 <= 128) {
			stack[top++] = n->first;
			stack[top++] = n->first + 1;
		}
	}
	return 0;
}

/* Closest hit among front-facing triangles (det > 0, i.e. the normal
 * (v1-v0)x(v2-v0) faces the ray origin), t in [t_min, t_max]; ties go to the
 * smaller primitive index.  Checker for the product's primary-visibility kernel,
 * which stands in for the rasterised visibility pass of the reference
 * (src/shaders/visibility_pass.vert.glsl:27-33, back-face culling and depth test
 * at src/main.c:501-507,537-542). */
static int ray_triangle_front(const float* t, const float o[3], const float d[3], float t_min, float t_max, float* out_dist) {
	float e1[3] = {t[3] - t[0], t[4] - t[1], t[5] - t[2]};
	float e2[3] = {t[6] - t[0], t[7] - t[1], t[8] - t[2]};
	float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
	float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
	if (!(det > 0.0f)) return 0;
	float s[3] = {o[0] - t[0], o[1] - t[1], o[2] - t[2]};
	float U = ((s[0] * p[0] + s[1] * p[1]) + s[2] * p[2]) * 1.0f;
	if (!(U >= 0.0f && U <= det)) return 0;
	float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
	float V = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * 1.0f;
	if (!(V >= 0.0f && U + V <= det)) return 0;
	float T = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * 1.0f;
	if (!(T >= t_min * det && T <= t_max * det)) return 0;
	*out_dist = T / det;
	return 1;
}

uint32_t oracle_bvh_closest_front_hit(const void* handle, const float o[3], const float d[3], float t_min, float t_max) {
	const bvh_t* b = (const bvh_t*) handle;
	uint32_t best = 0xFFFFFFFFu;
	if (!b) return best;
	float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
	uint32_t stack[128];
	int top = 0;
	stack[top++] = 0;
	while (top) {
		const bvh_node_t* n = &b->nodes[stack[--top]];
		if (!ray_box(n, o, inv, t_min, t_max)) continue;
		if (n->count) {
			for (uint32_t i = n->first; i != n->first + n->count; ++i) {
				uint32_t primitive = b->order[i];
				float dist;
				if (ray_triangle_front(b->vertices + 9 * (size_t) primitive, o, d, t_min, t_max, &dist)) {
					if (dist < t_max || primitive < best) { t_max = dist; best = primitive; }
				}
			}
		}
		else if (top + 2 <= 128) {
			stack[top++] = n->first;
			stack[top++] = n->first + 1;
		}
	}
	return best;
}
