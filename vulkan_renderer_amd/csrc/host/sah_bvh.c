/* Top-down binned-SAH builder for the threaded BVH (host side, C99).
 *
 * The reference asks the driver for a PREFER_FAST_TRACE acceleration structure once
 * per scene (reference src/scene.c:254-262); the equivalent here is a surface-area-
 * heuristic build on the host when the scene is loaded.  It writes the intermediate layout
 * that every builder of lbvh_build.hip produces: 2n-1 fp32 nodes of 32 bytes (lo.xyz, hi.xyz,
 * skip, leaf slot) in depth-first order, one triangle per leaf, triangles as three float4 in
 * leaf order.  That is NOT what the kernels walk: the device code quantises these nodes
 * afterwards into the traversal layouts of csrc/lbvh.h (mandatory second step).
 *
 * Why it matters: in scenes with a large floor the Morton-code tree mixes floor and
 * objects down to small cells, so every ray that leaves the floor descends ten
 * levels before it is outside all boxes.  The SAH separates such layers near the
 * root (measured: node visits per shadow ray, see profiles/). */
#include "vkr_internal.h"
#include <float.h>

#define SAH_BIN_COUNT 16

typedef struct {
	float lo[3], hi[3];
} box_t;

typedef struct {
	/* per input triangle */
	const float* vertices; /* 9 floats */
	box_t* bounds;
	float* centroids; /* 3 floats */
	/* permutation being partitioned; position in it = leaf slot */
	uint32_t* order;
	/* outputs */
	float* nodes;     /* 8 floats per node */
	float* triangles; /* 12 floats per leaf slot */
	float pad;
} sah_builder_t;

static inline void box_reset(box_t* b) {
	for (int j = 0; j != 3; ++j) { b->lo[j] = FLT_MAX; b->hi[j] = -FLT_MAX; }
}

static inline void box_merge(box_t* b, const box_t* other) {
	for (int j = 0; j != 3; ++j) {
		if (other->lo[j] < b->lo[j]) b->lo[j] = other->lo[j];
		if (other->hi[j] > b->hi[j]) b->hi[j] = other->hi[j];
	}
}

static inline float box_half_area(const box_t* b) {
	float x = b->hi[0] - b->lo[0], y = b->hi[1] - b->lo[1], z = b->hi[2] - b->lo[2];
	return x * y + y * z + z * x;
}

static void write_node(sah_builder_t* b, uint32_t position, const box_t* box, uint32_t skip, uint32_t leaf) {
	float* n = b->nodes + 8 * (size_t) position;
	n[0] = box->lo[0] - b->pad; n[1] = box->lo[1] - b->pad; n[2] = box->lo[2] - b->pad;
	n[3] = box->hi[0] + b->pad; n[4] = box->hi[1] + b->pad; n[5] = box->hi[2] + b->pad;
	memcpy(n + 6, &skip, sizeof(uint32_t));
	memcpy(n + 7, &leaf, sizeof(uint32_t));
}

/* Builds the subtree over order[first, first + count) at node `position`. */
static void build_range(sah_builder_t* b, uint32_t first, uint32_t count, uint32_t position) {
	for (;;) {
		box_t box, centroid_box;
		box_reset(&box);
		box_reset(&centroid_box);
		for (uint32_t i = first; i != first + count; ++i) {
			uint32_t t = b->order[i];
			box_merge(&box, &b->bounds[t]);
			const float* c = b->centroids + 3 * (size_t) t;
			for (int j = 0; j != 3; ++j) {
				if (c[j] < centroid_box.lo[j]) centroid_box.lo[j] = c[j];
				if (c[j] > centroid_box.hi[j]) centroid_box.hi[j] = c[j];
			}
		}
		if (count == 1) {
			uint32_t t = b->order[first];
			write_node(b, position, &box, position + 1, first);
			float* out = b->triangles + 12 * (size_t) first;
			for (int v = 0; v != 3; ++v) {
				memcpy(out + 4 * v, b->vertices + 9 * (size_t) t + 3 * v, sizeof(float) * 3);
				out[4 * v + 3] = 0.0f;
			}
			memcpy(out + 3, &t, sizeof(uint32_t));
			return;
		}
		write_node(b, position, &box, position + 2 * count - 1, 0xFFFFFFFFu);
		/* bin the centroids along all three axes at once */
		box_t bins[3][SAH_BIN_COUNT];
		uint32_t counts[3][SAH_BIN_COUNT];
		float scale[3];
		for (int j = 0; j != 3; ++j) {
			float width = centroid_box.hi[j] - centroid_box.lo[j];
			scale[j] = width > 0.0f ? (float) SAH_BIN_COUNT * (1.0f - 1.0e-6f) / width : 0.0f;
			for (int k = 0; k != SAH_BIN_COUNT; ++k) { box_reset(&bins[j][k]); counts[j][k] = 0; }
		}
		for (uint32_t i = first; i != first + count; ++i) {
			uint32_t t = b->order[i];
			const float* c = b->centroids + 3 * (size_t) t;
			for (int j = 0; j != 3; ++j) {
				int k = (int) ((c[j] - centroid_box.lo[j]) * scale[j]);
				if (k < 0) k = 0;
				if (k >= SAH_BIN_COUNT) k = SAH_BIN_COUNT - 1;
				box_merge(&bins[j][k], &b->bounds[t]);
				++counts[j][k];
			}
		}
		/* cheapest of the 3 x 15 splits: area(left) * count(left) + area(right) * count(right) */
		float best_cost = FLT_MAX;
		int best_axis = -1, best_split = 0;
		for (int j = 0; j != 3; ++j) {
			if (!(scale[j] > 0.0f)) continue;
			float right_area[SAH_BIN_COUNT];
			uint32_t right_count[SAH_BIN_COUNT];
			box_t sweep;
			box_reset(&sweep);
			uint32_t n = 0;
			for (int k = SAH_BIN_COUNT - 1; k > 0; --k) {
				if (counts[j][k]) box_merge(&sweep, &bins[j][k]);
				n += counts[j][k];
				right_area[k] = n ? box_half_area(&sweep) : 0.0f;
				right_count[k] = n;
			}
			box_reset(&sweep);
			n = 0;
			for (int k = 0; k != SAH_BIN_COUNT - 1; ++k) {
				if (counts[j][k]) box_merge(&sweep, &bins[j][k]);
				n += counts[j][k];
				if (n == 0 || right_count[k + 1] == 0) continue;
				float cost = box_half_area(&sweep) * (float) n + right_area[k + 1] * (float) right_count[k + 1];
				if (cost < best_cost) { best_cost = cost; best_axis = j; best_split = k; }
			}
		}
		uint32_t left_count;
		if (best_axis < 0) {
			/* all centroids coincide: any split is as good as another */
			left_count = count / 2;
		}
		else {
			/* partition in place: bins <= best_split go left */
			uint32_t i = first, k = first + count;
			float lo = centroid_box.lo[best_axis], s = scale[best_axis];
			while (i < k) {
				uint32_t t = b->order[i];
				int bin = (int) ((b->centroids[3 * (size_t) t + best_axis] - lo) * s);
				if (bin < 0) bin = 0;
				if (bin >= SAH_BIN_COUNT) bin = SAH_BIN_COUNT - 1;
				if (bin <= best_split) ++i;
				else { --k; b->order[i] = b->order[k]; b->order[k] = t; }
			}
			left_count = i - first;
			if (left_count == 0 || left_count == count) left_count = count / 2;
		}
		/* recurse into the smaller side, loop on the larger one (bounded stack depth) */
		uint32_t right_count_total = count - left_count;
		if (left_count <= right_count_total) {
			build_range(b, first, left_count, position + 1);
			first += left_count; position += 2 * left_count; count = right_count_total;
		}
		else {
			build_range(b, first + left_count, right_count_total, position + 2 * left_count);
			position += 1; count = left_count;
		}
	}
}

int vkr_build_sah_bvh_host(const mesh_t* mesh, float pad, float** out_nodes, float** out_triangles, uint32_t* out_node_count) {
	*out_nodes = NULL; *out_triangles = NULL; *out_node_count = 0;
	uint32_t n = (uint32_t) mesh->triangle_count;
	if (n == 0 || !mesh->host_positions) return 1;
	sah_builder_t b;
	memset(&b, 0, sizeof(b));
	float* vertices = (float*) malloc(sizeof(float) * 9 * (size_t) n);
	b.bounds = (box_t*) malloc(sizeof(box_t) * (size_t) n);
	b.centroids = (float*) malloc(sizeof(float) * 3 * (size_t) n);
	b.order = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) n);
	b.nodes = (float*) malloc(sizeof(float) * 8 * (2 * (size_t) n - 1));
	b.triangles = (float*) malloc(sizeof(float) * 12 * (size_t) n);
	b.pad = pad;
	b.vertices = vertices;
	int failed = !vertices || !b.bounds || !b.centroids || !b.order || !b.nodes || !b.triangles;
	if (!failed) {
		/* de-quantise like reference scene.c:176-187: multiply, then add (two roundings) */
		for (size_t i = 0; i != 3 * (size_t) n; ++i) {
			uint32_t q0 = mesh->host_positions[2 * i], q1 = mesh->host_positions[2 * i + 1];
			float p[3] = {
				(float) (q0 & 0x1FFFFF),
				(float) (((q0 & 0xFFE00000u) >> 21) | ((q1 & 0x3FF) << 11)),
				(float) ((q1 & 0x7FFFFC00u) >> 10)};
			for (int j = 0; j != 3; ++j) {
				volatile float product = p[j] * mesh->dequantization_factor[j];
				vertices[3 * i + j] = product + mesh->dequantization_summand[j];
			}
		}
		for (uint32_t t = 0; t != n; ++t) {
			const float* v = vertices + 9 * (size_t) t;
			box_reset(&b.bounds[t]);
			for (int k = 0; k != 3; ++k)
				for (int j = 0; j != 3; ++j) {
					if (v[3 * k + j] < b.bounds[t].lo[j]) b.bounds[t].lo[j] = v[3 * k + j];
					if (v[3 * k + j] > b.bounds[t].hi[j]) b.bounds[t].hi[j] = v[3 * k + j];
				}
			for (int j = 0; j != 3; ++j) b.centroids[3 * (size_t) t + j] = 0.5f * (b.bounds[t].lo[j] + b.bounds[t].hi[j]);
			b.order[t] = t;
		}
		build_range(&b, 0, n, 0);
		*out_nodes = b.nodes;
		*out_triangles = b.triangles;
		*out_node_count = 2 * n - 1;
	}
	else {
		free(b.nodes);
		free(b.triangles);
	}
	free(vertices);
	free(b.bounds);
	free(b.centroids);
	free(b.order);
	return failed;
}
