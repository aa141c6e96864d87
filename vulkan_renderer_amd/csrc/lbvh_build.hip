// BVH construction on the GPU.  Replaces create_acceleration_structure of the reference
// (src/scene.c:142-406), which hands the same de-quantised triangle soup to the Vulkan driver.
// Runs once per scene.  Three builders produce the same intermediate form (fp32 nodes in
// depth-first "threaded" order, one triangle per leaf):
//   - binned surface-area heuristic, breadth-first, by HIP kernels (the default: "sah device")
//   - Morton-code LBVH (Karras 2012: Morton codes -> radix sort -> binary radix tree -> refit)
//   - the binned SAH on the host (host/sah_bvh.c), the plain C statement of the first one
// and two kernels turn it into what the traversal reads (lbvh.h): k_quantize_nodes (16-byte
// binary nodes) and k_collapse_level (64-byte four-wide nodes).
#include "lbvh.h"
#include "host/vkr_internal.h"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <time.h>

using namespace vkr;

namespace {

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
// inside a do { } while (0) that cleans up behind itself
#define HIP_OK_BREAK(call) if ((call) != hipSuccess) break

struct build_params {
	const uint2* quantized_positions;
	// primitives of the build: triangles, or - device SAH with split triangles - fragments of triangles
	uint32_t triangle_count;
	float factor[3], summand[3];
	float pad;
	// Split triangles (see "fragments" below): primitive t is the part of triangle fragment_triangle[t] inside the box
	// fragment_boxes[6 t ... 6 t + 5] (lo.xyz, hi.xyz).  NULL: every primitive is a whole triangle.
	const uint32_t* fragment_triangle;
	const float* fragment_boxes;
};

// De-quantisation with two roundings (multiply, then add) like scene.c:176-187; the
// shading path uses a fused decode instead (mesh_quantization.glsl:38-45).
__device__ __forceinline__ f3 dequantize(uint2 q, const build_params& p) {
	float x = (float) (q.x & 0x1FFFFF);
	float y = (float) (((q.x & 0xFFE00000u) >> 21) | ((q.y & 0x3FF) << 11));
	float z = (float) ((q.y & 0x7FFFFC00u) >> 10);
	return mk3(__fadd_rn(__fmul_rn(x, p.factor[0]), p.summand[0]), __fadd_rn(__fmul_rn(y, p.factor[1]), p.summand[1]), __fadd_rn(__fmul_rn(z, p.factor[2]), p.summand[2]));
}

__device__ __forceinline__ uint32_t spread_bits_10(uint32_t v) {
	v &= 0x3FF;
	v = (v | (v << 16)) & 0x030000FF;
	v = (v | (v << 8)) & 0x0300F00F;
	v = (v | (v << 4)) & 0x030C30C3;
	v = (v | (v << 2)) & 0x09249249;
	return v;
}

__global__ void __launch_bounds__(256) k_morton_keys(build_params p, uint64_t* keys) {
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= p.triangle_count) return;
	// centroid in quantisation units: 21 bits per axis, keep the top 10
	uint32_t sum[3] = {0, 0, 0};
	for (int v = 0; v != 3; ++v) {
		uint2 q = p.quantized_positions[3 * (size_t) t + v];
		sum[0] += q.x & 0x1FFFFF;
		sum[1] += ((q.x & 0xFFE00000u) >> 21) | ((q.y & 0x3FF) << 11);
		sum[2] += (q.y & 0x7FFFFC00u) >> 10;
	}
	uint32_t cx = (sum[0] / 3) >> 11, cy = (sum[1] / 3) >> 11, cz = (sum[2] / 3) >> 11;
	uint32_t morton = (spread_bits_10(cx) << 2) | (spread_bits_10(cy) << 1) | spread_bits_10(cz);
	keys[t] = ((uint64_t) morton << 32) | t;
}

__global__ void __launch_bounds__(256) k_write_leaves(build_params p, const uint64_t* sorted_keys, float4* triangles, float* lo, float* hi) {
	uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= p.triangle_count) return;
	uint32_t t = (uint32_t) (sorted_keys[slot] & 0xFFFFFFFFu);
	f3 v[3];
	for (int i = 0; i != 3; ++i) v[i] = dequantize(p.quantized_positions[3 * (size_t) t + i], p);
	triangles[3 * (size_t) slot + 0] = make_float4(v[0].x, v[0].y, v[0].z, __uint_as_float(t));
	triangles[3 * (size_t) slot + 1] = make_float4(v[1].x, v[1].y, v[1].z, 0.0f);
	triangles[3 * (size_t) slot + 2] = make_float4(v[2].x, v[2].y, v[2].z, 0.0f);
	// leaf boxes live behind the inner-node boxes
	size_t n = p.triangle_count - 1 + slot;
	lo[3 * n + 0] = fminf(v[0].x, fminf(v[1].x, v[2].x)) - p.pad;
	lo[3 * n + 1] = fminf(v[0].y, fminf(v[1].y, v[2].y)) - p.pad;
	lo[3 * n + 2] = fminf(v[0].z, fminf(v[1].z, v[2].z)) - p.pad;
	hi[3 * n + 0] = fmaxf(v[0].x, fmaxf(v[1].x, v[2].x)) + p.pad;
	hi[3 * n + 1] = fmaxf(v[0].y, fmaxf(v[1].y, v[2].y)) + p.pad;
	hi[3 * n + 2] = fmaxf(v[0].z, fmaxf(v[1].z, v[2].z)) + p.pad;
}

__device__ __forceinline__ int common_prefix(const uint64_t* keys, int n, int i, int j) {
	if (j < 0 || j >= n) return -1;
	return __clzll((long long) (keys[i] ^ keys[j]));
}

// One thread per inner node: range and split of the binary radix tree.  Inner
// node i covers leaves [first, last]; children are inner nodes or leaves.
__global__ void __launch_bounds__(256) k_build_hierarchy(const uint64_t* keys, int leaf_count, bvh_build_node* nodes, uint32_t* leaf_parents) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= leaf_count - 1) return;
	int direction = (common_prefix(keys, leaf_count, i, i + 1) - common_prefix(keys, leaf_count, i, i - 1)) >= 0 ? 1 : -1;
	int min_prefix = common_prefix(keys, leaf_count, i, i - direction);
	int max_length = 2;
	while (common_prefix(keys, leaf_count, i, i + max_length * direction) > min_prefix) max_length *= 2;
	int length = 0;
	for (int step = max_length / 2; step >= 1; step /= 2)
		if (common_prefix(keys, leaf_count, i, i + (length + step) * direction) > min_prefix) length += step;
	int j = i + length * direction;
	int node_prefix = common_prefix(keys, leaf_count, i, j);
	int split_offset = 0;
	int step = length;
	do {
		step = (step + 1) >> 1;
		if (common_prefix(keys, leaf_count, i, i + (split_offset + step) * direction) > node_prefix) split_offset += step;
	} while (step > 1);
	int split = i + split_offset * direction + min(direction, 0);
	int first = min(i, j), last = max(i, j);
	uint32_t left = (split == first) ? (kLeafBit | (uint32_t) split) : (uint32_t) split;
	uint32_t right = (split + 1 == last) ? (kLeafBit | (uint32_t) (split + 1)) : (uint32_t) (split + 1);
	nodes[i].links.x = left;
	nodes[i].links.y = right;
	nodes[i].links.w = (uint32_t) first;
	nodes[i].range = make_uint2((uint32_t) last, (uint32_t) (split - first + 1));
	if (i == 0) nodes[0].links.z = 0xFFFFFFFFu;
	if (left & kLeafBit) leaf_parents[left & ~kLeafBit] = (uint32_t) i; else nodes[left].links.z = (uint32_t) i;
	if (right & kLeafBit) leaf_parents[right & ~kLeafBit] = (uint32_t) i; else nodes[right].links.z = (uint32_t) i;
}

__device__ __forceinline__ void load_box(const float* lo, const float* hi, size_t index, f3& out_lo, f3& out_hi) {
	// boxes written by other workgroups: bypass the (non-coherent) vector L1
	out_lo = mk3(__builtin_nontemporal_load(lo + 3 * index), __builtin_nontemporal_load(lo + 3 * index + 1), __builtin_nontemporal_load(lo + 3 * index + 2));
	out_hi = mk3(__builtin_nontemporal_load(hi + 3 * index), __builtin_nontemporal_load(hi + 3 * index + 1), __builtin_nontemporal_load(hi + 3 * index + 2));
}

// Bottom-up refit: each leaf climbs; the second thread to reach a node merges the
// child boxes.  Cross-workgroup visibility: agent-scope release before the arrival
// counter, agent-scope acquire after it (per-XCD L2s are not coherent).
__global__ void __launch_bounds__(256) k_refit(int leaf_count, const bvh_build_node* nodes, const uint32_t* leaf_parents, float* lo, float* hi, uint32_t* arrivals) {
	int slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= leaf_count) return;
	uint32_t node = leaf_parents[slot];
	while (node != 0xFFFFFFFFu) {
		__threadfence();
		if (atomicAdd(&arrivals[node], 1u) == 0) return;
		__threadfence();
		uint32_t left = nodes[node].links.x, right = nodes[node].links.y;
		size_t il = (left & kLeafBit) ? (size_t) (leaf_count - 1) + (left & ~kLeafBit) : (size_t) left;
		size_t ir = (right & kLeafBit) ? (size_t) (leaf_count - 1) + (right & ~kLeafBit) : (size_t) right;
		f3 lo0, hi0, lo1, hi1;
		load_box(lo, hi, il, lo0, hi0);
		load_box(lo, hi, ir, lo1, hi1);
		lo[3 * (size_t) node + 0] = fminf(lo0.x, lo1.x);
		lo[3 * (size_t) node + 1] = fminf(lo0.y, lo1.y);
		lo[3 * (size_t) node + 2] = fminf(lo0.z, lo1.z);
		hi[3 * (size_t) node + 0] = fmaxf(hi0.x, hi1.x);
		hi[3 * (size_t) node + 1] = fmaxf(hi0.y, hi1.y);
		hi[3 * (size_t) node + 2] = fmaxf(hi0.z, hi1.z);
		node = nodes[node].links.z;
	}
}

// Depth-first ("threaded") layout.  One thread per node of the radix tree (inner
// nodes first, then leaves).  The position of a node in depth-first order is the
// number of nodes visited before it: walking up to the root, every step from a left
// child adds the parent itself, every step from a right child adds the parent and
// the parent's whole left subtree (2 * leaves - 1 nodes).
__global__ void __launch_bounds__(256) k_thread_nodes(int leaf_count, const bvh_build_node* nodes, const uint32_t* leaf_parents, const float* lo, const float* hi, float4* threaded) {
	int id = blockIdx.x * blockDim.x + threadIdx.x;
	int total = 2 * leaf_count - 1;
	if (id >= total) return;
	bool is_leaf = id >= leaf_count - 1;
	uint32_t slot = is_leaf ? (uint32_t) (id - (leaf_count - 1)) : 0u;
	uint32_t code = is_leaf ? (kLeafBit | slot) : (uint32_t) id;
	uint32_t leaves = is_leaf ? 1u : (nodes[id].range.x - nodes[id].links.w + 1u);
	uint32_t parent = is_leaf ? (leaf_count > 1 ? leaf_parents[slot] : 0xFFFFFFFFu) : nodes[id].links.z;
	uint32_t position = 0;
	while (parent != 0xFFFFFFFFu) {
		bool right = nodes[parent].links.y == code;
		position += right ? 2u * nodes[parent].range.y : 1u;
		code = parent;
		parent = nodes[parent].links.z;
	}
	uint32_t skip = position + 2u * leaves - 1u;
	threaded[2 * (size_t) position] = make_float4(lo[3 * (size_t) id], lo[3 * (size_t) id + 1], lo[3 * (size_t) id + 2], hi[3 * (size_t) id]);
	threaded[2 * (size_t) position + 1] = make_float4(hi[3 * (size_t) id + 1], hi[3 * (size_t) id + 2], __uint_as_float(skip), __uint_as_float(is_leaf ? slot : kNoLeaf));
}

// fp32 threaded nodes (32 bytes: lo.xyz, hi.x | hi.yz, skip, leaf) -> 16-byte nodes with
// boxes rounded outwards on the 15-bit grid (layout: lbvh.h)
__global__ void __launch_bounds__(256) k_quantize_nodes(uint32_t node_count, const float4* nodes, f3 origin, f3 inverse_cell, uint4* quantized) {
	uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= node_count) return;
	float4 a = nodes[2 * (size_t) id], b = nodes[2 * (size_t) id + 1];
	float lo[3] = {a.x, a.y, a.z}, hi[3] = {a.w, b.x, b.y};
	float g0[3] = {origin.x, origin.y, origin.z}, scale[3] = {inverse_cell.x, inverse_cell.y, inverse_cell.z};
	uint32_t packed[3];
	for (int j = 0; j != 3; ++j) {
		float q0 = floorf((lo[j] - g0[j]) * scale[j] - kGridMargin);
		float q1 = ceilf((hi[j] - g0[j]) * scale[j] + kGridMargin);
		q0 = fminf(fmaxf(q0, 0.0f), kGridMax);
		q1 = fminf(fmaxf(q1, 0.0f), kGridMax);
		packed[j] = (uint32_t) q0 | ((uint32_t) q1 << 16);
	}
	uint32_t skip = __float_as_uint(b.z), leaf = __float_as_uint(b.w);
	quantized[id] = make_uint4(packed[0], packed[1], packed[2], leaf != kNoLeaf ? (kLeafBit | leaf) : skip);
}


// ---- binned SAH, breadth-first on the device ----------------------------------------------
// The algorithm of host/sah_bvh.c (16 centroid bins per axis, cost = area x count of the two
// sides, one triangle per leaf) without its recursion and without moving triangles around:
// a node of the depth-first layout is fully described by (position, first leaf slot, triangle
// count) - the left child of (p, f, c) with l triangles on the left is (p + 1, f, l), the right
// one (p + 2 l, f + l, c - l) - so a triangle only has to remember which node of the current
// level it is in.  Per level: bin the triangles of every open node (atomics on the node's bins),
// pick each node's split, hand every triangle to its child (growing the child's boxes with
// atomics) or, when the child holds one triangle, write the leaf.  Float minima / maxima are
// atomics on an order-preserving integer encoding, hence independent of the order of arrival.
constexpr int kSahBinCount = 16;
constexpr uint32_t kSahDone = 0xFFFFFFFFu;
constexpr uint32_t kSahLeafChild = 0xFFFFFFFEu;

struct sah_open_node {
	uint32_t position, first, count;
	uint32_t fallback_rank;        // hands out ranks when no plane separates the centroids
	uint32_t bounds[6];            // lo.xyz, hi.xyz of the triangles, ordered encoding
	uint32_t centroid_bounds[6];
	// the split (k_sah_split)
	int32_t axis, split_bin;
	float bin_origin, bin_scale;
	uint32_t left_count;
	uint32_t child[2];             // index among the next level's open nodes or kSahLeafChild
	uint32_t pad;
};

struct sah_bin {
	uint32_t count;
	uint32_t lo[3], hi[3];
};

__device__ __forceinline__ uint32_t ordered(float f) {
	uint32_t bits = __float_as_uint(f);
	return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u) {
	return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ void reset_open_node(sah_open_node* node, uint32_t position, uint32_t first, uint32_t count) {
	node->position = position; node->first = first; node->count = count; node->fallback_rank = 0;
	for (int j = 0; j != 3; ++j) {
		node->bounds[j] = node->centroid_bounds[j] = 0xFFFFFFFFu;
		node->bounds[3 + j] = node->centroid_bounds[3 + j] = 0u;
	}
}

// Grows the boxes of an open node by one triangle, directly in global memory
__device__ __forceinline__ void grow_open_node(sah_open_node* node, f3 lo, f3 hi) {
	const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
	for (int j = 0; j != 3; ++j) {
		// centroids like sah_bvh.c: 0.5 (lo + hi)
		uint32_t c = ordered(0.5f * (l[j] + h[j]));
		atomicMin(&node->bounds[j], ordered(l[j])); atomicMax(&node->bounds[3 + j], ordered(h[j]));
		atomicMin(&node->centroid_bounds[j], c); atomicMax(&node->centroid_bounds[3 + j], c);
	}
}

// index in the mesh of the triangle that primitive t is (a fragment of)
__device__ __forceinline__ uint32_t source_triangle(const build_params& p, uint32_t t) { return p.fragment_triangle ? p.fragment_triangle[t] : t; }

__device__ __forceinline__ void triangle_bounds(const build_params& p, uint32_t t, f3 (&v)[3], f3& lo, f3& hi) {
	uint32_t triangle = source_triangle(p, t);
	for (int i = 0; i != 3; ++i) v[i] = dequantize(p.quantized_positions[3 * (size_t) triangle + i], p);
	if (p.fragment_boxes) {
		const float* box = p.fragment_boxes + 6 * (size_t) t;
		lo = mk3(box[0], box[1], box[2]);
		hi = mk3(box[3], box[4], box[5]);
		return;
	}
	lo = mk3(fminf(v[0].x, fminf(v[1].x, v[2].x)), fminf(v[0].y, fminf(v[1].y, v[2].y)), fminf(v[0].z, fminf(v[1].z, v[2].z)));
	hi = mk3(fmaxf(v[0].x, fmaxf(v[1].x, v[2].x)), fmaxf(v[0].y, fmaxf(v[1].y, v[2].y)), fmaxf(v[0].z, fmaxf(v[1].z, v[2].z)));
}

__device__ __forceinline__ void write_threaded_node(float4* threaded, uint32_t position, f3 lo, f3 hi, float pad, uint32_t skip, uint32_t leaf) {
	threaded[2 * (size_t) position] = make_float4(lo.x - pad, lo.y - pad, lo.z - pad, hi.x + pad);
	threaded[2 * (size_t) position + 1] = make_float4(hi.y + pad, hi.z + pad, __uint_as_float(skip), __uint_as_float(leaf));
}

__device__ __forceinline__ void write_leaf(const build_params& p, uint32_t t, const f3 (&v)[3], f3 lo, f3 hi, uint32_t position, uint32_t slot, float4* threaded, float4* triangles) {
	write_threaded_node(threaded, position, lo, hi, p.pad, position + 1u, slot);
	// (w of the first vertex: the triangle's index in the mesh - what primary visibility reports)
	triangles[3 * (size_t) slot + 0] = make_float4(v[0].x, v[0].y, v[0].z, __uint_as_float(source_triangle(p, t)));
	triangles[3 * (size_t) slot + 1] = make_float4(v[1].x, v[1].y, v[1].z, 0.0f);
	triangles[3 * (size_t) slot + 2] = make_float4(v[2].x, v[2].y, v[2].z, 0.0f);
}

// ---- contributions of a workgroup combined in LDS ------------------------------------------
// Atomics of device scope are performed behind the L2s of the eight XCDs, a few billion per second
// whatever their address; the levels of a build issue 12 (boxes of the children) and 21 (bins) of
// them per triangle.  The 256 consecutive triangles of a workgroup mostly belong to a handful of
// open nodes, so the workgroup combines their contributions in LDS - a 16-slot table keyed by the
// node - and sends one atomic per field that received something; a triangle that finds the table
// full (deep levels: few triangles per node, hence little contention) goes to global memory directly.
// Counts add up and bounds are minima / maxima: the tree is the one of the direct atomics.
constexpr uint32_t kSahGroupSlots = 16;
constexpr uint32_t kSahEmptyKey = 0xFFFFFFFFu;
__device__ __forceinline__ int group_slot(uint32_t* keys, uint32_t key) {
	uint32_t h = (key * 2654435761u) >> 28;
	for (uint32_t probe = 0; probe != kSahGroupSlots; ++probe) {
		uint32_t old = atomicCAS(&keys[h], kSahEmptyKey, key);
		if (old == kSahEmptyKey || old == key) return (int) h;
		h = (h + 1u) & (kSahGroupSlots - 1u);
	}
	return -1;
}
// the twelve words of an open node's boxes as one array: bounds lo.xyz hi.xyz, centroid bounds lo.xyz hi.xyz
__device__ __forceinline__ void reset_box_words(uint32_t* words) {
	for (int j = 0; j != 12; ++j) words[j] = (j % 6) < 3 ? 0xFFFFFFFFu : 0u;
}
__device__ __forceinline__ void grow_box_words(uint32_t* words, f3 lo, f3 hi) {
	const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
	for (int j = 0; j != 3; ++j) {
		uint32_t c = ordered(0.5f * (l[j] + h[j]));
		atomicMin(&words[j], ordered(l[j])); atomicMax(&words[3 + j], ordered(h[j]));
		atomicMin(&words[6 + j], c); atomicMax(&words[9 + j], c);
	}
}
__device__ __forceinline__ void flush_box_word(sah_open_node* node, uint32_t field, uint32_t value) {
	uint32_t* target = field < 6 ? &node->bounds[field] : &node->centroid_bounds[field - 6];
	if ((field % 6) < 3) atomicMin(target, value);
	else atomicMax(target, value);
}

// level 0: every triangle is in the root
__global__ void __launch_bounds__(256) k_sah_init(build_params p, uint32_t* triangle_node, sah_open_node* root, float4* threaded, float4* triangles) {
	__shared__ uint32_t words[12];
	if (threadIdx.x < 12) words[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xFFFFFFFFu : 0u;
	__syncthreads();
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	bool active = t < p.triangle_count;
	f3 v[3], lo = mk3(0.0f, 0.0f, 0.0f), hi = lo;
	if (active) {
		triangle_bounds(p, t, v, lo, hi);
		triangle_node[t] = p.triangle_count > 1 ? 0u : kSahDone;
		if (p.triangle_count == 1) write_leaf(p, t, v, lo, hi, 0u, 0u, threaded, triangles);
		grow_box_words(words, lo, hi);
	}
	__syncthreads();
	if (threadIdx.x < 12) flush_box_word(root, threadIdx.x, words[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_sah_reset_root(sah_open_node* root, uint32_t triangle_count, uint32_t* counters) {
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		reset_open_node(root, 0u, 0u, triangle_count);
		counters[0] = counters[1] = 0u;
	}
}

__global__ void __launch_bounds__(256) k_sah_clear_bins(sah_bin* bins, uint32_t bin_count) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= bin_count) return;
	sah_bin b;
	b.count = 0;
	for (int j = 0; j != 3; ++j) { b.lo[j] = 0xFFFFFFFFu; b.hi[j] = 0u; }
	bins[i] = b;
}

// bin index of a centroid coordinate, the arithmetic of sah_bvh.c
__device__ __forceinline__ float sah_bin_scale(float lo, float hi) {
	float width = hi - lo;
	return width > 0.0f ? (float) kSahBinCount * (1.0f - 1.0e-6f) / width : 0.0f;
}
__device__ __forceinline__ int sah_bin_index(float c, float lo, float scale) {
	int k = (int) ((c - lo) * scale);
	return k < 0 ? 0 : (k >= kSahBinCount ? kSahBinCount - 1 : k);
}

__device__ __forceinline__ void bin_triangle(sah_bin* node_bins, const sah_open_node* node, f3 lo, f3 hi) {
	const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
	for (int j = 0; j != 3; ++j) {
		float c_lo = unordered(node->centroid_bounds[j]), c_hi = unordered(node->centroid_bounds[3 + j]);
		float scale = sah_bin_scale(c_lo, c_hi);
		if (!(scale > 0.0f)) continue;
		int k = sah_bin_index(0.5f * (l[j] + h[j]), c_lo, scale);
		sah_bin* bin = node_bins + j * kSahBinCount + k;
		atomicAdd(&bin->count, 1u);
		for (int a = 0; a != 3; ++a) {
			atomicMin(&bin->lo[a], ordered(l[a]));
			atomicMax(&bin->hi[a], ordered(h[a]));
		}
	}
}
__device__ __forceinline__ void flush_bin(sah_bin* out, const sah_bin& in) {
	if (!in.count) return;
	atomicAdd(&out->count, in.count);
	for (int a = 0; a != 3; ++a) {
		atomicMin(&out->lo[a], in.lo[a]);
		atomicMax(&out->hi[a], in.hi[a]);
	}
}

// Bins of the open nodes of a level (more than kSahSharedNodes of them, see below), combined per workgroup
__global__ void __launch_bounds__(256) k_sah_bin(build_params p, const uint32_t* triangle_node, const sah_open_node* open, sah_bin* bins) {
	constexpr uint32_t kBinsPerNode = 3 * kSahBinCount;
	__shared__ uint32_t keys[kSahGroupSlots];
	__shared__ sah_bin shared_bins[kSahGroupSlots * kBinsPerNode];
	if (threadIdx.x < kSahGroupSlots) keys[threadIdx.x] = kSahEmptyKey;
	for (uint32_t i = threadIdx.x; i < kSahGroupSlots * kBinsPerNode; i += 256u) {
		shared_bins[i].count = 0;
		for (int a = 0; a != 3; ++a) { shared_bins[i].lo[a] = 0xFFFFFFFFu; shared_bins[i].hi[a] = 0u; }
	}
	__syncthreads();
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t id = t < p.triangle_count ? triangle_node[t] : kSahDone;
	if (id != kSahDone) {
		f3 v[3], lo, hi;
		triangle_bounds(p, t, v, lo, hi);
		int slot = group_slot(keys, id);
		if (slot >= 0) bin_triangle(shared_bins + (uint32_t) slot * kBinsPerNode, open + id, lo, hi);
		else bin_triangle(bins + (size_t) id * kBinsPerNode, open + id, lo, hi);
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < kSahGroupSlots * kBinsPerNode; i += 256u) {
		uint32_t key = keys[i / kBinsPerNode];
		if (key != kSahEmptyKey) flush_bin(bins + (size_t) key * kBinsPerNode + i % kBinsPerNode, shared_bins[i]);
	}
}

// The first levels, where a few open nodes hold all triangles (level 0: 2.8 M atomics on 48 bins
// were 11 ms of a 38 ms build): the table is indexed by the node itself, no triangle goes around it.
constexpr uint32_t kSahSharedNodes = 32;  // (8 until round 3: the first level without LDS binning took 1.3 ms of a 15 ms build)
__global__ void __launch_bounds__(256) k_sah_bin_shared(build_params p, const uint32_t* triangle_node, const sah_open_node* open, uint32_t open_count, sah_bin* bins) {
	__shared__ sah_bin shared_bins[kSahSharedNodes * 3 * kSahBinCount];
	const uint32_t bin_count = open_count * 3u * kSahBinCount;
	for (uint32_t i = threadIdx.x; i < bin_count; i += 256u) {
		shared_bins[i].count = 0;
		for (int a = 0; a != 3; ++a) { shared_bins[i].lo[a] = 0xFFFFFFFFu; shared_bins[i].hi[a] = 0u; }
	}
	__syncthreads();
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t id = t < p.triangle_count ? triangle_node[t] : kSahDone;
	if (id != kSahDone) {
		f3 v[3], lo, hi;
		triangle_bounds(p, t, v, lo, hi);
		bin_triangle(shared_bins + id * 3u * kSahBinCount, open + id, lo, hi);
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < bin_count; i += 256u) flush_bin(bins + i, shared_bins[i]);
}

struct sah_box {
	float lo[3], hi[3];
};
__device__ __forceinline__ void sah_box_reset(sah_box& b) {
	for (int j = 0; j != 3; ++j) { b.lo[j] = 3.402823466e+38f; b.hi[j] = -3.402823466e+38f; }
}
__device__ __forceinline__ void sah_box_merge(sah_box& b, const sah_bin& bin) {
	for (int j = 0; j != 3; ++j) {
		b.lo[j] = fminf(b.lo[j], unordered(bin.lo[j]));
		b.hi[j] = fmaxf(b.hi[j], unordered(bin.hi[j]));
	}
}
__device__ __forceinline__ float sah_half_area(const sah_box& b) {
	float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
	return x * y + y * z + z * x;
}

// The cheapest of the 3 x 15 splits of every open node, the node itself in the output, its children
// as open nodes of the next level.  counters[0]: open nodes of the next level.
// A wave looks at one node at a time: lane 16 j + k holds bin k of axis j (sixteen lanes idle), the
// boxes and counts left and right of every plane come from segmented scans - minima, maxima and
// integer sums, so the order of the merges does not matter - the 45 costs are evaluated side by side
// with the expression of sah_bvh.c, and a reduction over (cost, lane) picks the first of the cheapest in
// the order axis, bin: what the strict comparison in sah_bvh.c's loops picks.  A workgroup handles 64
// nodes, sixteen per wave, and then one lane per node writes the node and allocates its children, so
// that the compiler can combine the wave's allocations into one atomic (one wave per node meant 130 k
// atomics on one address at the widest level: 0.5 ms).  Measured (profiles/r03w/): 5 us for the first
// levels (one thread per node walking 90 bins one after the other took 37 us whatever the level); the
// deep levels stay at 40 - 57 us either way, because their 66 k open nodes have 88 MB of bins to read.
constexpr uint32_t kSplitNodesPerGroup = 64;
__global__ void __launch_bounds__(256) k_sah_split(sah_open_node* open, uint32_t open_count, const sah_bin* bins, sah_open_node* next_open, uint32_t* counters, float4* threaded, float pad) {
	__shared__ uint32_t best_lanes[kSplitNodesPerGroup], best_lefts[kSplitNodesPerGroup];
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t axis = lane >> 4, k = lane & 15u;
	for (uint32_t i = 0; i != kSplitNodesPerGroup / 4u; ++i) {
		const uint32_t local = i * 4u + wave;
		const uint32_t id = blockIdx.x * kSplitNodesPerGroup + local;
		if (id >= open_count) break;
		const sah_open_node* node = open + id;
		bool splittable = false;
		sah_box left, right;
		sah_box_reset(left);
		uint32_t left_n = 0;
		if (axis < 3u) {
			splittable = sah_bin_scale(unordered(node->centroid_bounds[axis]), unordered(node->centroid_bounds[3 + axis])) > 0.0f;
			const sah_bin bin = bins[(size_t) id * (3 * kSahBinCount) + lane];
			if (bin.count) sah_box_merge(left, bin);
			left_n = bin.count;
		}
		right = left;
		uint32_t right_n = left_n;
#pragma unroll
		for (uint32_t offset = 1; offset < (uint32_t) kSahBinCount; offset <<= 1) {
			const bool from_below = k >= offset, from_above = k + offset < (uint32_t) kSahBinCount;
			uint32_t n = (uint32_t) __shfl_up((int) left_n, offset, kSahBinCount);
			left_n += from_below ? n : 0u;
			n = (uint32_t) __shfl_down((int) right_n, offset, kSahBinCount);
			right_n += from_above ? n : 0u;
			for (int j = 0; j != 3; ++j) {
				float lo = __shfl_up(left.lo[j], offset, kSahBinCount), hi = __shfl_up(left.hi[j], offset, kSahBinCount);
				if (from_below) { left.lo[j] = fminf(left.lo[j], lo); left.hi[j] = fmaxf(left.hi[j], hi); }
				lo = __shfl_down(right.lo[j], offset, kSahBinCount); hi = __shfl_down(right.hi[j], offset, kSahBinCount);
				if (from_above) { right.lo[j] = fminf(right.lo[j], lo); right.hi[j] = fmaxf(right.hi[j], hi); }
			}
		}
		// the plane behind bin k: bins 0 ... k on the left, k + 1 ... 15 on the right
		const float right_area = right_n ? sah_half_area(right) : 0.0f;
		const float next_area = __shfl_down(right_area, 1, kSahBinCount);
		const uint32_t next_n = (uint32_t) __shfl_down((int) right_n, 1, kSahBinCount);
		float cost = 3.402823466e+38f;
		if (splittable && k + 1u < (uint32_t) kSahBinCount && left_n != 0u && next_n != 0u)
			cost = sah_half_area(left) * (float) left_n + next_area * (float) next_n;
		uint32_t best = cost < 3.402823466e+38f ? lane : 0xFFFFFFFFu;
		if (best == 0xFFFFFFFFu) cost = 3.402823466e+38f;
#pragma unroll
		for (int offset = 32; offset > 0; offset >>= 1) {
			float other_cost = __shfl_xor(cost, offset);
			uint32_t other = (uint32_t) __shfl_xor((int) best, offset);
			bool take = other_cost < cost || (other_cost == cost && other < best);
			cost = take ? other_cost : cost;
			best = take ? other : best;
		}
		const uint32_t best_left = (uint32_t) __shfl((int) left_n, (int) (best & 63u));
		if (lane == 0) { best_lanes[local] = best; best_lefts[local] = best_left; }
	}
	__syncthreads();
	const uint32_t id = blockIdx.x * kSplitNodesPerGroup + threadIdx.x;
	if (threadIdx.x >= kSplitNodesPerGroup || id >= open_count) return;
	sah_open_node* node = open + id;
	const uint32_t best = best_lanes[threadIdx.x], best_left = best_lefts[threadIdx.x];
	uint32_t count = node->count;
	int best_axis = best == 0xFFFFFFFFu ? -1 : (int) (best >> 4), best_split = best == 0xFFFFFFFFu ? 0 : (int) (best & 15u);
	// all centroids coincide: any split is as good as another (triangles are dealt by rank)
	uint32_t left_count = best_axis < 0 ? count / 2u : best_left;
	node->axis = best_axis;
	node->split_bin = best_split;
	node->left_count = left_count;
	if (best_axis >= 0) {
		node->bin_origin = unordered(node->centroid_bounds[best_axis]);
		node->bin_scale = sah_bin_scale(node->bin_origin, unordered(node->centroid_bounds[3 + best_axis]));
	}
	f3 lo = mk3(unordered(node->bounds[0]), unordered(node->bounds[1]), unordered(node->bounds[2]));
	f3 hi = mk3(unordered(node->bounds[3]), unordered(node->bounds[4]), unordered(node->bounds[5]));
	write_threaded_node(threaded, node->position, lo, hi, pad, node->position + 2u * count - 1u, kNoLeaf);
	for (uint32_t side = 0; side != 2; ++side) {
		uint32_t child_count = side ? count - left_count : left_count;
		if (child_count < 2) { node->child[side] = kSahLeafChild; continue; }
		uint32_t child = atomicAdd(&counters[0], 1u);
		node->child[side] = child;
		reset_open_node(next_open + child, side ? node->position + 2u * left_count : node->position + 1u, side ? node->first + left_count : node->first, child_count);
	}
}

// Every triangle of an open node moves to the child on its side of the split, or becomes a leaf.
// The boxes of the children grow by the triangles they receive (combined per workgroup, see above).
__global__ void __launch_bounds__(256) k_sah_assign(build_params p, uint32_t* triangle_node, sah_open_node* open, sah_open_node* next_open, float4* threaded, float4* triangles) {
	__shared__ uint32_t keys[kSahGroupSlots];
	__shared__ uint32_t words[kSahGroupSlots * 12];
	if (threadIdx.x < kSahGroupSlots) keys[threadIdx.x] = kSahEmptyKey;
	if (threadIdx.x < kSahGroupSlots * 12) words[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xFFFFFFFFu : 0u;
	__syncthreads();
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t id = t < p.triangle_count ? triangle_node[t] : kSahDone;
	if (id != kSahDone) {
		sah_open_node* node = open + id;
		f3 v[3], lo, hi;
		triangle_bounds(p, t, v, lo, hi);
		int axis = node->axis;
		uint32_t side;
		if (axis >= 0) {
			float c = 0.5f * (axis == 0 ? lo.x + hi.x : (axis == 1 ? lo.y + hi.y : lo.z + hi.z));
			side = sah_bin_index(c, node->bin_origin, node->bin_scale) > node->split_bin ? 1u : 0u;
		}
		else side = atomicAdd(&node->fallback_rank, 1u) >= node->left_count ? 1u : 0u;
		uint32_t child = node->child[side];
		if (child == kSahLeafChild) {
			uint32_t position = side ? node->position + 2u * node->left_count : node->position + 1u;
			uint32_t slot = side ? node->first + node->left_count : node->first;
			write_leaf(p, t, v, lo, hi, position, slot, threaded, triangles);
			triangle_node[t] = kSahDone;
		}
		else {
			triangle_node[t] = child;
			int slot = group_slot(keys, child);
			if (slot >= 0) grow_box_words(words + 12 * slot, lo, hi);
			else grow_open_node(next_open + child, lo, hi);
		}
	}
	__syncthreads();
	if (threadIdx.x < kSahGroupSlots * 12) {
		uint32_t key = keys[threadIdx.x / 12];
		if (key != kSahEmptyKey) flush_box_word(next_open + key, threadIdx.x % 12, words[threadIdx.x]);
	}
}

// ---- fragments: long thin triangles that run diagonally through their boxes ---------------------
// The box of a slat 3 m long and 6 cm wide that lies at 45 degrees to the axes measures 2.1 m x 2.1 m: a tree of
// such boxes sends every ray through a neighbourhood of the slat into its triangle test (the large scene of
// synthetic.py: 30 triangle tests per shadow ray).  The reference's scenes are full of such triangles (railings,
// awnings, wires), and the drivers it relies on split them.  Here a triangle whose box has more than four times the
// half-area it needs (twice the triangle's area is what a triangle in an axis plane takes) is cut into up to 16
// slabs along the longest axis of its box, and every slab becomes a primitive of the build with the bounds of the
// triangle INSIDE the slab - its box, exactly (the triangle clipped by the two planes), so every point of the
// triangle lies in the box of some fragment, and a leaf per fragment names the whole triangle.  Ray queries
// return what they returned (the same triangles are tested, some by more than one leaf); the tree is what changes.
constexpr uint32_t kMostFragments = 16;

// `most`: the cap per triangle (kMostFragments, or less for a mesh that would otherwise grow more than fourfold)
__device__ __forceinline__ uint32_t fragments_wanted(const f3 (&v)[3], f3 lo, f3 hi, float least_extent, uint32_t most, int& out_axis) {
	f3 e = hi - lo;
	out_axis = (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2);
	float longest = out_axis == 0 ? e.x : (out_axis == 1 ? e.y : e.z);
	if (!(longest > least_extent)) return 1u;
	f3 normal = cross(v[1] - v[0], v[2] - v[0]);
	float twice_area = __builtin_sqrtf(dot(normal, normal));
	float box_half_area = e.x * e.y + e.y * e.z + e.z * e.x;
	float ratio = box_half_area / fmaxf(twice_area, 1.0e-30f);
	if (!(ratio > 4.0f)) return 1u;
	float wanted = ceilf(0.5f * ratio);
	return wanted >= (float) most ? most : (uint32_t) wanted;
}

__global__ void __launch_bounds__(256) k_count_fragments(build_params p, float least_extent, uint32_t most, uint32_t* counts) {
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= p.triangle_count) return;
	f3 v[3], lo, hi;
	triangle_bounds(p, t, v, lo, hi);
	int axis;
	counts[t] = fragments_wanted(v, lo, hi, least_extent, most, axis);
}

// bounds of the part of the triangle with s0 <= x_axis <= s1 (the triangle is cut by the two planes)
__device__ __forceinline__ void slab_bounds(const f3 (&v)[3], int axis, float s0, float s1, f3& lo, f3& hi) {
	float px[8], py[8], pz[8], qx[8], qy[8], qz[8];
	int count = 3;
	for (int i = 0; i != 3; ++i) { px[i] = v[i].x; py[i] = v[i].y; pz[i] = v[i].z; }
	for (int side = 0; side != 2; ++side) {
		// keep x_axis >= s0 (side 0), x_axis <= s1 (side 1)
		float plane = side ? s1 : s0, sign = side ? -1.0f : 1.0f;
		int kept = 0;
		for (int i = 0; i != count; ++i) {
			int j = (i + 1 == count) ? 0 : i + 1;
			float ci = axis == 0 ? px[i] : (axis == 1 ? py[i] : pz[i]), cj = axis == 0 ? px[j] : (axis == 1 ? py[j] : pz[j]);
			float di = sign * (ci - plane), dj = sign * (cj - plane);
			if (di >= 0.0f) { qx[kept] = px[i]; qy[kept] = py[i]; qz[kept] = pz[i]; ++kept; }
			if ((di >= 0.0f) != (dj >= 0.0f)) {
				float w = di / (di - dj);
				qx[kept] = fmaf(w, px[j] - px[i], px[i]); qy[kept] = fmaf(w, py[j] - py[i], py[i]); qz[kept] = fmaf(w, pz[j] - pz[i], pz[i]);
				// (the point lies on the plane: say so exactly, whatever the interpolation rounded to)
				if (axis == 0) qx[kept] = plane; else if (axis == 1) qy[kept] = plane; else qz[kept] = plane;
				++kept;
			}
		}
		count = kept;
		for (int i = 0; i != count; ++i) { px[i] = qx[i]; py[i] = qy[i]; pz[i] = qz[i]; }
	}
	lo = mk3(3.0e38f, 3.0e38f, 3.0e38f); hi = mk3(-3.0e38f, -3.0e38f, -3.0e38f);
	for (int i = 0; i != count; ++i) {
		lo = mk3(fminf(lo.x, px[i]), fminf(lo.y, py[i]), fminf(lo.z, pz[i]));
		hi = mk3(fmaxf(hi.x, px[i]), fmaxf(hi.y, py[i]), fmaxf(hi.z, pz[i]));
	}
}

// first[t]: exclusive prefix sum of the counts
__global__ void __launch_bounds__(256) k_write_fragments(build_params p, float least_extent, uint32_t most, const uint32_t* first, uint32_t* fragment_triangle, float* fragment_boxes) {
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= p.triangle_count) return;
	f3 v[3], lo, hi;
	triangle_bounds(p, t, v, lo, hi);
	int axis;
	uint32_t count = fragments_wanted(v, lo, hi, least_extent, most, axis);
	uint32_t base = first[t];
	float a0 = axis == 0 ? lo.x : (axis == 1 ? lo.y : lo.z), a1 = axis == 0 ? hi.x : (axis == 1 ? hi.y : hi.z);
	for (uint32_t j = 0; j != count; ++j) {
		f3 flo = lo, fhi = hi;
		if (count > 1u) {
			// (the planes between the slabs are shared: slab j ends where slab j + 1 begins, bit for bit)
			float s0 = j == 0u ? a0 : fmaf((float) j / (float) count, a1 - a0, a0);
			float s1 = j + 1u == count ? a1 : fmaf((float) (j + 1u) / (float) count, a1 - a0, a0);
			slab_bounds(v, axis, s0, s1, flo, fhi);
			// a slab that the clipping left empty (rounding at the ends) keeps a point of the triangle's box
			if (!(flo.x <= fhi.x)) { flo = lo; fhi = lo; }
			// never beyond the triangle's own box
			flo = mk3(fmaxf(flo.x, lo.x), fmaxf(flo.y, lo.y), fmaxf(flo.z, lo.z));
			fhi = mk3(fminf(fhi.x, hi.x), fminf(fhi.y, hi.y), fminf(fhi.z, hi.z));
		}
		fragment_triangle[base + j] = t;
		float* box = fragment_boxes + 6 * (size_t) (base + j);
		box[0] = flo.x; box[1] = flo.y; box[2] = flo.z; box[3] = fhi.x; box[4] = fhi.y; box[5] = fhi.z;
	}
}

// ---- collapse to the four-wide layout (lbvh.h) ------------------------------------------------

struct wide_item {
	uint32_t wide_index, binary_position, need;
};

__device__ __forceinline__ float quantized_half_area(uint4 n) {
	float x = (float) ((n.x >> 16) - (n.x & 0xFFFFu)), y = (float) ((n.y >> 16) - (n.y & 0xFFFFu)), z = (float) ((n.z >> 16) - (n.z & 0xFFFFu));
	return x * y + y * z + z * x;
}

// One thread per wide node of this level.  counters: [0] items of the next level, [1] wide nodes
// allocated so far, [2] deepest stack a ray can need.
// order_children (VKR_WIDE_CHILD_ORDER=1, off by default): the children of a node are stored largest box first.
// trace_shadow_rays_wide takes child 0 off its stack first, and a shadow ray is done with the first triangle it hits;
// the idea - VERDICT round 3 - was that the child that covers the most space is the one most likely to hold a blocker
// (any-hit results do not depend on the order, only the work of blocked rays).  Measured (profiles/r05i/): on the
// benchmark scene, 13 % of whose rays are blocked, a blocked ray fetches 6.98 instead of 7.20 nodes and the frame
// moves by 0.4 % (1.560 / 1.566 ms without light shafts); on the large scene (70 % blocked) the large boxes are the
// EMPTY ones - a blocked ray fetches 29.7 instead of 28.2 nodes, tests 29.1 instead of 26.9 triangles, and the frame
// takes 11.63 instead of 10.72 ms.  The order of the binary tree stays.
__global__ void __launch_bounds__(64) k_collapse_level(const uint4* binary, const wide_item* items, uint32_t item_count, wide_item* next_items, uint32_t* counters, uint4* wide, uint32_t order_children) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= item_count) return;
	wide_item item = items[i];
	uint32_t position[4];
	uint4 node[4];
	position[0] = item.binary_position + 1u;
	node[0] = binary[position[0]];
	position[1] = (node[0].w & kLeafBit) ? item.binary_position + 2u : node[0].w;
	node[1] = binary[position[1]];
	uint32_t count = 2;
	while (count < 4) {
		// open the inner child with the largest box
		int widest = -1;
		float widest_area = -1.0f;
		for (uint32_t c = 0; c != count; ++c) {
			float area = quantized_half_area(node[c]);
			if (!(node[c].w & kLeafBit) && area > widest_area) { widest = (int) c; widest_area = area; }
		}
		if (widest < 0) break;
		uint32_t parent = position[widest];
		uint32_t left = parent + 1u;
		uint4 left_node = binary[left];
		uint32_t right = (left_node.w & kLeafBit) ? parent + 2u : left_node.w;
		position[widest] = left; node[widest] = left_node;
		position[count] = right; node[count] = binary[right];
		++count;
	}
	if (order_children) {
		// insertion sort of at most four children by the area of their boxes, descending
		for (uint32_t c = 1; c < count; ++c) {
			uint4 moved = node[c];
			uint32_t moved_position = position[c];
			float area = quantized_half_area(moved);
			uint32_t slot = c;
			while (slot > 0 && quantized_half_area(node[slot - 1]) < area) {
				node[slot] = node[slot - 1]; position[slot] = position[slot - 1];
				--slot;
			}
			node[slot] = moved; position[slot] = moved_position;
		}
	}
	uint32_t inner = 0;
	for (uint32_t c = 0; c != count; ++c) inner += (node[c].w & kLeafBit) ? 0u : 1u;
	uint32_t base = inner ? atomicAdd(&counters[1], inner) : 0u;
	uint32_t next_base = inner ? atomicAdd(&counters[0], inner) : 0u;
	// a ray that hits all children has them all on its stack before it takes the first one off again
	uint32_t need = item.need + count - 1u;
	atomicMax(&counters[2], item.need + count);
	uint32_t qx[4] = {kWideEmptyBox, kWideEmptyBox, kWideEmptyBox, kWideEmptyBox}, qy[4] = {kWideEmptyBox, kWideEmptyBox, kWideEmptyBox, kWideEmptyBox},
		qz[4] = {kWideEmptyBox, kWideEmptyBox, kWideEmptyBox, kWideEmptyBox}, link[4] = {kWideEmpty, kWideEmpty, kWideEmpty, kWideEmpty};
	uint32_t k = 0;
	for (uint32_t c = 0; c != count; ++c) {
		qx[c] = node[c].x; qy[c] = node[c].y; qz[c] = node[c].z;
		if (node[c].w & kLeafBit) link[c] = node[c].w;
		else {
			link[c] = base + k;
			wide_item child = {base + k, position[c], need};
			next_items[next_base + k] = child;
			++k;
		}
	}
	uint4* out = wide + 4 * (size_t) item.wide_index;
	out[0] = make_uint4(qx[0], qx[1], qx[2], qx[3]);
	out[1] = make_uint4(qy[0], qy[1], qy[2], qy[3]);
	out[2] = make_uint4(qz[0], qz[1], qz[2], qz[3]);
	out[3] = make_uint4(link[0], link[1], link[2], link[3]);
}

// The host learns how many nodes the next level has: the counters go to pinned host memory (a store
// of the device instead of a copy command per level and direction) and the per-level one starts over.
__global__ void __launch_bounds__(64) k_publish_counters(uint32_t* counters, uint32_t count, uint32_t* host_words) {
	if (blockIdx.x == 0 && threadIdx.x < count) {
		host_words[threadIdx.x] = counters[threadIdx.x];
		if (threadIdx.x == 0) counters[0] = 0u;
		__threadfence_system();
	}
}

}  // namespace

// Replaces structure->nodes (fp32 threaded nodes on the device) by the quantised nodes
static int quantize_nodes(acceleration_structure_t* structure, const device_t* device) {
	hipStream_t stream = (hipStream_t) device->stream;
	float root[8];
	if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(root, structure->nodes, sizeof(root), hipMemcpyDeviceToHost) != hipSuccess) return 1;
	const float lo[3] = {root[0], root[1], root[2]}, hi[3] = {root[3], root[4], root[5]};
	for (int j = 0; j != 3; ++j) {
		// one spare cell on either side keeps every box strictly inside [0, kGridMax]
		float extent = fmaxf(hi[j] - lo[j], 1.0e-20f);
		float cell = extent / (kGridMax - 2.0f);
		structure->grid_origin[j] = lo[j] - cell;
		structure->grid_inverse_cell[j] = 1.0f / cell;
	}
	void* quantized = NULL;
	if (hipMalloc(&quantized, sizeof(uint4) * (size_t) structure->node_count) != hipSuccess) return 1;
	k_quantize_nodes<<<(structure->node_count + 255) / 256, 256, 0, stream>>>(structure->node_count, (const float4*) structure->nodes,
		f3{structure->grid_origin[0], structure->grid_origin[1], structure->grid_origin[2]},
		f3{structure->grid_inverse_cell[0], structure->grid_inverse_cell[1], structure->grid_inverse_cell[2]}, (uint4*) quantized);
	if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) { (void) hipFree(quantized); return 1; }
	(void) hipFree(structure->nodes);
	structure->nodes = quantized;
	return 0;
}

static __global__ void k_empty() {}
extern "C" int vkr_launch_empty_kernel(void* stream) {
	k_empty<<<1, 64, 0, (hipStream_t) stream>>>();
	return hipGetLastError() != hipSuccess;
}

extern "C" void vkr_destroy_acceleration_structure(acceleration_structure_t* structure, const device_t* device) {
	(void) device;
	if (structure->triangle_vertices) (void) hipFree(structure->triangle_vertices);
	if (structure->nodes) (void) hipFree(structure->nodes);
	if (structure->wide_nodes) (void) hipFree(structure->wide_nodes);
	memset(structure, 0, sizeof(*structure));
}

// Binary quantised nodes -> four-wide nodes, one level per launch (a level's node count is only
// known when the level before it has been written).  Leaves structure->wide_nodes NULL (the
// kernels then walk the binary tree) if the tree is a single leaf or a ray could need a deeper
// stack than the kernels provide.
static int collapse_to_wide(acceleration_structure_t* structure, const device_t* device, uint32_t* host_words) {
	uint32_t triangle_count = (structure->node_count + 1) / 2;
	if (triangle_count < 2) return 0;
	hipStream_t stream = (hipStream_t) device->stream;
	// VKR_WIDE_CHILD_ORDER=1: children largest box first (an experiment of round 4, see k_collapse_level); default:
	// the order of the binary tree
	const char* order_knob = getenv("VKR_WIDE_CHILD_ORDER");
	const uint32_t order_children = (order_knob && order_knob[0] == '1') ? 1u : 0u;
	wide_item* items[2] = {NULL, NULL};
	uint32_t* counters = NULL;
	uint8_t* arena = NULL;
	uint4* wide = NULL;
	int failed = 1;
	uint32_t host_counters[3] = {0, 1, 0};
	const size_t item_bytes = (sizeof(wide_item) * (size_t) triangle_count + 255) & ~(size_t) 255;
	do {
		HIP_OK_BREAK(hipMalloc(&arena, 2 * item_bytes + 256));
		items[0] = (wide_item*) arena;
		items[1] = (wide_item*) (arena + item_bytes);
		counters = (uint32_t*) (arena + 2 * item_bytes);
		HIP_OK_BREAK(hipMalloc(&wide, sizeof(uint4) * 4 * (size_t) (triangle_count - 1)));
		wide_item root = {0u, 0u, 0u};
		HIP_OK_BREAK(hipMemcpyAsync(items[0], &root, sizeof(root), hipMemcpyHostToDevice, stream));
		HIP_OK_BREAK(hipMemcpyAsync(counters, host_counters, sizeof(host_counters), hipMemcpyHostToDevice, stream));
		uint32_t item_count = 1, level = 0;
		bool ok = true;
		while (item_count && ok) {
			k_collapse_level<<<(item_count + 63) / 64, 64, 0, stream>>>((const uint4*) structure->nodes, items[level & 1], item_count, items[(level + 1) & 1], counters, wide, order_children);
			k_publish_counters<<<1, 64, 0, stream>>>(counters, 3u, host_words);
			ok = hipStreamSynchronize(stream) == hipSuccess && ++level < 4096;
			for (int i = 0; i != 3; ++i) host_counters[i] = ((volatile uint32_t*) host_words)[i];
			item_count = host_counters[0];
		}
		if (!ok || hipGetLastError() != hipSuccess) break;
		failed = 0;
	} while (0);
	(void) hipFree(arena);
	if (failed || host_counters[2] > kWideStackMax) {
		if (!failed) printf("The four-wide BVH would need a stack of %u entries per ray (at most %u are provided); shadow rays walk the binary tree.\n", host_counters[2], kWideStackMax);
		(void) hipFree(wide);
		return failed;
	}
	structure->wide_nodes = wide;
	structure->wide_node_count = host_counters[1];
	structure->wide_stack_need = host_counters[2];
	return 0;
}

// The binned SAH build by HIP kernels (see above); leaves fp32 threaded nodes in structure->nodes
static int build_sah_on_device(acceleration_structure_t* structure, const device_t* device, const build_params& p, uint32_t* host_words) {
	hipStream_t stream = (hipStream_t) device->stream;
	uint32_t n = p.triangle_count;
	uint32_t total_nodes = 2 * n - 1;
	// at most n / 2 nodes with two or more triangles are open at a time
	uint32_t open_capacity = n / 2 + 1;
	sah_open_node* open[2] = {NULL, NULL};
	sah_bin* bins = NULL;
	uint32_t *triangle_node = NULL, *counters = NULL;
	// the temporaries of the build in ONE allocation (five hipMalloc / hipFree pairs of up to 90 MB were a
	// third of the build's wall-clock time)
	uint8_t* arena = NULL;
	auto aligned = [](size_t bytes) { return (bytes + 255) & ~(size_t) 255; };
	const size_t open_bytes = aligned(sizeof(sah_open_node) * (size_t) open_capacity), bin_bytes = aligned(sizeof(sah_bin) * 3 * kSahBinCount * (size_t) open_capacity),
		triangle_node_bytes = aligned(sizeof(uint32_t) * (size_t) n);
	int failed = 1;
	do {
		HIP_OK_BREAK(hipMalloc(&structure->triangle_vertices, sizeof(float4) * 3 * (size_t) n));
		HIP_OK_BREAK(hipMalloc(&structure->nodes, sizeof(float4) * 2 * (size_t) total_nodes));
		HIP_OK_BREAK(hipMalloc(&arena, 2 * open_bytes + bin_bytes + triangle_node_bytes + 256));
		open[0] = (sah_open_node*) arena;
		open[1] = (sah_open_node*) (arena + open_bytes);
		bins = (sah_bin*) (arena + 2 * open_bytes);
		triangle_node = (uint32_t*) (arena + 2 * open_bytes + bin_bytes);
		counters = (uint32_t*) (arena + 2 * open_bytes + bin_bytes + triangle_node_bytes);
		uint32_t blocks = (n + 255) / 256;
		k_sah_reset_root<<<1, 64, 0, stream>>>(open[0], n, counters);
		k_sah_init<<<blocks, 256, 0, stream>>>(p, triangle_node, open[0], (float4*) structure->nodes, (float4*) structure->triangle_vertices);
		uint32_t open_count = n > 1 ? 1u : 0u, level = 0;
		bool ok = true;
		while (open_count && ok) {
			sah_open_node *now = open[level & 1], *next = open[(level + 1) & 1];
			uint32_t bin_count = open_count * 3u * kSahBinCount;
			k_sah_clear_bins<<<(bin_count + 255) / 256, 256, 0, stream>>>(bins, bin_count);
			if (open_count <= kSahSharedNodes) k_sah_bin_shared<<<blocks, 256, 0, stream>>>(p, triangle_node, now, open_count, bins);
			else k_sah_bin<<<blocks, 256, 0, stream>>>(p, triangle_node, now, bins);
			k_sah_split<<<(open_count + kSplitNodesPerGroup - 1u) / kSplitNodesPerGroup, 256, 0, stream>>>(now, open_count, bins, next, counters, (float4*) structure->nodes, p.pad);
			k_sah_assign<<<blocks, 256, 0, stream>>>(p, triangle_node, now, next, (float4*) structure->nodes, (float4*) structure->triangle_vertices);
			k_publish_counters<<<1, 64, 0, stream>>>(counters, 1u, host_words);
			ok = hipStreamSynchronize(stream) == hipSuccess;
			uint32_t next_count = ((volatile uint32_t*) host_words)[0];
			ok = ok && next_count <= open_capacity && ++level < 4096;
			open_count = next_count;
		}
		if (!ok || hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) break;
		structure->node_count = total_nodes;
		structure->root = 0;
		failed = 0;
	} while (0);
	(void) hipFree(arena);
	return failed;
}

static int build_lbvh_on_device(acceleration_structure_t* structure, const device_t* device, const build_params& p) {
	hipStream_t stream = (hipStream_t) device->stream;
	uint32_t n = p.triangle_count;
	uint32_t inner_count = n > 1 ? n - 1 : 1;
	uint32_t total_nodes = 2 * n - 1;
	bvh_build_node* build_nodes = NULL;
	uint64_t *keys = NULL, *sorted_keys = NULL;
	float *lo = NULL, *hi = NULL;
	uint32_t *leaf_parents = NULL, *arrivals = NULL;
	void* sort_storage = NULL;
	size_t sort_bytes = 0;
	int failed = 1;
	do {
		if (hipMalloc(&structure->triangle_vertices, sizeof(float4) * 3 * (size_t) n) != hipSuccess) break;
		if (hipMalloc(&structure->nodes, sizeof(float4) * 2 * (size_t) total_nodes) != hipSuccess) break;
		if (hipMalloc(&build_nodes, sizeof(bvh_build_node) * (size_t) inner_count) != hipSuccess) break;
		if (hipMalloc(&keys, sizeof(uint64_t) * n) != hipSuccess || hipMalloc(&sorted_keys, sizeof(uint64_t) * n) != hipSuccess) break;
		if (hipMalloc(&lo, sizeof(float) * 3 * (2 * (size_t) n)) != hipSuccess || hipMalloc(&hi, sizeof(float) * 3 * (2 * (size_t) n)) != hipSuccess) break;
		if (hipMalloc(&leaf_parents, sizeof(uint32_t) * n) != hipSuccess || hipMalloc(&arrivals, sizeof(uint32_t) * inner_count) != hipSuccess) break;
		if (hipMemsetAsync(arrivals, 0, sizeof(uint32_t) * inner_count, stream) != hipSuccess) break;
		if (hipMemsetAsync(build_nodes, 0, sizeof(bvh_build_node) * (size_t) inner_count, stream) != hipSuccess) break;
		uint32_t blocks = (n + 255) / 256;
		k_morton_keys<<<blocks, 256, 0, stream>>>(p, keys);
		if (hipcub::DeviceRadixSort::SortKeys(NULL, sort_bytes, keys, sorted_keys, (int) n, 0, 62, stream) != hipSuccess) break;
		if (hipMalloc(&sort_storage, sort_bytes ? sort_bytes : 1) != hipSuccess) break;
		if (hipcub::DeviceRadixSort::SortKeys(sort_storage, sort_bytes, keys, sorted_keys, (int) n, 0, 62, stream) != hipSuccess) break;
		k_write_leaves<<<blocks, 256, 0, stream>>>(p, sorted_keys, (float4*) structure->triangle_vertices, lo, hi);
		if (n > 1) {
			k_build_hierarchy<<<(n - 1 + 255) / 256, 256, 0, stream>>>(sorted_keys, (int) n, build_nodes, leaf_parents);
			k_refit<<<blocks, 256, 0, stream>>>((int) n, build_nodes, leaf_parents, lo, hi, arrivals);
		}
		k_thread_nodes<<<(total_nodes + 255) / 256, 256, 0, stream>>>((int) n, build_nodes, leaf_parents, lo, hi, (float4*) structure->nodes);
		structure->root = 0;
		if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) break;
		structure->node_count = total_nodes;
		failed = 0;
	} while (0);
	(void) hipFree(keys); (void) hipFree(sorted_keys); (void) hipFree(lo); (void) hipFree(hi);
	(void) hipFree(leaf_parents); (void) hipFree(arrivals); (void) hipFree(sort_storage); (void) hipFree(build_nodes);
	return failed;
}

static int build_sah_on_host(acceleration_structure_t* structure, const device_t* device, const mesh_t* mesh, float pad) {
	// surface-area heuristic on the host, then one upload
	float *host_nodes = NULL, *host_triangles = NULL;
	uint32_t node_count = 0;
	if (vkr_build_sah_bvh_host(mesh, pad, &host_nodes, &host_triangles, &node_count)) {
		printf("Building the SAH BVH over %llu triangles on the host failed (out of memory or no host copy of the positions).\n", (unsigned long long) mesh->triangle_count);
		return 1;
	}
	int upload_failed = vkr_device_upload(&structure->nodes, device, host_nodes, sizeof(float) * 8 * (size_t) node_count, "BVH nodes")
		|| vkr_device_upload(&structure->triangle_vertices, device, host_triangles, sizeof(float) * 12 * (size_t) mesh->triangle_count, "BVH triangles");
	free(host_nodes);
	free(host_triangles);
	structure->node_count = node_count;
	structure->root = 0;
	return upload_failed;
}

extern "C" int vkr_build_acceleration_structure(acceleration_structure_t* structure, const device_t* device, const mesh_t* mesh, int builder) {
	memset(structure, 0, sizeof(*structure));
	if (builder <= (int) acceleration_structure_none || builder >= (int) acceleration_structure_builder_count) {
		printf("Invalid acceleration structure builder %d.\n", builder);
		return 1;
	}
	if (mesh->triangle_count == 0 || mesh->triangle_count > 0x7FFFFFFFull) {
		printf("The BVH builders support 1 to 2^31 triangles, the mesh has %llu.\n", (unsigned long long) mesh->triangle_count);
		return 1;
	}
	hipStream_t stream = (hipStream_t) device->stream;
	uint32_t n = (uint32_t) mesh->triangle_count;
	build_params p;
	p.quantized_positions = (const uint2*) mesh->positions;
	p.triangle_count = n;
	p.fragment_triangle = NULL;
	p.fragment_boxes = NULL;
	float extent = 0.0f;
	for (int j = 0; j != 3; ++j) {
		p.factor[j] = mesh->dequantization_factor[j];
		p.summand[j] = mesh->dequantization_summand[j];
		// largest coordinate magnitude and extent of the quantisation grid
		extent = fmaxf(extent, 2097152.0f * fabsf(mesh->dequantization_factor[j]));
		extent = fmaxf(extent, fmaxf(fabsf(mesh->dequantization_summand[j]), fabsf(mesh->dequantization_summand[j] + 2097152.0f * mesh->dequantization_factor[j])));
	}
	// Conservative padding.  The slab test evaluates plane * (1/d) - o * (1/d) with a
	// 1-ulp reciprocal and two roundings, i.e. it misplaces a plane by about
	// |o| 2^-23; the triangle test may accept hits about as far outside the triangle.
	// 2e-6 of the largest coordinate is sixteen times that.  (The padding must stay
	// far below t_min = 1e-3: a ray leaving a flat floor would otherwise start inside
	// the padded boxes of the floor and walk down to its own triangle.)
	p.pad = 2.0e-6f * extent;
	if (hipStreamSynchronize(stream) != hipSuccess) return 1;
	struct timespec start, end;
	clock_gettime(CLOCK_MONOTONIC, &start);
	// VKR_BVH_BUILD_TRACE=1 prints the wall-clock time of the three phases (diagnostics)
	const bool trace = getenv("VKR_BVH_BUILD_TRACE") != NULL;
	struct timespec phase[3] = {start, start, start};
	// (pinned: where the level loops of the build read their counters, k_publish_counters)
	uint32_t* host_words = NULL;
	if (hipHostMalloc(&host_words, 64) != hipSuccess) {
		printf("Failed to allocate pinned host memory for the BVH build.\n");
		return 1;
	}
	// device SAH: long thin triangles that lie diagonally in their boxes become several primitives ("fragments" above;
	// VKR_BVH_SPLIT_TRIANGLES=0 builds over whole triangles as until round 3)
	void* fragment_memory = NULL;
	const char* split_knob = getenv("VKR_BVH_SPLIT_TRIANGLES");
	// (a triangle becomes at most 16 fragments: up to 2^31 / 16 triangles the 32-bit prefix sum of the counts cannot wrap;
	// beyond that the mesh is built unsplit)
	if (builder == (int) acceleration_structure_sah_device && n > 1 && n <= 0x7FFFFFFFu / 16u && !(split_knob && split_knob[0] == '0')) {
		uint32_t* counts = NULL;
		void* scan_storage = NULL;
		size_t scan_bytes = 0;
		const float least_extent = extent * (1.0f / 512.0f);
		uint32_t blocks = (n + 255) / 256, total = n;
		bool ok = hipMalloc(&counts, sizeof(uint32_t) * 2 * (size_t) n) == hipSuccess;
		uint32_t* first = counts ? counts + n : NULL;
		// A mesh of nothing but slivers would grow leaves, nodes and triangle_vertices sixteenfold.  Beyond four leaves per
		// triangle on average the cap per triangle is halved until the mesh fits (16, 8, 4, 2: with two fragments a mesh at
		// most doubles), so that the longest slivers - the ones for which splitting pays most - are still split (ADVICE round
		// 5: until then such a mesh was built unsplit, a performance cliff exactly where splitting matters).
		uint32_t most = kMostFragments;
		while (ok) {
			k_count_fragments<<<blocks, 256, 0, stream>>>(p, least_extent, most, counts);
			if (!scan_storage)
				ok = hipcub::DeviceScan::ExclusiveSum(NULL, scan_bytes, counts, first, (int) n, stream) == hipSuccess
					&& hipMalloc(&scan_storage, scan_bytes ? scan_bytes : 1) == hipSuccess;
			ok = ok && hipcub::DeviceScan::ExclusiveSum(scan_storage, scan_bytes, counts, first, (int) n, stream) == hipSuccess;
			uint32_t last[2] = {0, 0};
			ok = ok && hipMemcpyAsync(&last[0], counts + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess
				&& hipMemcpyAsync(&last[1], first + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess
				&& hipStreamSynchronize(stream) == hipSuccess;
			if (!ok) break;
			total = last[0] + last[1];
			if ((uint64_t) total <= 4ull * n || most <= 2u) break;
			most /= 2u;
		}
		if (ok && most != kMostFragments && trace)
			printf("Long thin triangles are split into at most %u fragments (16 would turn %u triangles into more than %llu leaves).\n", most, n, 4ull * n);
		if (ok && total > n && (uint64_t) total <= 4ull * n) {
			// fragment_triangle[total], then fragment_boxes[6 total]
			ok = hipMalloc(&fragment_memory, sizeof(uint32_t) * (size_t) total + sizeof(float) * 6 * (size_t) total + 16) == hipSuccess;
			if (ok) {
				uint32_t* fragment_triangle = (uint32_t*) fragment_memory;
				float* fragment_boxes = (float*) (fragment_triangle + total);
				k_write_fragments<<<blocks, 256, 0, stream>>>(p, least_extent, most, first, fragment_triangle, fragment_boxes);
				ok = hipStreamSynchronize(stream) == hipSuccess && hipGetLastError() == hipSuccess;
				if (ok) {
					p.fragment_triangle = fragment_triangle;
					p.fragment_boxes = fragment_boxes;
					p.triangle_count = total;
				}
			}
		}
		(void) hipFree(scan_storage);
		(void) hipFree(counts);
		if (!ok) {
			printf("Splitting the long thin triangles of the mesh failed; the BVH is built over whole triangles.\n");
			(void) hipFree(fragment_memory);
			fragment_memory = NULL;
			p.fragment_triangle = NULL; p.fragment_boxes = NULL; p.triangle_count = n;
			(void) hipGetLastError();
		}
	}
	int failed = builder == (int) acceleration_structure_sah_host ? build_sah_on_host(structure, device, mesh, p.pad)
		: (builder == (int) acceleration_structure_lbvh_device ? build_lbvh_on_device(structure, device, p) : build_sah_on_device(structure, device, p, host_words));
	(void) hipFree(fragment_memory);
	if (trace) clock_gettime(CLOCK_MONOTONIC, &phase[0]);
	if (!failed) {
		failed = quantize_nodes(structure, device);
		if (failed) printf("Quantising the BVH nodes failed.\n");
	}
	if (trace) clock_gettime(CLOCK_MONOTONIC, &phase[1]);
	if (!failed) failed = collapse_to_wide(structure, device, host_words);
	(void) hipHostFree(host_words);
	if (trace) {
		clock_gettime(CLOCK_MONOTONIC, &phase[2]);
		auto ms = [](const struct timespec& a, const struct timespec& b) { return (double) (b.tv_sec - a.tv_sec) * 1.0e3 + (double) (b.tv_nsec - a.tv_nsec) * 1.0e-6; };
		printf("BVH build over %u triangles: tree %.3f ms, quantisation %.3f ms, four-wide collapse %.3f ms\n", n, ms(start, phase[0]), ms(phase[0], phase[1]), ms(phase[1], phase[2]));
	}
	if (failed) {
		printf("Building the BVH over %u triangles failed: %s\n", n, hipGetErrorString(hipGetLastError()));
		vkr_destroy_acceleration_structure(structure, device);
		return 1;
	}
	clock_gettime(CLOCK_MONOTONIC, &end);
	structure->builder = (uint32_t) builder;
	structure->leaf_count = (structure->node_count + 1u) / 2u;
	structure->build_milliseconds = (float) ((double) (end.tv_sec - start.tv_sec) * 1.0e3 + (double) (end.tv_nsec - start.tv_nsec) * 1.0e-6);
	return 0;
}
