// LBVH construction on the GPU (Karras 2012: Morton codes -> radix sort ->
// binary radix tree -> bottom-up refit -> depth-first "threaded" layout).  Replaces create_acceleration_structure
// of the reference (src/scene.c:142-406), which hands the same de-quantised
// triangle soup to the Vulkan driver.  Runs once per scene.
#include "lbvh.h"
#include "host/vkr_internal.h"
#include <hipcub/hipcub.hpp>

using namespace vkr;

namespace {

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct build_params {
	const uint2* quantized_positions;
	uint32_t triangle_count;
	float factor[3], summand[3];
	float pad;
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
// boxes rounded outwards on the 16-bit grid (layout: lbvh.h)
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
		q0 = fminf(fmaxf(q0, 0.0f), 65535.0f);
		q1 = fminf(fmaxf(q1, 0.0f), 65535.0f);
		packed[j] = (uint32_t) q0 | ((uint32_t) q1 << 16);
	}
	uint32_t skip = __float_as_uint(b.z), leaf = __float_as_uint(b.w);
	quantized[id] = make_uint4(packed[0], packed[1], packed[2], leaf != kNoLeaf ? (kLeafBit | leaf) : skip);
}

}  // namespace

// Replaces structure->nodes (fp32 threaded nodes on the device) by the quantised nodes
static int quantize_nodes(acceleration_structure_t* structure, const device_t* device) {
	hipStream_t stream = (hipStream_t) device->stream;
	float root[8];
	if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(root, structure->nodes, sizeof(root), hipMemcpyDeviceToHost) != hipSuccess) return 1;
	const float lo[3] = {root[0], root[1], root[2]}, hi[3] = {root[3], root[4], root[5]};
	for (int j = 0; j != 3; ++j) {
		// one spare cell on either side keeps every box strictly inside [0, 65535]
		float extent = fmaxf(hi[j] - lo[j], 1.0e-20f);
		float cell = extent / 65533.0f;
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

extern "C" void vkr_destroy_acceleration_structure(acceleration_structure_t* structure, const device_t* device) {
	(void) device;
	if (structure->triangle_vertices) (void) hipFree(structure->triangle_vertices);
	if (structure->nodes) (void) hipFree(structure->nodes);
	memset(structure, 0, sizeof(*structure));
}

extern "C" int vkr_build_acceleration_structure(acceleration_structure_t* structure, const device_t* device, const mesh_t* mesh, int builder) {
	memset(structure, 0, sizeof(*structure));
	const char* requested = getenv("VKR_BVH_BUILDER");
	if (requested && strcmp(requested, "lbvh") == 0) builder = 2;
	if (requested && strcmp(requested, "sah") == 0) builder = 1;
	if (mesh->triangle_count > 0x7FFFFFFFull) {
		printf("The LBVH supports at most 2^31 triangles.\n");
		return 1;
	}
	hipStream_t stream = (hipStream_t) device->stream;
	uint32_t n = (uint32_t) mesh->triangle_count;
	build_params p;
	p.quantized_positions = (const uint2*) mesh->positions;
	p.triangle_count = n;
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
	if (builder != 2) {
		// surface-area heuristic on the host, then one upload
		float *host_nodes = NULL, *host_triangles = NULL;
		uint32_t node_count = 0;
		if (vkr_build_sah_bvh_host(mesh, p.pad, &host_nodes, &host_triangles, &node_count)) {
			printf("Building the SAH BVH over %u triangles failed (out of memory or no host copy of the positions).\n", n);
			return 1;
		}
		int upload_failed = vkr_device_upload(&structure->nodes, device, host_nodes, sizeof(float) * 8 * (size_t) node_count, "BVH nodes")
			|| vkr_device_upload(&structure->triangle_vertices, device, host_triangles, sizeof(float) * 12 * (size_t) n, "BVH triangles");
		free(host_nodes);
		free(host_triangles);
		if (upload_failed) {
			vkr_destroy_acceleration_structure(structure, device);
			return 1;
		}
		structure->node_count = node_count;
		structure->root = 0;
		if (quantize_nodes(structure, device)) {
			printf("Quantising the BVH nodes failed.\n");
			vkr_destroy_acceleration_structure(structure, device);
			return 1;
		}
		return 0;
	}
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
		structure->triangle_indices = NULL;
		failed = quantize_nodes(structure, device);
	} while (0);
	(void) hipFree(keys); (void) hipFree(sorted_keys); (void) hipFree(lo); (void) hipFree(hi);
	(void) hipFree(leaf_parents); (void) hipFree(arrivals); (void) hipFree(sort_storage); (void) hipFree(build_nodes);
	if (failed) {
		printf("Building the LBVH over %u triangles failed: %s\n", n, hipGetErrorString(hipGetLastError()));
		vkr_destroy_acceleration_structure(structure, device);
	}
	return failed;
}
