// The wavefront kernels behind shade_pixels (shading_kernel.h): persistent waves that trace the
// queued shadow rays, and the kernel that replays every pixel's sums in program order.
// Instantiated once, in shading_pass.hip.  Ray-query contract: reference
// src/shaders/shading_pass.frag.glsl:120-138 (opaque, any hit terminates, t in [1e-3, t_max]).
#pragma once
#include "shading_kernel.h"

namespace vkr {

// ---- wavefront: trace and resolve (instantiated once, in shading_pass.hip) -------------------

// Persistent tracing waves per SIMD that the wide kernel is compiled for (register budget 512 / n)
#ifndef VKR_WIDE_TRACE_WAVES
#define VKR_WIDE_TRACE_WAVES 8
#endif
constexpr uint32_t kWideTraceWaves = VKR_WIDE_TRACE_WAVES;
// Idle lanes of a tracing wave before the next rays are handed out (trace_shadow_rays_wide, `refill_lanes`; the
// run-time knob is VKR_WIDE_REFILL, 0 = always a batch at a time)
#ifndef VKR_WIDE_REFILL_LANES
#define VKR_WIDE_REFILL_LANES 16
#endif
constexpr uint32_t kWideRefillLanes = VKR_WIDE_REFILL_LANES;
// ... by a wave whose batches kept less than this share (in 1/256) of its lanes busy (VKR_WIDE_REFILL_BELOW)
constexpr uint32_t kWideRefillBelow = 166;

// Experiment of round 3 (north_star: "LDS-staged BVH node packets"; profiles/r03_trace.md has the
// measurement): the first VKR_LDS_TOP_NODES nodes of the four-wide tree - its top levels, breadth
// first: 1 + 4 + 16 = 21 nodes are three levels - are copied into LDS by every workgroup and fetched
// from there (flat loads: the address decides between LDS and global memory).  0: off (the default).
#ifndef VKR_TRACE_BLOCKER_CACHE
#define VKR_TRACE_BLOCKER_CACHE 1
#endif
#ifndef VKR_LDS_TOP_NODES
#define VKR_LDS_TOP_NODES 0
#endif

// What the tracing kernels read: the queues that the shading kernel filled (shade_params has the
// same pointers, writable)
struct ray_stream {
	const float4* directions;   // [queue][slot]: direction, t_max
	const uint32_t* records;    // [queue][slot]: thread | code cursor << thread_bits, or kNullRay
	const float4* origins;      // [thread]: where all rays of that pixel start
	const uint32_t* sizes;      // slots used per queue
	uint32_t capacity, thread_bits, thread_count;
};

// Wave-uniform bookkeeping of the persistent tracing waves: which chunk of which queue the wave
// works on.  Queues 64 x ... 64 x + 63 are served only by workgroups that run on XCD x (workgroup b
// is placed on XCD b % 8; used for cache affinity only, never for correctness), so counters and
// cursors never bounce between the eight L2s.  A wave claims a chunk of rays with one atomic on its
// XCD's cursor and locates it with a wave-wide prefix sum over the 64 queue sizes of that XCD.
// (Tried in round 3 and dropped: waves of an XCD that has run dry helping the next XCD.  The scan had
// to be redone per claim - its registers are needed in the walk - and the kernel got slower,
// 610 -> 659 us alone at config 3: the XCDs finish within a few percent of each other anyway.)
struct chunk_cursor {
	uint32_t xcd, my_size, my_chunks, exclusive, inclusive, total_chunks, chunk_size;
	size_t chunk_first;  // slot (over all queues) of the first ray of the claimed chunk
	uint32_t chunk_count, chunk_next;
	bool chunks_left;
};

VKR_DEV chunk_cursor make_chunk_cursor(const ray_stream& rays, uint32_t waves_per_workgroup) {
	chunk_cursor c;
	const uint32_t lane = threadIdx.x & 63u;
	// (the host launches multiples of eight workgroups: fewer would leave the queues of an XCD unserved)
	c.xcd = blockIdx.x & 7u;
	c.my_size = rays.sizes[c.xcd * 64u + lane];
	// chunk size: large enough to keep the atomics rare, small enough that every resident
	// wave of this XCD gets about two chunks (few rays: config 2 queues 0.9 M, config 3 29 M)
	uint32_t xcd_rays = c.my_size;
#pragma unroll
	for (int offset = 32; offset > 0; offset >>= 1) xcd_rays += __shfl_xor(xcd_rays, offset);
	const uint32_t xcd_waves = max(1u, (gridDim.x / 8u) * waves_per_workgroup);
	c.chunk_size = min(kRayChunk, max(64u, ((xcd_rays / (2u * xcd_waves) + 63u) / 64u) * 64u));
	c.my_chunks = (c.my_size + c.chunk_size - 1u) / c.chunk_size;
	c.inclusive = c.my_chunks;
#pragma unroll
	for (int offset = 1; offset < 64; offset <<= 1) {
		uint32_t other = __shfl_up(c.inclusive, offset);
		if (lane >= (uint32_t) offset) c.inclusive += other;
	}
	c.exclusive = c.inclusive - c.my_chunks;
	c.total_chunks = __shfl(c.inclusive, 63);
	c.chunk_first = 0;
	c.chunk_count = c.chunk_next = 0;
	c.chunks_left = true;
	return c;
}

// Claims the next chunk of this XCD for the wave; false when the queues are exhausted
VKR_DEV bool claim_chunk(chunk_cursor& c, const ray_stream& rays, uint32_t* work_cursors) {
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t chunk = 0;
	if (lane == 0) chunk = atomicAdd(work_cursors + c.xcd * kCursorStride, 1u);
	chunk = __builtin_amdgcn_readfirstlane(chunk);
	if (chunk >= c.total_chunks) { c.chunks_left = false; return false; }
	uint64_t owner = __ballot(c.my_chunks != 0 && c.exclusive <= chunk && chunk < c.inclusive);
	int owner_lane = __ffsll((unsigned long long) owner) - 1;
	uint32_t queue = c.xcd * 64u + (uint32_t) owner_lane;
	uint32_t first = (chunk - __shfl(c.exclusive, owner_lane)) * c.chunk_size;
	uint32_t size = __shfl(c.my_size, owner_lane);
	c.chunk_first = (size_t) queue * rays.capacity + first;
	c.chunk_count = min(c.chunk_size, size - first);
	c.chunk_next = 0;
	return true;
}

// Persistent waves trace the queued shadow rays on the binary tree.  Few registers, no LDS, no scratch:
// 8 waves per SIMD hide the latency of the dependent node fetches.  Lanes whose ray has finished are
// refilled from the chunk while the others keep walking (shadow rays differ a lot in length), so lanes
// stay busy.  A ray that reaches the light flips its term's code to kCodeVisible.
__global__ void __launch_bounds__(256) trace_shadow_rays(bvh_view bvh, ray_stream rays, uint32_t* work_cursors, uint8_t* codes, uint32_t refill_threshold) {
	chunk_cursor cursor = make_chunk_cursor(rays, 4u);
	const uint32_t end = bvh.node_count;
	// per-lane ray; a lane without a ray has the cursor kIdle (the walk and the ballots test the
	// cursor itself: a separate flag costs two more instructions per step)
	constexpr uint32_t kIdle = 0xFFFFFFFFu;
	f3 o = mk3(0.0f, 0.0f, 0.0f), d = o;
	grid_ray ray = {o, o};
	float t_max = 0.0f;
	uint32_t node = kIdle;
	size_t code_index = 0;
	while (true) {
		// ---- hand new rays to idle lanes ------------------------------------------------
		uint64_t idle = __ballot(node == kIdle);
		while (idle != 0 && (cursor.chunk_next < cursor.chunk_count || cursor.chunks_left)) {
			if (cursor.chunk_next >= cursor.chunk_count && !claim_chunk(cursor, rays, work_cursors)) break;
			uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
			uint32_t index = cursor.chunk_next + rank;
			if (node == kIdle && index < cursor.chunk_count) {
				float4 a = rays.directions[cursor.chunk_first + index];
				uint32_t record = rays.records[cursor.chunk_first + index];
				// (a slot that the shading wave reserved and did not need holds kNullRay)
				if (record != kNullRay) {
					uint32_t tid = ray_record_thread(rays.thread_bits, record);
					float4 b = rays.origins[tid];
					o = mk3(b.x, b.y, b.z); d = mk3(a.x, a.y, a.z); t_max = a.w;
					code_index = code_slot(rays.thread_count, ray_record_cursor(rays.thread_bits, record), tid);
					ray = make_grid_ray(bvh, o, d);
					node = 0;
					if (!(t_max >= 1.0e-3f)) {
						// empty interval: nothing can block the ray (same rule as any_hit)
						codes[code_index] = (uint8_t) kCodeVisible;
						node = kIdle;
					}
				}
			}
			cursor.chunk_next += (uint32_t) __popcll((unsigned long long) idle);
			idle = __ballot(node == kIdle);
		}
		uint64_t busy = __ballot(node != kIdle);
		if (busy == 0) break;
		// ---- walk until too many lanes have run dry (then refill) -------------------------
		bool may_refill = cursor.chunk_next < cursor.chunk_count || cursor.chunks_left;
		do {
			if (node != kIdle) {
				uint4 n = bvh.nodes[node];
				bool is_leaf = (n.w & kLeafBit) != 0;
				bool hit = ray_box(n, ray, 1.0e-3f, t_max);
				bool blocked = false;
				if (hit && is_leaf) {
					const float4* t = bvh.triangles + 3 * (size_t) (n.w & ~kLeafBit);
					float dist;
					blocked = ray_triangle<false>(t[0], t[1], t[2], o, d, 1.0e-3f, t_max, dist);
				}
				node = (hit || is_leaf) ? node + 1 : n.w;
				if (!blocked && node >= end) codes[code_index] = (uint8_t) kCodeVisible;
				if (blocked || node >= end) node = kIdle;
			}
			busy = __ballot(node != kIdle);
		} while (busy != 0 && (!may_refill || __popcll((unsigned long long) busy) > refill_threshold));
	}
}

// The same persistent scheme on the four-wide tree (lbvh.h "wide BVH"): a visit fetches one
// 64-byte node and tests its four boxes (wide_ray_box: 6 v_perm_b32, 6 FMAs, 2 x min3 / max3, one
// compare per box, no branches - an absent child has a box that cannot be hit).  All children that
// were hit go to the lane's stack and the top one is taken off again: kWideStackLds entries in LDS,
// [entry][thread], written without a condition (the cursor only advances for a hit), which needs
// no selection logic at all.  A lane within four entries of the end of the LDS part takes the slow
// path, which decides per entry between LDS and `spill` ([entry][global thread], sized by the
// build's worst case acceleration_structure_t.wide_stack_need).  A lane whose next item is a
// triangle waits until `leaf_batch` lanes of the wave have one (or no lane has a node left), so
// that the triangle test runs with many lanes: the two kinds of work do not share every step.
// Rays are taken 64 at a time, one per lane: the 20 bytes of a ray are two coalesced wave loads,
// and the loads of the NEXT batch are issued before the walk of the current one begins, so that the
// cold read of the ray stream (nothing else in this kernel misses the L2 as often) hides behind the
// walk instead of standing between two batches; only the origin - a read of the pixel's position
// that several consecutive batches share - is fetched when a batch starts.  (What that bought: 629 ->
// 610 us alone at config 3, together with the 20-byte records.  The kernel is bound by the issue of
// its box tests and by the dependent node fetches, not by this read: profiles/r03_trace.md.)
// THREADS: 256, or 64 - one wave per workgroup, which then leaves on its own when it finds no more
// chunks and fits into whatever a SIMD has free.  Measured (profiles/): with the three-wave shading
// kernels and many rays (config 3) single waves overlap the neighbouring frame's shading better
// (-2.6 %, config 4 -0.9 %); with few rays (config 2) or two-wave shading kernels launching four
// times as many workgroups costs 2 % instead.  The host picks (shading_pass.hip).
// Handing rays to idle lanes (round 5).  The scheme above walks a batch of 64 rays until its longest ray is done.  On the
// benchmark scene a ray fetches 5.8 nodes and the longest one 16: 77 % of the lane-steps of a batch do work.  On the large
// scene it is 22 fetches against 137 - 56 %, and the kernel is bound by the issue of its box tests (2.7e9 wave instructions
// per launch, 94 % of its node reads hit the L1: profiles/r07b/large_scene_trace_counters.txt), i.e. by steps that most
// lanes sit out.  In the second mode of the kernel a lane whose ray is done takes the next ray of the pending batch as soon
// as `refill_lanes` lanes of the wave are idle (or the whole wave is), wherever the other lanes are in their walks.  The
// pending batch stays where the loads put it (lane i holds ray i) and is handed out in lane order, so the source of the
// r-th idle lane is lane pending_next + r: five ds_bpermute_b32, no search; the origins of the batch - a dependent read
// behind the records - are fetched once, when its first ray is handed out, and wait in LDS (one float4 per lane: with the
// 4 KB of the stack 5 KB per wave, four of the 1280-byte granules in which gfx950 hands out LDS - 32 waves still fit a
// CU).  The batch after it is requested when the last ray of the pending one has been handed out and is not looked at
// before the next hand-out: its cold read still hides behind the walk.
// Measured (profiles/r07c, r07d; frame period, three frames in flight): large scene 5.31 -> 3.95 ms with 16 idle lanes (4, 8,
// 24, 32, 48: 4.12, 4.01, 3.96, 4.04, 4.80); but the benchmark scene LOSES - config 2 0.135 -> 0.140 ms, config 3 1.147 ->
// 1.157, config 4 18.95 -> 19.13, about a tenth of the kernel each time, also with hand-outs at least 4 or 8 steps apart:
// short rays all reach their leaves at about the same step when they start together, and staggered they wait for the
// sixteen that make a batch of triangle tests.  So a wave DECIDES: it walks its first batches the old way and counts how
// busy its lanes were (lane-steps / 64 steps); a wave whose batches were less than `refill_below` / 256 busy (default 0.65,
// between the two scenes' 0.77 and 0.56) hands out rays from then on.  VKR_WIDE_REFILL=0: never; VKR_WIDE_REFILL_BELOW=256:
// from the first batch on.
template <uint32_t THREADS>
__global__ void __launch_bounds__(THREADS, kWideTraceWaves) trace_shadow_rays_wide(bvh_view bvh, const uint4* __restrict__ wide_nodes, uint32_t wide_node_count, ray_stream rays, uint32_t* work_cursors, uint8_t* codes, uint32_t* spill, uint32_t leaf_batch, uint32_t lds_entries, uint32_t refill_lanes, uint32_t refill_below) {
	__shared__ uint32_t stack[kWideStackLds * THREADS];
#if VKR_LDS_TOP_NODES
	__shared__ uint4 top_nodes[VKR_LDS_TOP_NODES * 4];
	const uint32_t top_count = min((uint32_t) VKR_LDS_TOP_NODES, wide_node_count);
	for (uint32_t i = threadIdx.x; i < top_count * 4u; i += THREADS) top_nodes[i] = wide_nodes[i];
	__syncthreads();
#else
	(void) wide_node_count;
#endif
	const uint32_t lane = threadIdx.x & 63u;
	chunk_cursor cursor = make_chunk_cursor(rays, THREADS / 64u);
	// `item`: what the lane looks at next - a wide node (index), a triangle (kLeafBit | slot) or
	// nothing (kIdle: the lane has no ray)
	constexpr uint32_t kIdle = 0xFFFFFFFFu;
	f3 o = mk3(0.0f, 0.0f, 0.0f), d = o;
	wide_ray ray = {o, o, 0u, 0u, 0u};
	float t_max = 0.0f;
	// (byte index of the term's code: below 2^32, the host checks)
	uint32_t item = kIdle, code_index = 0;
#if VKR_TRACE_BLOCKER_CACHE
	// the triangle that blocked this lane's most recent blocked ray, or kIdle (see try_last_blocker below)
	uint32_t last_blocker = kIdle;
#endif
	// The stack pointer is the lane's LDS address itself (entries are THREADS x 4 bytes apart), so
	// that a push is a store and a conditional add; entries beyond the LDS part only exist as a
	// depth (`top` then points behind the LDS part and is never dereferenced)
	// (as 32-bit LDS byte addresses: generic pointers make the compiler do the arithmetic in 64 bits)
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	constexpr uint32_t kEntry = THREADS * 4u;
	const uint32_t my_stack = (uint32_t) (uintptr_t) (lds_u32*) (stack + threadIdx.x);
	// (lds_entries <= kWideStackLds: tests shrink the LDS part to drive rays through the spill path)
	const uint32_t lds_end = my_stack + lds_entries * kEntry;
	uint32_t top = my_stack;
#define VKR_STACK_AT(address) (*(lds_u32*) (uintptr_t) (address))
	uint32_t* my_spill = spill + (size_t) blockIdx.x * THREADS + threadIdx.x;
	const size_t spill_stride = (size_t) gridDim.x * THREADS;
	// the batch after the current one, already on its way: this lane's direction and record word
	float4 next_direction = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
	uint32_t next_record = kNullRay;
	// Takes the next (up to) 64 rays of the chunk - claiming a new chunk if this one is used up - and
	// issues this lane's loads.  false: no work left for this wave.
	auto fetch_batch = [&]() -> bool {
		if (cursor.chunk_next >= cursor.chunk_count && (!cursor.chunks_left || !claim_chunk(cursor, rays, work_cursors))) return false;
		uint32_t index = cursor.chunk_next + lane;
		next_record = kNullRay;
		if (index < cursor.chunk_count) {
			next_direction = rays.directions[cursor.chunk_first + index];
			next_record = rays.records[cursor.chunk_first + index];
		}
		cursor.chunk_next += 64u;
		return true;
	};
	// One step of every walk of the wave: lanes at a node fetch it, test its four boxes and push the children that were
	// hit; lanes at a triangle test it once enough of them have gathered; then every lane that is through with its
	// item takes the next one off its stack - or is done, its ray unblocked.
	auto walk_step = [&](bool at_node, bool at_leaf, uint64_t node_lanes, uint64_t leaf_lanes) {
		bool pop = false;
		if (at_node) {
			const uint4* n = (const uint4*) ((const uint8_t*) wide_nodes + ((size_t) item << 6));
#if VKR_LDS_TOP_NODES
			if (item < top_count) n = top_nodes + 4u * item;
#endif
			uint4 qx = n[0], qy = n[1], qz = n[2], link = n[3];
			bool h0 = wide_ray_box(qx.x, qy.x, qz.x, ray, 1.0e-3f, t_max);
			bool h1 = wide_ray_box(qx.y, qy.y, qz.y, ray, 1.0e-3f, t_max);
			bool h2 = wide_ray_box(qx.z, qy.z, qz.z, ray, 1.0e-3f, t_max);
			bool h3 = wide_ray_box(qx.w, qy.w, qz.w, ray, 1.0e-3f, t_max);
			if (top + 4u * kEntry <= lds_end) {
				// (the last child first: the first one comes off the stack first, the order of
				// the scheme with a register for the next item)
				VKR_STACK_AT(top) = link.w; top += h3 ? kEntry : 0u;
				VKR_STACK_AT(top) = link.z; top += h2 ? kEntry : 0u;
				VKR_STACK_AT(top) = link.y; top += h1 ? kEntry : 0u;
				VKR_STACK_AT(top) = link.x; top += h0 ? kEntry : 0u;
			}
			else {
				const bool hits[4] = {h3, h2, h1, h0};
				const uint32_t links[4] = {link.w, link.z, link.y, link.x};
#pragma unroll
				for (int c = 0; c != 4; ++c) {
					// (an absent child cannot be hit - unless a NaN slab let it through: never push its link, which is kIdle)
					if (!hits[c] || links[c] == kWideEmpty) continue;
					if (top < lds_end) VKR_STACK_AT(top) = links[c];
					else my_spill[(size_t) ((top - lds_end) / kEntry) * spill_stride] = links[c];
					top += kEntry;
				}
			}
			pop = true;
		}
		bool test_leaves = leaf_lanes != 0 && (node_lanes == 0 || (uint32_t) __popcll((unsigned long long) leaf_lanes) >= leaf_batch);
		if (test_leaves && at_leaf) {
			const float4* t = bvh.triangles + 3 * (size_t) (item & ~kLeafBit);
			float dist;
			bool blocked = ray_triangle<false>(t[0], t[1], t[2], o, d, 1.0e-3f, t_max, dist);
			// a blocked ray is done: its term keeps the code the shading kernel gave it
#if VKR_TRACE_BLOCKER_CACHE
			if (blocked) last_blocker = item & ~kLeafBit;
#endif
			if (blocked) { item = kIdle; top = my_stack; }
			else pop = true;
		}
		if (pop) {
			if (top == my_stack) {
				codes[code_index] = (uint8_t) kCodeVisible;
				item = kIdle;
			}
			else {
				top -= kEntry;
				item = top < lds_end ? VKR_STACK_AT(top) : my_spill[(size_t) ((top - lds_end) / kEntry) * spill_stride];
			}
		}
	};
	// Blocker cache (round 6).  A lane's consecutive rays mostly come from one pixel (or its neighbour) and go toward one
	// light - the 64 rays of a batch are what the lanes of a shading wave queued for one sample of one light, and the next batch
	// is the next sample - so the triangle that blocked the lane's last blocked ray is the most likely blocker of its next ray.
	// A ray that has just been taken is tested against it before it walks: a hit ends the ray at once (any hit is a hit: the
	// result of the query is the same boolean), a miss costs one triangle test next to the 6 - 22 node fetches of a walk.
	auto try_last_blocker = [&]() {
#if VKR_TRACE_BLOCKER_CACHE
		if (item == 0u && last_blocker != kIdle) {
			const float4* t = bvh.triangles + 3 * (size_t) last_blocker;
			float dist;
			if (ray_triangle<false>(t[0], t[1], t[2], o, d, 1.0e-3f, t_max, dist)) item = kIdle;
		}
#endif
	};
	bool batch_pending = fetch_batch();
	// ---- a batch at a time, until the wave finds its lanes idle too often ------------------------------
	bool handing = refill_lanes != 0u && refill_below >= 256u;
	uint32_t steps = 0, lane_steps = 0;
	while (batch_pending && !handing) {
		// the prefetched batch becomes the current one
		{
			float4 a = next_direction;
			uint32_t record = next_record;
			uint32_t tid = ray_record_thread(rays.thread_bits, record);
			float4 b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			// (a slot that the shading wave reserved and did not need holds kNullRay)
			if (record != kNullRay) b = rays.origins[tid];
			// ... and the one after it is requested before the walk starts
			batch_pending = fetch_batch();
			if (record != kNullRay) {
				o = mk3(b.x, b.y, b.z); d = mk3(a.x, a.y, a.z); t_max = a.w;
				code_index = (uint32_t) code_slot(rays.thread_count, ray_record_cursor(rays.thread_bits, record), tid);
				ray = make_wide_ray(make_grid_ray(bvh, o, d));
				item = 0;
				top = my_stack;
				if (!(t_max >= 1.0e-3f)) {
					// empty interval: nothing can block the ray (same rule as any_hit)
					codes[code_index] = (uint8_t) kCodeVisible;
					item = kIdle;
				}
			}
			try_last_blocker();
		}
		// walk until every lane has run dry
		while (true) {
			bool at_node = item != kIdle && !(item & kLeafBit);
			bool at_leaf = item != kIdle && (item & kLeafBit) != 0;
			uint64_t node_lanes = __ballot(at_node), leaf_lanes = __ballot(at_leaf);
			if ((node_lanes | leaf_lanes) == 0) break;
			++steps;
			lane_steps += (uint32_t) __popcll((unsigned long long) (node_lanes | leaf_lanes));
			walk_step(at_node, at_leaf, node_lanes, leaf_lanes);
		}
		// (over all batches of the wave so far; a batch of null rays takes no step and decides nothing)
		handing = refill_lanes != 0u && steps >= 8u && 4u * lane_steps < refill_below * steps;
	}
	if (!batch_pending) return;
	// ---- rays handed to idle lanes ------------------------------------------------------------------------
	__shared__ float4 origin_stage[THREADS];
	float4* const my_wave_origins = origin_stage + (threadIdx.x & ~63u);
	bool origins_staged = false;
	// the rays of the pending batch that have not been handed out are those of lanes [pending_next, pending_end)
	uint32_t pending_next = 0, pending_end = min(64u, cursor.chunk_count + 64u - cursor.chunk_next);
	while (true) {
		uint64_t idle_lanes = __ballot(item == kIdle);
		uint32_t idle_count = (uint32_t) __popcll((unsigned long long) idle_lanes);
		if (batch_pending && (idle_count >= refill_lanes || idle_count == 64u)) {
			if (!origins_staged) {
				float4 mine = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				if (next_record != kNullRay) mine = rays.origins[ray_record_thread(rays.thread_bits, next_record)];
				my_wave_origins[lane] = mine;
				origins_staged = true;
				__builtin_amdgcn_wave_barrier();
			}
			uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle_lanes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle_lanes, 0u));
			uint32_t source = pending_next + rank;
			// (every lane takes part in the exchange: a lane that is switched off would deliver nothing)
			float4 a = make_float4(__shfl(next_direction.x, (int) (source & 63u)), __shfl(next_direction.y, (int) (source & 63u)),
				__shfl(next_direction.z, (int) (source & 63u)), __shfl(next_direction.w, (int) (source & 63u)));
			uint32_t record = (uint32_t) __shfl((int) next_record, (int) (source & 63u));
			bool take = item == kIdle && source < pending_end;
			pending_next += idle_count;
			// (a slot that the shading wave reserved and did not need holds kNullRay)
			if (take && record != kNullRay) {
				uint32_t tid = ray_record_thread(rays.thread_bits, record);
				float4 b = my_wave_origins[source & 63u];
				o = mk3(b.x, b.y, b.z); d = mk3(a.x, a.y, a.z); t_max = a.w;
				code_index = (uint32_t) code_slot(rays.thread_count, ray_record_cursor(rays.thread_bits, record), tid);
				ray = make_wide_ray(make_grid_ray(bvh, o, d));
				item = 0;
				top = my_stack;
				if (!(t_max >= 1.0e-3f)) {
					// empty interval: nothing can block the ray (same rule as any_hit)
					codes[code_index] = (uint8_t) kCodeVisible;
					item = kIdle;
				}
				try_last_blocker();
			}
			if (pending_next >= pending_end) {
				// the batch after it: requested now, looked at when lanes have run dry again
				// (every lane has read its origin before the stage is written again: the wave runs in lock step, and
				// the barrier keeps the compiler from moving the accesses across it)
				__builtin_amdgcn_wave_barrier();
				batch_pending = fetch_batch();
				origins_staged = false;
				pending_next = 0;
				pending_end = batch_pending ? min(64u, cursor.chunk_count + 64u - cursor.chunk_next) : 0u;
			}
		}
		bool at_node = item != kIdle && !(item & kLeafBit);
		bool at_leaf = item != kIdle && (item & kLeafBit) != 0;
		uint64_t node_lanes = __ballot(at_node), leaf_lanes = __ballot(at_leaf);
		if ((node_lanes | leaf_lanes) == 0) {
			if (!batch_pending) break;
			continue;
		}
		walk_step(at_node, at_leaf, node_lanes, leaf_lanes);
	}
}

#undef VKR_STACK_AT

// Replays every pixel's sums in the order of the shading program: terms of one light
// are added one after the other, the light's sum is scaled by 1 / SAMPLE_COUNT and
// added to the colour (shading_pass.frag.glsl:710, :858), then NaN check and exposure.
VKR_DEV void resolve_shadow_terms_body(const shade_params& p) {
	uint32_t px, py;
	size_t out_index;
	if (!locate_pixel(p, p.first_block + blockIdx.x, threadIdx.x, px, py, out_index)) return;
	uint32_t tid = blockIdx.x * 256u + threadIdx.x;
	// (the colour before the sampled terms is the light display's, +0 without it)
	f3 color = mk3(0.0f, 0.0f, 0.0f);
	if (p.show_polygonal_lights) {
		float4 base = p.base_color[tid];
		color = mk3(base.x, base.y, base.z);
	}
	f3 sum = mk3(0.0f, 0.0f, 0.0f);
	float rcp_samples = 1.0f / (float) p.sample_count;
	uint32_t term = 0;
	// Four codes per load (code_slot); the terms of a group are requested together - which ones is
	// known from the codes alone - and then added in program order.  (Adding the +0 of a term that
	// contributes nothing leaves the sum as it is: the sum starts at +0 and never becomes -0.)
	const uint32_t* code_words = (const uint32_t*) p.codes;
	const uint32_t groups = (p.max_codes + 3u) >> 2;
	bool ended = false;
	for (uint32_t g = 0; g < groups && !ended; ++g) {
		uint32_t word = code_words[(size_t) g * p.thread_count + tid];
		f3 value[4];
		uint32_t code[4];
		bool live = true;
#pragma unroll
		for (int j = 0; j != 4; ++j) {
			code[j] = (word >> (8 * j)) & 0xFFu;
			live = live && code[j] != kCodeEnd;
			bool is_term = live && code[j] != kCodeEndOfLight;
			size_t index = ((size_t) term * p.thread_count + tid) * 3;
			term += is_term ? 1u : 0u;
			value[j] = mk3(0.0f, 0.0f, 0.0f);
			if (is_term && (code[j] == kCodeVisible || code[j] == kCodeFinal))
				value[j] = mk3(p.terms_visible[index], p.terms_visible[index + 1], p.terms_visible[index + 2]);
			else if (is_term && code[j] == kCodePendingWithHidden)
				value[j] = mk3(p.terms_hidden[index], p.terms_hidden[index + 1], p.terms_hidden[index + 2]);
			else if (is_term && code[j] == kCodePendingHiddenNaN)
				value[j] = mk3(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
		}
#pragma unroll
		for (int j = 0; j != 4; ++j) {
			if (ended) continue;
			if (code[j] == kCodeEnd) ended = true;
			else if (code[j] == kCodeEndOfLight) {
				color = color + sum * rcp_samples;
				sum = mk3(0.0f, 0.0f, 0.0f);
			}
			else sum = sum + value[j];
		}
	}
	store_final_color(p, out_index, color);
}

// Leaves the ray queues empty for the next frame (saves two fill launches per frame) and
// keeps a copy of the counters for get_last_ray_count() / get_traversal_statistics().
// The rays of the launch are added to the frame's counter (get_last_ray_count()): what the shading
// waves counted (block-wise reservation, where the queue sizes include the null rays of partly used
// blocks), else the queue sizes.
__global__ void __launch_bounds__(256) resolve_shadow_terms_and_reset(const shade_params p) {
	resolve_shadow_terms_body(p);
	if (blockIdx.x == 0) {
		__shared__ uint32_t counted, queued;
		if (threadIdx.x == 0) { counted = 0; queued = 0; }
		__syncthreads();
		uint32_t* counters = const_cast<uint32_t*>(p.ray_queue_size);
		uint32_t my_queued = 0;
		for (uint32_t i = threadIdx.x; i < kRayCounterCount; i += 256u) {
			uint32_t value = counters[i];
			if (i < kRayQueueCount) my_queued += value;
			else if (i >= kRayCountOffset && (i - kRayCountOffset) % kCursorStride == 0 && value) atomicAdd(&counted, value);
			counters[kRayCounterCount + i] = value;
			counters[i] = 0;
		}
		if (my_queued) atomicAdd(&queued, my_queued);
		__syncthreads();
		if (threadIdx.x == 0 && p.ray_counter) {
			if (p.first_launch_of_frame) *p.ray_counter = (unsigned long long) (counted ? counted : queued);
			else atomicAdd(p.ray_counter, (unsigned long long) (counted ? counted : queued));
		}
	}
}

}  // namespace vkr
