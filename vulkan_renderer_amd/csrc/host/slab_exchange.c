/* Multi-GPU exchange of the shading pass (include/vkr_slab_exchange.h): shade this rank's
 * tiles -> ncclAllGather of the slabs over xGMI -> scatter into the frame, one frame after the
 * other without the host ever waiting.  Plain C on top of the HIP runtime's C API; RCCL is bound
 * with dlopen so that libvkr_shading.so itself does not depend on it.
 *
 * Streams of one frame k (buffer set b = k mod set_count):
 *   frame stream (the pass's own, render_shading_pass)   shade, trace, resolve [, encode] -> rendered[b]
 *   exchange stream (owned by the exchange)              wait rendered[b], all-gather [, scatter] -> assembled[b]
 * and the frame that reuses set b waits for assembled[b] before it starts, so the collective of
 * frame k runs while frame k + 1 is shaded.  Contract: BASELINE.json configs[3], SURVEY.md 8(e).
 *
 * Round 6: the all-gather is IN PLACE - a rank shades (or encodes) straight into its own slot of gathered[b],
 * send == gathered + rank * send_bytes, the form ncclAllGather documents as in place: no copy of the own slab, and
 * with one rank no data moves at all.  And the scatter is optional (exchange->assemble_on_demand): the gathered
 * slabs, tile-major, ARE the frame every rank holds; a reader that wants rows un-tiles them when it reads
 * (assemble_exchanged_frame, finish_slab_exchange) - SURVEY.md 8(e): "or un-tile on the consumer only". */
#include "vkr_internal.h"
#include "vkr_slab_exchange.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <pthread.h>

typedef struct rccl_binding_s {
	void* library;
	__typeof__(&ncclGetUniqueId) get_unique_id;
	__typeof__(&ncclCommInitRank) comm_init_rank;
	__typeof__(&ncclCommDestroy) comm_destroy;
	__typeof__(&ncclAllGather) all_gather;
	__typeof__(&ncclGetErrorString) get_error_string;
	ncclComm_t communicator;
} rccl_binding_t;

static int hip_failed(hipError_t error, const char* what) {
	if (error == hipSuccess) return 0;
	printf("HIP error while %s: %s\n", what, hipGetErrorString(error));
	return 1;
}

/* RCCL is loaded once per process and stays loaded: it registers exit handlers and keeps helper threads
   (the bootstrap thread of a rendezvous token, proxy threads of a communicator) whose code must not be
   unmapped under them - RTLD_NODELETE, and no dlclose() anywhere in this file. */
static void* g_rccl_library = NULL;
static pthread_mutex_t g_rccl_library_mutex = PTHREAD_MUTEX_INITIALIZER;

/* Binds the handful of RCCL entry points the exchange uses.  A process that already holds an
   RCCL (PyTorch brings its own copy) gets that one: the loader matches the soname. */
static int bind_rccl(rccl_binding_t* binding) {
	memset(binding, 0, sizeof(*binding));
	const char* requested = getenv("VKR_RCCL_LIBRARY");
	const char* candidates[] = {requested, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	pthread_mutex_lock(&g_rccl_library_mutex);
	for (uint32_t i = 0; i != VKR_COUNT_OF(candidates) && !g_rccl_library; ++i)
		if (candidates[i] && candidates[i][0]) g_rccl_library = dlopen(candidates[i], RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
	binding->library = g_rccl_library;
	pthread_mutex_unlock(&g_rccl_library_mutex);
	if (!binding->library) {
		const char* reason = dlerror();
		printf("The multi-GPU exchange needs RCCL, but librccl.so.1 could not be loaded (%s). Set VKR_RCCL_LIBRARY to its path.\n", reason ? reason : "no loader message");
		return 1;
	}
	*(void**) &binding->get_unique_id = dlsym(binding->library, "ncclGetUniqueId");
	*(void**) &binding->comm_init_rank = dlsym(binding->library, "ncclCommInitRank");
	*(void**) &binding->comm_destroy = dlsym(binding->library, "ncclCommDestroy");
	*(void**) &binding->all_gather = dlsym(binding->library, "ncclAllGather");
	*(void**) &binding->get_error_string = dlsym(binding->library, "ncclGetErrorString");
	if (!binding->get_unique_id || !binding->comm_init_rank || !binding->comm_destroy || !binding->all_gather || !binding->get_error_string) {
		printf("The RCCL library lacks one of ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString.\n");
		memset(binding, 0, sizeof(*binding));
		return 1;
	}
	return 0;
}

static int rccl_failed(const rccl_binding_t* binding, ncclResult_t result, const char* what) {
	if (result == ncclSuccess) return 0;
	printf("RCCL error while %s: %s\n", what, binding->get_error_string(result));
	return 1;
}

int get_slab_exchange_id(slab_exchange_id_t* id) {
	memset(id, 0, sizeof(*id));
	rccl_binding_t binding;
	if (bind_rccl(&binding)) return 1;
	ncclUniqueId unique;
	_Static_assert(sizeof(ncclUniqueId) <= sizeof(slab_exchange_id_t), "slab_exchange_id_t must hold an ncclUniqueId");
	int failed = rccl_failed(&binding, binding.get_unique_id(&unique), "creating the rendezvous token");
	if (!failed) memcpy(id->bytes, &unique, sizeof(unique));
	return failed;
}

/* the default collective */
static int gather_with_rccl(void* context, uint32_t rank, uint32_t set, const void* send, void* gathered, uint64_t send_bytes, void* stream) {
	rccl_binding_t* binding = (rccl_binding_t*) context;
	(void) rank; (void) set;
	/* bytes, so that both formats share one call; slabs are multiples of 256 pixels, i.e. of 16 bytes */
	return rccl_failed(binding, binding->all_gather(send, gathered, (size_t) send_bytes, ncclUint8, binding->communicator, (hipStream_t) stream), "gathering the slabs");
}

/* ---- ranks of one process: copies instead of a collective ------------------------------------ */

struct local_slab_group_s {
	uint32_t rank_count, joined;
	/* buffer sets per rank (the first rank to join decides; the others must agree: sets are addressed by index across ranks) */
	uint32_t set_count;
	/* a rank has left the frame loop with an error: every rendezvous returns at once, now and from now on */
	int aborted;
	pthread_mutex_t mutex;
	pthread_cond_t changed;
	/* rendezvous of the ranks' threads: `generation` advances when the last rank arrives */
	uint32_t waiting;
	uint64_t generation;
	/* per rank: its exchange (buffer sets, `assembled` events) and, per set, the event behind the
	   copies that rank queued */
	slab_exchange_t* exchanges[64];
	hipEvent_t copied[64][VKR_MAX_FRAMES_IN_FLIGHT];
	uint64_t frames[64];
};

local_slab_group_t* create_local_slab_group(uint32_t rank_count) {
	if (rank_count == 0 || rank_count > 64) {
		printf("A local slab group has 1 to 64 ranks.\n");
		return NULL;
	}
	local_slab_group_t* group = (local_slab_group_t*) calloc(1, sizeof(local_slab_group_t));
	if (!group) return NULL;
	group->rank_count = rank_count;
	pthread_mutex_init(&group->mutex, NULL);
	pthread_cond_init(&group->changed, NULL);
	return group;
}

void destroy_local_slab_group(local_slab_group_t* group) {
	if (!group) return;
	for (uint32_t r = 0; r != 64; ++r)
		for (uint32_t b = 0; b != VKR_MAX_FRAMES_IN_FLIGHT; ++b)
			if (group->copied[r][b]) (void) hipEventDestroy(group->copied[r][b]);
	pthread_cond_destroy(&group->changed);
	pthread_mutex_destroy(&group->mutex);
	free(group);
}

/* all ranks' threads meet here; 1 if a rank has aborted the group (then nobody waits) */
static int local_group_rendezvous(local_slab_group_t* group) {
	pthread_mutex_lock(&group->mutex);
	uint64_t generation = group->generation;
	if (!group->aborted) {
		if (++group->waiting == group->rank_count) {
			group->waiting = 0;
			++group->generation;
			pthread_cond_broadcast(&group->changed);
		}
		else
			while (group->generation == generation && !group->aborted) pthread_cond_wait(&group->changed, &group->mutex);
	}
	int aborted = group->aborted;
	pthread_mutex_unlock(&group->mutex);
	return aborted;
}

/* A rank that cannot reach the rendezvous of a frame (its frame failed before the gather) says so, or its
   peers would wait for it forever */
static void local_group_abort(local_slab_group_t* group) {
	pthread_mutex_lock(&group->mutex);
	group->aborted = 1;
	pthread_cond_broadcast(&group->changed);
	pthread_mutex_unlock(&group->mutex);
}

/* Rank r copies its slab into slot r of every rank's `gathered` buffer of this set - behind that
   rank's scatter of the frame that used the set before -, marks the copies with an event, meets the
   other ranks' threads (so that their events exist) and makes its stream wait for their copies. */
static int gather_with_copies(void* context, uint32_t rank, uint32_t set, const void* send, void* gathered, uint64_t send_bytes, void* stream) {
	local_slab_group_t* group = (local_slab_group_t*) context;
	hipStream_t s = (hipStream_t) stream;
	(void) gathered;
	int failed = 0;
	int reused = group->frames[rank] >= group->set_count;
	for (uint32_t q = 0; q != group->rank_count && !failed; ++q) {
		const slab_exchange_t* peer = group->exchanges[q];
		if (!peer || !peer->gathered[set]) {
			printf("Rank %u of the local slab group has no exchange (its create_local_slab_exchange failed).\n", q);
			failed = 1;
			break;
		}
		if (reused && q != rank) failed = hip_failed(hipStreamWaitEvent(s, (hipEvent_t) peer->assembled[set], 0), "waiting for a peer's scatter");
		/* (the rank's own slot is where the frame was shaded into: in place, nothing to copy) */
		void* slot = (uint8_t*) peer->gathered[set] + (size_t) rank * send_bytes;
		if (!failed && slot != send) failed = hip_failed(hipMemcpyAsync(slot, send, (size_t) send_bytes, hipMemcpyDeviceToDevice, s), "copying a slab to a peer");
	}
	if (!failed) failed = hip_failed(hipEventRecord(group->copied[rank][set], s), "marking the copies");
	++group->frames[rank];
	/* (a rank that failed says so - its peers must not queue waits for copies that were never made) */
	if (failed) local_group_abort(group);
	if (local_group_rendezvous(group)) {
		if (!failed) printf("Another rank of the local slab group has failed; rank %u leaves the frame.\n", rank);
		return 1;
	}
	for (uint32_t q = 0; q != group->rank_count && !failed; ++q)
		if (q != rank) failed = hip_failed(hipStreamWaitEvent(s, group->copied[q][set], 0), "waiting for a peer's copies");
	return failed;
}


void destroy_slab_exchange(slab_exchange_t* exchange, application_t* app) {
	/* frames in flight and collectives still use the buffers that are freed below */
	if (exchange->stream) (void) hipStreamSynchronize((hipStream_t) exchange->stream);
	/* (also frames without the wavefront pipeline - inline rays on device->stream - write the slabs) */
	if (app) (void) wait_for_device(&app->device);
	rccl_binding_t* binding = (rccl_binding_t*) exchange->binding;
	if (binding) {
		if (binding->communicator) (void) binding->comm_destroy(binding->communicator);
		free(binding);
	}
	for (uint32_t b = 0; b != VKR_MAX_FRAMES_IN_FLIGHT; ++b) {
		/* (rgba32f: slab_radiance[b] == send[b] is the rank's slot of gathered[b]; rgb8: send[b] is, and the float slab is its own) */
		if (exchange->slab_radiance[b] && exchange->slab_radiance[b] != exchange->send[b]) (void) hipFree(exchange->slab_radiance[b]);
		if (exchange->gathered[b]) (void) hipFree(exchange->gathered[b]);
		if (exchange->rendered[b]) (void) hipEventDestroy((hipEvent_t) exchange->rendered[b]);
		/* (the pass may still hold the event as the reader of this set's slab: vkr_note_target_reader) */
		if (exchange->assembled[b] && app) vkr_forget_target_reader(app, exchange->assembled[b]);
		if (exchange->assembled[b]) (void) hipEventDestroy((hipEvent_t) exchange->assembled[b]);
		for (uint32_t i = 0; i != 5; ++i)
			if (exchange->timing[b][i]) (void) hipEventDestroy((hipEvent_t) exchange->timing[b][i]);
	}
	if (exchange->stream) (void) hipStreamDestroy((hipStream_t) exchange->stream);
	memset(exchange, 0, sizeof(*exchange));
}

/* Everything of an exchange but its collective: ranks, format, buffer sets, stream, events */
/* one_device: every rank of the exchange runs on this device (one rank, or the local transport).  Then the events that hand
   a slab to the collective and the gathered slabs back to the pass release to DEVICE scope like every event of the pass (an
   event without the flag releases to system scope: the L2s are written back and invalidated under the frames that are
   running, every time it is recorded).  With peers on other devices they keep the default, system scope: a peer's kernel may
   read this rank's slab over xGMI, and what the peers wrote into the gathered buffer has to be seen by the kernels that read
   it here, past lines of the buffer's previous use that this device's L2 still holds. */
static int create_exchange_buffers(slab_exchange_t* exchange, application_t* app, slab_format_t format, int one_device) {
	memset(exchange, 0, sizeof(*exchange));
	const tile_schedule_t* schedule = &app->tile_schedule;
	uint32_t rank_count = schedule->rank_count > 1 ? schedule->rank_count : 1;
	if ((int) format < 0 || format >= slab_format_count || schedule->rank >= rank_count || (rank_count == 1 && !schedule->slab_layout)) {
		printf("A slab exchange needs a slab format, a rank below the rank count and, with a single rank, tile_schedule.slab_layout.\n");
		return 1;
	}
	exchange->rank = schedule->rank;
	exchange->rank_count = rank_count;
	exchange->format = format;
	exchange->slab_pixel_count = get_slab_pixel_count(app, 0);
	exchange->send_bytes = exchange->slab_pixel_count * (format == slab_format_rgba32f ? 16u : 3u);
	exchange->timing_stride = app->shading_pass.timing_stride;
	uint32_t sets = app->shading_pass.frames_in_flight;
	if (sets < 2) sets = 2;
	if (sets > VKR_MAX_FRAMES_IN_FLIGHT) sets = VKR_MAX_FRAMES_IN_FLIGHT;
	exchange->set_count = sets;
	if (hip_failed(hipSetDevice(app->device.hip_device), "selecting the device")) return 1;
	int failed = hip_failed(hipStreamCreateWithFlags((hipStream_t*) &exchange->stream, hipStreamNonBlocking), "creating the exchange stream");
	const unsigned event_flags = (one_device || rank_count == 1) ? (hipEventDisableTiming | hipEventReleaseToDevice) : hipEventDisableTiming;
	for (uint32_t b = 0; b != sets && !failed; ++b) {
		failed = hip_failed(hipMalloc(&exchange->gathered[b], exchange->send_bytes * rank_count), "allocating the gathered slabs")
			|| hip_failed(hipEventCreateWithFlags((hipEvent_t*) &exchange->rendered[b], event_flags), "creating events")
			|| hip_failed(hipEventCreateWithFlags((hipEvent_t*) &exchange->assembled[b], event_flags), "creating events");
		/* in place: what this rank sends is its own slot of the gathered slabs */
		if (!failed) exchange->send[b] = (uint8_t*) exchange->gathered[b] + (size_t) exchange->rank * exchange->send_bytes;
		if (!failed && format == slab_format_rgba32f) exchange->slab_radiance[b] = exchange->send[b];
		else if (!failed) failed = hip_failed(hipMalloc(&exchange->slab_radiance[b], exchange->slab_pixel_count * 16u), "allocating a slab");
		for (uint32_t i = 0; i != 5 && !failed; ++i)
			failed = hip_failed(hipEventCreate((hipEvent_t*) &exchange->timing[b][i]), "creating timing events");
		/* padding slots of the last tile row / column are never written by the kernels (the other ranks' slots arrive
		   with their padding cleared the same way) */
		if (!failed) failed = hip_failed(hipMemsetAsync(exchange->gathered[b], 0, exchange->send_bytes * rank_count, (hipStream_t) exchange->stream), "clearing the gathered slabs");
		if (!failed && exchange->slab_radiance[b] != exchange->send[b]) failed = hip_failed(hipMemsetAsync(exchange->slab_radiance[b], 0, exchange->slab_pixel_count * 16u, (hipStream_t) exchange->stream), "clearing a slab");
	}
	/* the frames that write the slabs run on non-blocking streams, which nothing orders behind the clears */
	if (!failed) failed = hip_failed(hipStreamSynchronize((hipStream_t) exchange->stream), "clearing the slabs");
	if (failed) {
		destroy_slab_exchange(exchange, app);
		return 1;
	}
	return 0;
}

int create_slab_exchange(slab_exchange_t* exchange, application_t* app, const slab_exchange_id_t* id, slab_format_t format) {
	if (create_exchange_buffers(exchange, app, format, 0)) return 1;
	rccl_binding_t* binding = (rccl_binding_t*) calloc(1, sizeof(rccl_binding_t));
	exchange->binding = binding;
	if (!binding || bind_rccl(binding)) {
		destroy_slab_exchange(exchange, app);
		return 1;
	}
	ncclUniqueId unique;
	memcpy(&unique, id->bytes, sizeof(unique));
	if (rccl_failed(binding, binding->comm_init_rank(&binding->communicator, (int) exchange->rank_count, unique, (int) exchange->rank), "joining the communicator")) {
		binding->communicator = NULL;
		destroy_slab_exchange(exchange, app);
		return 1;
	}
	exchange->gather = gather_with_rccl;
	exchange->gather_context = binding;
	return 0;
}

static int create_exchange_with_gather(slab_exchange_t* exchange, application_t* app, slab_gather_function_t gather, void* gather_context, slab_format_t format, int one_device) {
	if (!gather) {
		printf("create_slab_exchange_with_gather() needs a gather function.\n");
		memset(exchange, 0, sizeof(*exchange));
		return 1;
	}
	if (create_exchange_buffers(exchange, app, format, one_device)) return 1;
	exchange->gather = gather;
	exchange->gather_context = gather_context;
	return 0;
}

int create_slab_exchange_with_gather(slab_exchange_t* exchange, application_t* app, slab_gather_function_t gather, void* gather_context, slab_format_t format) {
	/* (a transport of the caller may reach other devices) */
	return create_exchange_with_gather(exchange, app, gather, gather_context, format, 0);
}

int create_local_slab_exchange(slab_exchange_t* exchange, application_t* app, local_slab_group_t* group, slab_format_t format) {
	uint32_t rank_count = app->tile_schedule.rank_count > 1 ? app->tile_schedule.rank_count : 1;
	if (!group || group->rank_count != rank_count) {
		printf("create_local_slab_exchange() needs a group with as many ranks as the tile schedule has.\n");
		memset(exchange, 0, sizeof(*exchange));
		return 1;
	}
	int failed = create_exchange_with_gather(exchange, app, gather_with_copies, group, format, 1);
	uint32_t rank = app->tile_schedule.rank;
	if (!failed) {
		pthread_mutex_lock(&group->mutex);
		if (!group->set_count) group->set_count = exchange->set_count;
		int mismatch = group->set_count != exchange->set_count;
		pthread_mutex_unlock(&group->mutex);
		if (mismatch) {
			printf("Rank %u joins the local slab group with %u buffer sets (frames in flight) but the group has %u: all ranks need the same.\n", rank, exchange->set_count, group->set_count);
			failed = 1;
		}
	}
	for (uint32_t b = 0; b != VKR_MAX_FRAMES_IN_FLIGHT && !failed; ++b)
		if (!group->copied[rank][b]) failed = hip_failed(hipEventCreateWithFlags(&group->copied[rank][b], hipEventDisableTiming | hipEventReleaseToDevice), "creating events");
	if (!failed) {
		group->exchanges[rank] = exchange;
		group->frames[rank] = 0;
	}
	/* every rank's buffers and events must exist before the first frame copies into them
	   (a rank that failed still shows up; its peers find out at their first frame) */
	(void) local_group_rendezvous(group);
	if (failed && exchange->stream) destroy_slab_exchange(exchange, app);
	return failed;
}

int all_gather_slabs(slab_exchange_t* exchange, const void* send, void* gathered, void* stream) {
	if (!exchange->gather) {
		printf("all_gather_slabs() needs an exchange made by one of the create_*_slab_exchange() functions.\n");
		return 1;
	}
	/* transports that address buffers by set (the local copies write into the PEERS' gathered[set]) need one of
	   the exchange's own buffers; ncclAllGather takes any */
	uint32_t set = exchange->set_count;
	for (uint32_t b = 0; b != exchange->set_count; ++b) if (exchange->gathered[b] == gathered) set = b;
	if (set == exchange->set_count) {
		if (exchange->gather != gather_with_rccl) {
			printf("all_gather_slabs(): `gathered` must be one of exchange->gathered[] with this transport.\n");
			return 1;
		}
		set = 0;
	}
	return exchange->gather(exchange->gather_context, exchange->rank, set, send, gathered, exchange->send_bytes, stream);
}

int render_and_exchange_frame(application_t* app, slab_exchange_t* exchange, void* out_frame) {
	if (!exchange->gather || exchange->slab_pixel_count != get_slab_pixel_count(app, 0)
		|| exchange->rank != app->tile_schedule.rank || exchange->rank_count != (app->tile_schedule.rank_count > 1 ? app->tile_schedule.rank_count : 1))
	{
		printf("The slab exchange does not match the tile schedule or extent. Recreate it.\n");
		if (exchange->gather == gather_with_copies) local_group_abort((local_slab_group_t*) exchange->gather_context);
		return 1;
	}
	uint32_t b = exchange->next_set;
	exchange->next_set = (b + 1) % exchange->set_count;
	int timed = exchange->timing_stride <= 1 || exchange->frame_counter % exchange->timing_stride == 0;
	int reused = exchange->frame_counter >= exchange->set_count;
	++exchange->frame_counter;
	hipStream_t exchange_stream = (hipStream_t) exchange->stream;
	/* The set's previous frame must have left the buffers before this frame writes them.  Which stream
	   the frame takes - device->stream or one of the frame streams - is the pass's decision (it depends
	   on the ray mode, the scene, the pipeline depth, and differs for the first frame of a fresh pass), so
	   the pass itself waits for the event, on the stream it picks, before its first kernel.
	   Round 6: not the whole frame waits but the kernel that writes what the collective of the set's previous frame reads -
	   this rank's slot of gathered[b] (vkr_note_target_reader, shading_pass.hip). */
	if (reused) vkr_note_target_reader(app, exchange->assembled[b], exchange->send[b], (size_t) exchange->send_bytes);
	/* (an untimed frame leaves the set's timing events alone: they keep the last timed frame; the
	   start mark goes to the stream the frame is expected on, which only the very first frame may miss) */
	if (timed) (void) hipEventRecord((hipEvent_t) exchange->timing[b][0], (hipStream_t) get_next_frame_stream(app));
	int failed = exchange->format == slab_format_rgb8
		? render_shading_pass_encoded(app, exchange->slab_radiance[b], exchange->send[b])
		: render_shading_pass(app, exchange->slab_radiance[b]);
	app->shading_pass.wait_before_next_frame = NULL;
	if (failed) {
		if (exchange->gather == gather_with_copies) local_group_abort((local_slab_group_t*) exchange->gather_context);
		return 1;
	}
	/* ... and the stream it did pick carries the frame */
	hipStream_t used_stream = (hipStream_t) app->shading_pass.last_frame_stream;
	if (timed) (void) hipEventRecord((hipEvent_t) exchange->timing[b][1], used_stream);
	if (hip_failed(hipEventRecord((hipEvent_t) exchange->rendered[b], used_stream), "marking the frame")
		|| hip_failed(hipStreamWaitEvent(exchange_stream, (hipEvent_t) exchange->rendered[b], 0), "waiting for the frame"))
	{
		if (exchange->gather == gather_with_copies) local_group_abort((local_slab_group_t*) exchange->gather_context);
		return 1;
	}
	if (timed) (void) hipEventRecord((hipEvent_t) exchange->timing[b][2], exchange_stream);
	if (exchange->gather(exchange->gather_context, exchange->rank, b, exchange->send[b], exchange->gathered[b], exchange->send_bytes, exchange_stream)) return 1;
	if (timed) (void) hipEventRecord((hipEvent_t) exchange->timing[b][3], exchange_stream);
	void* target = out_frame ? out_frame : (exchange->format == slab_format_rgba32f ? app->render_targets.radiance : app->render_targets.encoded);
	/* on demand: the gathered slabs of this set are the frame until a reader asks for rows (assemble_exchanged_frame) */
	exchange->last_set = b;
	exchange->last_frame_assembled = 0;
	if (!exchange->assemble_on_demand || out_frame) {
		if (vkr_assemble_slabs_on_stream(app, exchange->gathered[b], target, (int) exchange->format, exchange_stream)) return 1;
		exchange->last_frame_assembled = 1;
	}
	if (timed) (void) hipEventRecord((hipEvent_t) exchange->timing[b][4], exchange_stream);
	if (timed) exchange->timed[b] = 1;
	exchange->last_frame = target;
	return hip_failed(hipEventRecord((hipEvent_t) exchange->assembled[b], exchange_stream), "marking the assembled frame");
}

int assemble_exchanged_frame(application_t* app, slab_exchange_t* exchange, void* out_frame) {
	if (!exchange->frame_counter) {
		printf("assemble_exchanged_frame() needs a frame: call render_and_exchange_frame() first.\n");
		return 1;
	}
	uint32_t b = exchange->last_set;
	hipStream_t exchange_stream = (hipStream_t) exchange->stream;
	void* target = out_frame ? out_frame : (exchange->format == slab_format_rgba32f ? app->render_targets.radiance : app->render_targets.encoded);
	/* (the exchange stream is behind the gather of that frame; the set is not reused before `assembled` is recorded again) */
	if (vkr_assemble_slabs_on_stream(app, exchange->gathered[b], target, (int) exchange->format, exchange_stream)) return 1;
	exchange->last_frame = target;
	if (!out_frame) exchange->last_frame_assembled = 1;
	return hip_failed(hipEventRecord((hipEvent_t) exchange->assembled[b], exchange_stream), "marking the assembled frame");
}

int finish_slab_exchange(application_t* app, slab_exchange_t* exchange) {
	if (!exchange->frame_counter) return 0;
	/* the exchange stream completes frames in order: the most recent set is the last one */
	uint32_t last = (exchange->next_set + exchange->set_count - 1) % exchange->set_count;
	/* on demand: this is the reader - the most recent frame is un-tiled into the render target now, once */
	if (exchange->assemble_on_demand && !exchange->last_frame_assembled && assemble_exchanged_frame(app, exchange, NULL)) return 1;
	return finish_frames(app)
		|| hip_failed(hipStreamWaitEvent((hipStream_t) app->device.stream, (hipEvent_t) exchange->assembled[last], 0), "waiting for the assembled frame");
}

uint32_t get_slab_exchange_milliseconds(slab_exchange_t* exchange, float out_milliseconds[3]) {
	out_milliseconds[0] = out_milliseconds[1] = out_milliseconds[2] = 0.0f;
	/* newest timed set first */
	for (uint32_t i = 0; i != exchange->set_count; ++i) {
		uint32_t b = (exchange->next_set + 2 * exchange->set_count - 1 - i) % exchange->set_count;
		if (!exchange->timed[b]) continue;
		hipEvent_t* e = (hipEvent_t*) exchange->timing[b];
		if (hipEventSynchronize(e[4]) != hipSuccess) return 0;
		if (hipEventElapsedTime(&out_milliseconds[0], e[0], e[1]) != hipSuccess
			|| hipEventElapsedTime(&out_milliseconds[1], e[2], e[3]) != hipSuccess
			|| hipEventElapsedTime(&out_milliseconds[2], e[3], e[4]) != hipSuccess)
			return 0;
		return 1;
	}
	return 0;
}
