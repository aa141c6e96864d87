/* Multi-GPU exchange of the shading pass: every rank shades the tiles that
 * app->tile_schedule gives it into a dense slab, one all-gather over RCCL (xGMI)
 * delivers all slabs to all ranks, a scatter kernel turns them back into the frame.
 *
 * The reference renders on one GPU and has no counterpart; the contract is
 * BASELINE.json configs[3] ("image tiled across 8 x MI355X with RCCL all-gather over
 * xGMI") and SURVEY.md 8(e).  What the exchange moves is the output of the pass
 * (reference: the colour attachment written at shading_pass.frag.glsl:866-892):
 * either vec4(final_color * exposure, 1) as RGBA32F or the sRGB-encoded 8-bit colour
 * as packed RGB8.
 *
 * Host code is plain C; RCCL is bound at run time (dlopen of librccl.so.1, or the
 * library named by VKR_RCCL_LIBRARY), so that single-GPU users of libvkr_shading.so
 * need no RCCL.  One process (or thread) per GPU:
 *
 *     slab_exchange_id_t id;                       // rank 0
 *     get_slab_exchange_id(&id);                   //   ... broadcast it (MPI_Bcast, a file, a socket)
 *     app.tile_schedule = (tile_schedule_t) {32, rank, rank_count, 0};
 *     create_slab_exchange(&exchange, &app, &id, slab_format_rgba32f);
 *     for (;;) render_and_exchange_frame(&app, &exchange, NULL);   // asynchronous, frames overlap
 *     finish_slab_exchange(&app, &exchange);       // app.device.stream now sees the last frame
 */
#ifndef VKR_SLAB_EXCHANGE_H
#define VKR_SLAB_EXCHANGE_H
#include "vkr_shading_pass.h"

typedef enum slab_format_e {
	/*! 16 bytes per pixel: the radiance target as it is (render_targets_t.radiance) */
	slab_format_rgba32f = 0,
	/*! 3 bytes per pixel: the encoded output without its constant alpha (encode_slab_rgb8);
		the frame is assembled as RGBA8 (render_targets_t.encoded) */
	slab_format_rgb8 = 1,
	slab_format_count
} slab_format_t;

/*! The rendezvous token of a communicator (an ncclUniqueId, 128 bytes): made by one
	rank, handed to all ranks by whatever launched them */
typedef struct slab_exchange_id_s {
	char bytes[128];
} slab_exchange_id_t;

/*! The collective behind the exchange: rank-major concatenation of every rank's `send_bytes` bytes
	at `send` into `gathered` on EVERY rank, queued on hipStream_t `stream`; `set` is the buffer set
	of the frame (slab_exchange_t.gathered[set] == gathered).  Called once per frame by every rank.
	The exchange calls it IN PLACE: send == gathered + rank * send_bytes (the rank has shaded its slab into
	its own slot), which is ncclAllGather's in-place form; a transport must not read `send` after it has
	begun to overwrite that slot with anything else.
	The default is ncclAllGather (create_slab_exchange); create_slab_exchange_with_gather() takes any
	other transport, create_local_slab_exchange() one made of device-to-device copies for ranks that
	live in one process.  Returns 0 on success. */
typedef int (*slab_gather_function_t)(void* context, uint32_t rank, uint32_t set, const void* send, void* gathered, uint64_t send_bytes, void* stream);

/*! Ranks of one process that exchange their slabs with plain copies (opaque; see create_local_slab_group) */
typedef struct local_slab_group_s local_slab_group_t;

typedef struct slab_exchange_s {
	/*! RCCL binding and communicator (internal; NULL with a caller-supplied gather) */
	void* binding;
	/*! the collective and its context */
	slab_gather_function_t gather;
	void* gather_context;
	uint32_t rank, rank_count;
	slab_format_t format;
	/*! pixels of one rank's slab (get_slab_pixel_count(app, 0): all slabs are padded to it)
		and bytes of it in the exchanged format */
	uint64_t slab_pixel_count;
	uint64_t send_bytes;
	/*! one set of buffers per frame that may be in flight (at least two): the RGBA32F slab
		the frame is shaded into, the slab in the exchanged format (the same buffer for
		rgba32f) and the gathered slabs of all ranks */
	uint32_t set_count, next_set;
	void* slab_radiance[VKR_MAX_FRAMES_IN_FLIGHT];
	void* send[VKR_MAX_FRAMES_IN_FLIGHT];
	void* gathered[VKR_MAX_FRAMES_IN_FLIGHT];
	/*! hipStream_t of the collectives and the scatter kernels (owned) */
	void* stream;
	/*! hipEvent_t per set: frame shaded (and encoded); frame assembled */
	void* rendered[VKR_MAX_FRAMES_IN_FLIGHT];
	void* assembled[VKR_MAX_FRAMES_IN_FLIGHT];
	/*! timing events per set (render start, rendered, gather start, gathered, assembled) and
		whether the set's most recent frame recorded them */
	void* timing[VKR_MAX_FRAMES_IN_FLIGHT][5];
	uint32_t timed[VKR_MAX_FRAMES_IN_FLIGHT];
	/*! frames submitted so far; every timing_stride-th is timed (0 or 1: all) */
	uint64_t frame_counter;
	uint32_t timing_stride;
	/*! where the most recent frame is assembled */
	void* last_frame;
	/*! Set by the caller after creation.  0 (default): every frame is scattered into its target by
		render_and_exchange_frame().  1: a frame stays what the all-gather delivers - all slabs, tile-major, in
		gathered[last_set] on every rank - and is un-tiled only when a reader asks: assemble_exchanged_frame(), or
		finish_slab_exchange() for the most recent frame ("un-tile on the consumer only").  A frame
		rendered with an explicit out_frame is always scattered. */
	uint32_t assemble_on_demand;
	/*! buffer set of the most recent frame; whether that frame has been scattered into the default target */
	uint32_t last_set, last_frame_assembled;
} slab_exchange_t;

/*! Creates the rendezvous token (ncclGetUniqueId).  Call on one rank. */
VKR_API int get_slab_exchange_id(slab_exchange_id_t* id);
/*! Joins the communicator (ncclCommInitRank with app->tile_schedule.rank / rank_count;
	collective: returns when all ranks have joined) and allocates the buffer sets for the
	current extent, tile schedule and app->shading_pass.frames_in_flight.  With one rank,
	app->tile_schedule.slab_layout must be set.  Several ranks in one process: call from one
	thread per GPU.  0 on success; prints the reason and cleans up otherwise. */
VKR_API int create_slab_exchange(slab_exchange_t* exchange, application_t* app, const slab_exchange_id_t* id, slab_format_t format);
/*! The same with the collective supplied by the caller (no RCCL is loaded): `gather` is called from
	render_and_exchange_frame() on every rank, once per frame */
VKR_API int create_slab_exchange_with_gather(slab_exchange_t* exchange, application_t* app, slab_gather_function_t gather, void* gather_context, slab_format_t format);
/*! Ranks that live in ONE process (one thread per rank, each with its own application_t, on one
	device or on devices with peer access): the "collective" is rank_count device-to-device copies
	per rank and frame, ordered by events, with a host-side rendezvous of the ranks' threads per
	frame where RCCL's kernels would meet on the device.  What it is for: running the whole N-rank
	schedule - tiles, slabs, buffer sets, overlap with the next frame, scatter - where RCCL cannot (RCCL
	refuses two ranks on one device), e.g. on a single-GPU machine.  Create the group once, then one
	exchange per rank from that rank's thread; every rank must submit the same number of frames. */
VKR_API local_slab_group_t* create_local_slab_group(uint32_t rank_count);
VKR_API void destroy_local_slab_group(local_slab_group_t* group);
VKR_API int create_local_slab_exchange(slab_exchange_t* exchange, application_t* app, local_slab_group_t* group, slab_format_t format);
VKR_API void destroy_slab_exchange(slab_exchange_t* exchange, application_t* app);
/*! One frame of the multi-GPU pass, queued asynchronously: render_shading_pass() of this rank's
	tiles into a slab (encoded on the same stream for rgb8), ncclAllGather of the slabs on the
	exchange stream - in place: the slab is shaded into this rank's slot of the gathered buffer -, scatter into
	`out_frame` (NULL: render_targets.radiance for rgba32f, render_targets.encoded for rgb8; not at all with
	exchange->assemble_on_demand and no out_frame).  The collective and the scatter of frame k overlap the
	shading of frame k + 1; frames complete in order. */
VKR_API int render_and_exchange_frame(application_t* app, slab_exchange_t* exchange, void* out_frame);
/*! Un-tiles the most recent frame - the gathered slabs of its buffer set - into `out_frame` (NULL: render_targets.radiance
	for rgba32f, render_targets.encoded for rgb8) on the exchange stream, behind that frame's all-gather.  For exchanges with
	assemble_on_demand; call finish_slab_exchange() before reading the target on app->device.stream. */
VKR_API int assemble_exchanged_frame(application_t* app, slab_exchange_t* exchange, void* out_frame);
/*! The collective alone (ncclAllGather unless the exchange was created with another one):
	`send_bytes` bytes per rank from `send` into `gathered` (rank-major) on hipStream_t `stream`, for
	callers that schedule the steps themselves */
VKR_API int all_gather_slabs(slab_exchange_t* exchange, const void* send, void* gathered, void* stream);
/*! Makes app->device.stream wait (on the device) for every frame submitted so far to be
	assembled (assemble_on_demand: un-tiles the most recent frame into the default target first).  read_back_radiance() / read_back_encoded() / encode_output() then see that frame. */
VKR_API int finish_slab_exchange(application_t* app, slab_exchange_t* exchange);
/*! Durations in milliseconds of the most recent timed frame that has completed:
	{shading (+ encoding) of the slab, all-gather, scatter into the frame}.  Blocks until that
	frame is done.  Returns 0 if no frame has been timed yet, else 1. */
VKR_API uint32_t get_slab_exchange_milliseconds(slab_exchange_t* exchange, float out_milliseconds[3]);

#endif
