/* HIP context and memory helpers behind device_t.  Plain C on top of the HIP
 * runtime's C API. */
#include "vkr_internal.h"
#include <hip/hip_runtime_api.h>

static int check(hipError_t error, const char* what) {
	if (error == hipSuccess) return 0;
	printf("HIP error while %s: %s\n", what, hipGetErrorString(error));
	return 1;
}

/* Makes sure that the first `count` frame streams exist (and have run their first kernel).
   Tuning knob VKR_FRAME_STREAM_PRIORITY=low: the frame streams get the lowest priority, so
   that the small kernels which consume a frame on device->stream (output encoding, slab
   assembly) get their waves before the next frame's shading kernel fills the chip (measured:
   encode 50 -> 16 us, assemble 110 -> 10 us beside config 2's kernels).  Not the default:
   the frame rate did not change at config 2 and fell by 7 % at config 3 with an exchange
   per frame. */
int vkr_ensure_frame_streams(device_t* device, uint32_t count) {
	if (count > VKR_MAX_FRAMES_IN_FLIGHT) count = VKR_MAX_FRAMES_IN_FLIGHT;
	int least_priority = 0, greatest_priority = 0;
	const char* knob = getenv("VKR_FRAME_STREAM_PRIORITY");
	if (knob && strcmp(knob, "low") == 0) (void) hipDeviceGetStreamPriorityRange(&least_priority, &greatest_priority);
	for (uint32_t i = 0; i != count; ++i) {
		if (device->frame_streams[i]) continue;
		hipStream_t stream = NULL;
		if (check(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, least_priority), "creating a frame stream")) return 1;
		device->frame_streams[i] = stream;
		/* The first kernel a process launches on a stream pays for the stream's hardware queue and for
		   loading the code object: several milliseconds that would otherwise be billed to whatever
		   comes first.  An empty kernel moves them here.  VKR_NO_WARM_UP=1 leaves it out. */
		if (!getenv("VKR_NO_WARM_UP") && (vkr_launch_empty_kernel(stream) || check(hipStreamSynchronize(stream), "warming a frame stream up"))) return 1;
	}
	return 0;
}

int create_hip_device(device_t* device, int32_t hip_device, void* existing_stream) {
	memset(device, 0, sizeof(*device));
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		printf("No HIP device is available. The shading pass needs an MI355X (gfx950).\n");
		return 1;
	}
	if (hip_device < 0 || hip_device >= count) {
		printf("HIP device %d was requested but only %d devices exist.\n", hip_device, count);
		return 1;
	}
	if (check(hipSetDevice(hip_device), "selecting the device")) return 1;
	hipDeviceProp_t properties;
	if (check(hipGetDeviceProperties(&properties, hip_device), "querying device properties")) return 1;
	device->hip_device = hip_device;
	device->stream = existing_stream;
	device->ray_tracing_supported = VK_TRUE;
	device->compute_unit_count = properties.multiProcessorCount;
	strncpy(device->architecture, properties.gcnArchName, sizeof(device->architecture) - 1);
	if (strncmp(device->architecture, "gfx950", 6) != 0)
		printf("Warning: the kernels are built for gfx950 but device %d is %s.\n", hip_device, device->architecture);
	/* Frame streams: four now, the others when a pass first asks for more frames in flight (vkr_ensure_frame_streams).
	   The runtime deals streams onto GPU_MAX_HW_QUEUES hardware queues (8 in bench.py's environment, 4 by default), each
	   served in order: streams nobody uses should not share the queues of the busy ones.  (Round 6 suspected that of the
	   slab exchange's stream and measured eight against four streams: no difference, profiles/r10e/exchange_overhead.jsonl -
	   the queues were not what the exchange loses its time to.)  VKR_FRAME_STREAMS overrides the four. */
	const char* stream_knob = getenv("VKR_FRAME_STREAMS");
	long initial_streams = stream_knob ? strtol(stream_knob, NULL, 10) : 4;
	if (initial_streams < 0) initial_streams = 0;
	if (initial_streams > VKR_MAX_FRAMES_IN_FLIGHT) initial_streams = VKR_MAX_FRAMES_IN_FLIGHT;
	if (vkr_ensure_frame_streams(device, (uint32_t) initial_streams)) {
		destroy_hip_device(device);
		return 1;
	}
	/* tables the kernels read from device globals exist before anything can be launched on this device_t */
	if (vkr_fill_device_tables(device->stream)) {
		destroy_hip_device(device);
		return 1;
	}
	/* The first kernel a process launches on a stream pays for the stream's hardware queue and for
	   loading the code object: several milliseconds that would otherwise be billed to whatever
	   comes first (the BVH build of the first scene: 7.8 instead of 3.5 ms).  An empty kernel on
	   every stream moves them here, to the creation of the device.  VKR_NO_WARM_UP=1 leaves it out. */
	if (!getenv("VKR_NO_WARM_UP")) {
		if (vkr_launch_empty_kernel(device->stream) || wait_for_device(device)) {
			printf("Launching a kernel on the new device failed.\n");
			destroy_hip_device(device);
			return 1;
		}
	}
	return 0;
}

void destroy_hip_device(device_t* device) {
	for (int i = 0; i != VKR_MAX_FRAMES_IN_FLIGHT; ++i)
		if (device->frame_streams[i]) {
			(void) hipStreamSynchronize((hipStream_t) device->frame_streams[i]);
			(void) hipStreamDestroy((hipStream_t) device->frame_streams[i]);
		}
	memset(device, 0, sizeof(*device));
}

int wait_for_device(const device_t* device) {
	int failed = 0;
	for (int i = 0; i != VKR_MAX_FRAMES_IN_FLIGHT; ++i)
		if (device->frame_streams[i]) failed |= check(hipStreamSynchronize((hipStream_t) device->frame_streams[i]), "waiting for a frame stream");
	return failed | check(hipStreamSynchronize((hipStream_t) device->stream), "waiting for the stream");
}

int vkr_device_alloc(void** out, const device_t* device, size_t size, const char* what) {
	*out = NULL;
	if (!device) return 0;
	if (hipMalloc(out, size ? size : 1) != hipSuccess) {
		printf("Failed to allocate %llu bytes of device memory for %s.\n", (unsigned long long) size, what);
		*out = NULL;
		return 1;
	}
	return 0;
}

void vkr_device_free(void* pointer, const device_t* device) {
	(void) device;
	if (pointer) hipFree(pointer);
}

int vkr_device_upload(void** out, const device_t* device, const void* host, size_t size, const char* what) {
	if (vkr_device_alloc(out, device, size, what)) return 1;
	if (!device) return 0;
	if (check(hipMemcpy(*out, host, size, hipMemcpyHostToDevice), what)) {
		hipFree(*out);
		*out = NULL;
		return 1;
	}
	return 0;
}

int vkr_host_alloc_pinned(void** out, size_t size) {
	if (hipHostMalloc(out, size, hipHostMallocDefault) != hipSuccess) { *out = NULL; return 1; }
	return 0;
}

void vkr_host_free_pinned(void* pointer) {
	if (pointer) hipHostFree(pointer);
}

int vkr_copy_to_device_async(void* device_pointer, const void* host, size_t size, const device_t* device) {
	return check(hipMemcpyAsync(device_pointer, host, size, hipMemcpyHostToDevice, (hipStream_t) device->stream), "uploading");
}

int vkr_copy_to_host(void* host, const void* device_pointer, size_t size, const device_t* device) {
	if (check(hipMemcpyAsync(host, device_pointer, size, hipMemcpyDeviceToHost, (hipStream_t) device->stream), "reading back")) return 1;
	return check(hipStreamSynchronize((hipStream_t) device->stream), "reading back");
}
