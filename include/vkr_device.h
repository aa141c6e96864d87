/* C-ABI of the MI355X shading-pass library (libvkr_shading.so).
 *
 * The reference threads a `const device_t*` (a Vulkan device, reference
 * src/vulkan_basics.h:30-75) through every loader.  Here device_t is a HIP
 * context: the GPU ordinal and the stream all uploads and dispatches are
 * enqueued on.  Loaders accept device == NULL and then only fill the host side of
 * their output (parsing, quantisation), which is what the CPU-only tests use.
 *
 * A handful of Vulkan type names that appear in the reference's loader
 * signatures are kept as plain C typedefs so reference call sites compile
 * unchanged; nothing here depends on Vulkan. */
#ifndef VKR_DEVICE_H
#define VKR_DEVICE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
#define VKR_API extern "C" __attribute__((visibility("default")))
#else
#define VKR_API __attribute__((visibility("default")))
#endif

typedef uint32_t VkBool32;
#ifndef VK_TRUE
#define VK_TRUE 1u
#define VK_FALSE 0u
#endif
typedef struct VkExtent2D { uint32_t width, height; } VkExtent2D;
typedef struct VkExtent3D { uint32_t width, height, depth; } VkExtent3D;

/*! Replaces device_t of the reference (src/vulkan_basics.h:30-75). */
/*! Deepest frame pipeline of a shading pass (shading_pass_t.frames_in_flight) */
#define VKR_MAX_FRAMES_IN_FLIGHT 8

typedef struct device_s {
	/*! HIP device ordinal (LOCAL_RANK in multi-process runs) */
	int32_t hip_device;
	/*! hipStream_t used for uploads and kernels; NULL is the default stream.
		Pass the caller's stream (e.g. torch's current stream) to share it. */
	void* stream;
	/*! Mirrors device_t.ray_tracing_supported (vulkan_basics.h:62): always 1
		once a device exists because the LBVH traversal is a software path */
	VkBool32 ray_tracing_supported;
	/*! Compute units and architecture name reported by HIP, for logs */
	int32_t compute_unit_count;
	char architecture[64];
	/*! Internal hipStream_t on which consecutive frames of a shading pass with
		frames_in_flight >= 2 run in turn, so that the ray tracing of one frame overlaps the
		shading of the next ones (the analogue of the reference's frame queue,
		main.h:353-390).  Owned by the device.  Four exist from the start, the others are made when a pass
		first asks for a deeper pipeline (every stream takes a share of the few hardware queues of the GPU,
		GPU_MAX_HW_QUEUES; csrc/host/device.c). */
	void* frame_streams[VKR_MAX_FRAMES_IN_FLIGHT];
} device_t;

/*! Replaces create_vulkan_device (reference src/vulkan_basics.h:270): selects
	the HIP device and (optionally) adopts an existing stream.
	\return 0 on success, 1 if no usable GPU exists (prints the reason). */
VKR_API int create_hip_device(device_t* device, int32_t hip_device, void* existing_stream);
/*! Replaces destroy_vulkan_device (src/vulkan_basics.h:281) */
VKR_API void destroy_hip_device(device_t* device);
/*! Blocks until all work on device->stream and on the frame streams is done
	(vkDeviceWaitIdle analogue) */
VKR_API int wait_for_device(const device_t* device);

#endif
