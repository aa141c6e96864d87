/* Internal helpers shared by the host-side C files of libvkr_shading.so. */
#ifndef VKR_INTERNAL_H
#define VKR_INTERNAL_H
#include "vkr_shading_pass.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define VKR_PI_F 3.1415926535897932384626433832795f
#define VKR_COUNT_OF(a) (sizeof(a) / sizeof((a)[0]))

#ifdef __cplusplus
extern "C" {
#endif

/*! malloc'ed copy of a string, NULL for NULL */
char* vkr_copy_string(const char* s);
/*! malloc'ed concatenation of count strings */
char* vkr_concatenate(uint32_t count, const char* const* pieces);

/*! Device memory helpers (device.c).  All return 0 on success and print the
	HIP error otherwise.  A NULL device makes uploads a no-op that yields NULL. */
int vkr_device_alloc(void** out, const device_t* device, size_t size, const char* what);
void vkr_device_free(void* pointer, const device_t* device);
int vkr_device_upload(void** out, const device_t* device, const void* host, size_t size, const char* what);
int vkr_host_alloc_pinned(void** out, size_t size);
/* lbvh_build.hip: an empty kernel on `stream` (create_hip_device() warms the queues up with it) */
int vkr_launch_empty_kernel(void* stream);
/* fills the per-device tables of the kernels (sRGB code thresholds) and waits for it (shading_pass.hip) */
int vkr_fill_device_tables(void* stream);
void vkr_host_free_pinned(void* pointer);
int vkr_copy_to_device_async(void* device_pointer, const void* host, size_t size, const device_t* device);
int vkr_copy_to_host(void* host, const void* device_pointer, size_t size, const device_t* device);

/*! textures.c: a material texture decoded to RGBA8, all mip levels one after the other */
typedef struct vkr_host_texture_s {
	uint32_t width, height, mip_count, srgb;
	uint8_t* texels;
	uint64_t texel_count;
} vkr_host_texture_t;
/*! 0 on success, 1 if the file is absent, 2 if it is invalid, 3 if its format is not decoded here */
int vkr_load_texture_rgba8(vkr_host_texture_t* out, const char* path);
void vkr_free_host_texture(vkr_host_texture_t* texture);
void vkr_decode_bc1_block(const uint8_t block[8], uint8_t out_rgba[64], int has_alpha);
void vkr_decode_bc5_block(const uint8_t block[16], uint8_t out_rgba[64]);
/*! scene.c: the sRGB -> linear table of the texture samplers (identical to oracle_srgb_table) */
void vkr_fill_srgb_table(float table[256]);

/*! 4x4 inverse with the operation order of reference math_utilities.h:24-47 */
void vkr_matrix_inverse(float inverse[4][4], const float matrix[4][4]);
/*! reference math_utilities.h:50-57 */
uint32_t vkr_wang_random_number(uint32_t seed);

/*! Builds the BVH over the mesh with the given acceleration_structure_builder_t
	(include/vkr_scene.h) and derives both traversal layouts of csrc/lbvh.h from it */
int vkr_build_acceleration_structure(acceleration_structure_t* structure, const device_t* device, const mesh_t* mesh, int builder);
/*! sah_bvh.c: malloc'ed nodes (8 floats each, depth-first) and triangles (12 floats per leaf slot) */
int vkr_build_sah_bvh_host(const mesh_t* mesh, float pad, float** out_nodes, float** out_triangles, uint32_t* out_node_count);
void vkr_destroy_acceleration_structure(acceleration_structure_t* structure, const device_t* device);

/*! shading_pass.hip: scatters all-gathered slabs (format: slab_format_t of vkr_slab_exchange.h)
	into a frame on the given hipStream_t */
int vkr_assemble_slabs_on_stream(application_t* app, const void* gathered_slabs, void* out_frame, int format, void* stream);
/* (shading_pass.hip) `event`, a hipEvent_t, marks the end of a reader of [target, target + bytes): the next frame that
   writes that range waits for it in front of the kernel that writes */
void vkr_note_target_reader(application_t* app, void* event, const void* target, size_t bytes);
void vkr_forget_target_reader(application_t* app, void* event);
/* (device.c) creates the frame streams 0 ... count - 1 of the device that do not exist yet */
int vkr_ensure_frame_streams(device_t* device, uint32_t count);

#ifdef __cplusplus
}
#endif

#endif
