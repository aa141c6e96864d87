/* Scene loader: reference src/scene.h:47-184.  Vulkan buffers become device
 * pointers; the driver-built acceleration structure becomes an LBVH built by HIP
 * kernels; material textures are reduced to one constant texel per texture
 * (texture filtering is out of scope). */
#ifndef VKR_SCENE_H
#define VKR_SCENE_H
#include "vkr_device.h"

/*! reference scene.h:47-116 */
typedef struct mesh_s {
	uint64_t triangle_count;
	float dequantization_factor[3], dequantization_summand[3];
	/*! 2 uint32_t per vertex, 3 vertices per triangle (bit layout: scene.h:56-61) */
	uint32_t* host_positions;
	/*! 4 uint16_t per vertex: octahedral normal xy, texture coordinate xy */
	uint16_t* host_normals_and_tex_coords;
	/*! one uint8_t per triangle */
	uint8_t* host_material_indices;
	void* positions;
	void* normals_and_tex_coords;
	void* material_indices;
} mesh_t;

/*! reference scene.h:119-137 */
typedef enum material_texture_type_e {
	material_texture_type_base_color,
	material_texture_type_specular,
	material_texture_type_normal,
	material_texture_count
} material_texture_type_t;

/*! reference scene.h:143-156.  Half / float textures are reduced to one constant texel,
	8-bit and block-compressed ones are decoded to RGBA8 mip chains. */
typedef struct materials_s {
	uint64_t material_count;
	char** material_names;
	/*! 8 floats per material: base_color.rgb, specular.rgb (occlusion, linear
		roughness, metalicity), normal.xy (0.5, 0.5 is the geometric normal) */
	float* host_constants;
	void* constants;
	/*! VK_TRUE if at least one material texture is a real image (8-bit or BC1 / BC5 *.vkt):
		then the pass samples textures per pixel (anisotropic, up to 16 trilinear taps, in software) instead of constants */
	VkBool32 textured;
	/*! 3 per material, 4 uint32 each: first texel in texels, width, height,
		mip_count | srgb << 16.  Width 0: this texture is the constant in host_constants. */
	uint32_t* host_texture_descriptors;
	/*! RGBA8 texels of all textures and mip levels */
	uint8_t* host_texels;
	uint64_t texel_count;
	void* texture_descriptors;
	void* texels;
	/*! 256 floats: sRGB decoding table used by the sampler */
	void* srgb_table;
} materials_t;

/*! Who builds the BVH (request_acceleration_structure of load_scene()).  The reference asks the
	driver (vkCmdBuildAccelerationStructuresKHR, scene.c:254-262, PREFER_FAST_TRACE). */
typedef enum acceleration_structure_builder_e {
	acceleration_structure_none = 0,
	/*! binned surface-area heuristic, built breadth-first by HIP kernels (lbvh_build.hip); the
		default (VK_TRUE) and the counterpart of PREFER_FAST_TRACE */
	acceleration_structure_sah_device = 1,
	/*! Morton-code LBVH by HIP kernels: the counterpart of PREFER_FAST_BUILD (more node visits per ray) */
	acceleration_structure_lbvh_device = 2,
	/*! the same binned surface-area heuristic on the host (sah_bvh.c), kept as the plain C
		statement of the algorithm that the device builder is checked against */
	acceleration_structure_sah_host = 3,
	acceleration_structure_builder_count
} acceleration_structure_builder_t;

/*! Replaces reference scene.h:161-175: a binary BVH over the de-quantised
	triangle soup (same de-quantisation as scene.c:176-187). */
typedef struct acceleration_structure_s {
	/*! float4 per vertex, 3 per leaf, in leaf order */
	void* triangle_vertices;
	/*! unused (the original triangle index travels in w of the first vertex) */
	void* triangle_indices;
	/*! (2 * leaf_count - 1) nodes of 16 bytes in depth-first order with boxes
		quantised to a 15-bit grid, see vulkan_renderer_amd/csrc/lbvh.h */
	void* nodes;
	uint32_t node_count;
	uint32_t root;
	/*! the grid of the quantised boxes: world = grid_origin + q / grid_inverse_cell */
	float grid_origin[3], grid_inverse_cell[3];
	/*! The same tree collapsed to four children per node for the wavefront shadow-ray kernel:
		64-byte nodes (the children's boxes on the same grid + their links), children of a node
		stored next to each other, see vulkan_renderer_amd/csrc/lbvh.h.  NULL if the collapse
		was not requested or not possible; the kernels then walk `nodes`. */
	void* wide_nodes;
	uint32_t wide_node_count;
	/*! most stack entries a ray can need in the wide tree (sum over a root path of children - 1) */
	uint32_t wide_stack_need;
	/*! acceleration_structure_builder_t that made the tree, and the time from the de-quantised
		triangles on the device to the finished structures (HIP kernels, or host build + upload) */
	uint32_t builder;
	float build_milliseconds;
	/*! Leaves of the tree: the triangle count, or more when the device SAH builder has split long thin triangles
		that lie diagonally in their boxes into several leaves (each names the whole triangle; lbvh_build.hip
		"fragments"; environment VKR_BVH_SPLIT_TRIANGLES=0 turns it off).  node_count = 2 leaf_count - 1. */
	uint32_t leaf_count;
	uint32_t reserved;
} acceleration_structure_t;

/*! reference scene.h:161-166 */
typedef struct scene_s {
	mesh_t mesh;
	materials_t materials;
	acceleration_structure_t acceleration_structure;
} scene_t;

/*! reference scene.h:171 */
VKR_API const char* get_material_texture_suffix(material_texture_type_t type);
/*! reference scene.h:181 / scene.c:409-559.  texture_path/<material>_<suffix>.vkt
	files are read when present (uncompressed float / half formats only; the
	smallest mip level supplies the constant), otherwise defaults apply: base
	colour 0.8, specular (1, 0.5, 0), flat normal.
	request_acceleration_structure: an acceleration_structure_builder_t; VK_FALSE none, VK_TRUE
	the fast-trace build by HIP kernels (the counterpart of PREFER_FAST_TRACE at scene.c:254-262). */
VKR_API int load_scene(scene_t* scene, const device_t* device, const char* file_path, const char* texture_path, VkBool32 request_acceleration_structure);
/*! reference scene.h:184 */
VKR_API void destroy_scene(scene_t* scene, const device_t* device);

#endif
