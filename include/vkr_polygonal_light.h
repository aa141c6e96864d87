/* Polygonal light model: binary-compatible with the reference's
 * src/polygonal_light.h:100-129 (the first 160 bytes are copied verbatim into
 * the constant buffer and the first 88 into quicksaves).  Pure C, no GPU. */
#ifndef VKR_POLYGONAL_LIGHT_H
#define VKR_POLYGONAL_LIGHT_H
#include "vkr_device.h"

/*! Same values as sample_polygon_technique_t, reference polygonal_light.h:30-69.
	All thirteen techniques have kernel variants (the related-work samplers of Turk, Urena,
	Arvo and Hart for the strategies their shader branch exists for); create_shading_pass()
	rejects the combinations the reference's GUI rejects (user_interface.cpp:90-180). */
typedef enum sample_polygon_technique_e {
	sample_polygon_baseline,
	sample_polygon_area_turk,
	sample_polygon_rectangle_solid_angle_urena,
	sample_polygon_solid_angle_arvo,
	sample_polygon_solid_angle,
	sample_polygon_clipped_solid_angle,
	sample_polygon_bilinear_cosine_warp_hart,
	sample_polygon_bilinear_cosine_warp_clipping_hart,
	sample_polygon_biquadratic_cosine_warp_hart,
	sample_polygon_biquadratic_cosine_warp_clipping_hart,
	sample_polygon_projected_solid_angle_arvo,
	sample_polygon_projected_solid_angle,
	sample_polygon_projected_solid_angle_biased,
	sample_polygon_count
} sample_polygon_technique_t;

/*! Same values as polygon_texturing_technique_t, reference polygonal_light.h:75-90.
	All four are implemented (shading_pass.frag.glsl:151-185); a light that uses a texture needs
	create_and_assign_light_textures() before create_shading_pass(). */
typedef enum polygon_texturing_technique_e {
	polygon_texturing_none = 0,
	polygon_texturing_area = 1,
	polygon_texturing_portal = 2,
	polygon_texturing_ies_profile = 3,
	polygon_texturing_count,
	polygon_texturing_force_int = 0x7fffffff
} polygon_texturing_technique_t;

/*! Field for field the reference struct.  update_polygonal_light() fills
	everything that is derived (inverse scalings, radiance, plane, rotation,
	world-space vertices, fan areas). */
typedef struct polygonal_light_s {
	float rotation_angles[3];
	float scaling_x;
	float translation[3];
	float scaling_y;
	float radiant_flux[3];
	float inv_scaling_x;
	float surface_radiance[3];
	float inv_scaling_y;
	float plane[4];
	uint32_t vertex_count;
	polygon_texturing_technique_t texturing_technique;
	uint32_t texture_index;
	uint32_t padding_0;
	float rotation[3][4];
	float area, rcp_area;
	float padding_1[2];
	char* texture_file_path;
	/*! 4 floats per vertex (std140 padding); x, y used */
	float* vertices_plane_space;
	/*! 4 floats per vertex; x, y, z used */
	float* vertices_world_space;
	/*! 4 floats per fan triangle; x = triangle area, y = running fan area */
	float* fan_areas;
} polygonal_light_t;

#define POLYGONAL_LIGHT_QUICKSAVE_SIZE (sizeof(float) * 20 + sizeof(uint32_t) * 2)
#define POLYGONAL_LIGHT_FIXED_CONSTANT_BUFFER_SIZE (POLYGONAL_LIGHT_QUICKSAVE_SIZE + sizeof(uint32_t) * 2 + sizeof(float) * 16)

/*! reference polygonal_light.h:144 / polygonal_light.c:26-43 */
VKR_API int set_polygonal_light_vertex_count(polygonal_light_t* light, uint32_t vertex_count);
/*! reference polygonal_light.h:148 / polygonal_light.c:46-104 */
VKR_API void update_polygonal_light(polygonal_light_t* light);
/*! reference polygonal_light.h:151 / polygonal_light.c:107-117 */
VKR_API polygonal_light_t duplicate_polygonal_light(const polygonal_light_t* light);
/*! reference polygonal_light.h:154 / polygonal_light.c:120-126 */
VKR_API void destroy_polygonal_light(polygonal_light_t* light);

#endif
