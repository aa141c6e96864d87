/* Polygonal light bookkeeping, pure C.  Follows reference src/polygonal_light.c
 * operation for operation so that the derived members come out bit-identical
 * (checked against the reference object file in tests/test_host_boundary.py). */
#include "vkr_internal.h"

char* vkr_copy_string(const char* s) {
	if (!s) return NULL;
	size_t n = strlen(s) + 1;
	char* r = (char*) malloc(n);
	memcpy(r, s, n);
	return r;
}

char* vkr_concatenate(uint32_t count, const char* const* pieces) {
	size_t total = 1;
	for (uint32_t i = 0; i != count; ++i) total += strlen(pieces[i]);
	char* r = (char*) malloc(total);
	r[0] = 0;
	for (uint32_t i = 0; i != count; ++i) strcat(r, pieces[i]);
	return r;
}

static float* zeroed_floats(size_t count) {
	float* p = (float*) malloc(sizeof(float) * (count ? count : 1));
	memset(p, 0, sizeof(float) * (count ? count : 1));
	return p;
}

/* reference polygonal_light.c:26-43 (including its habit of always returning 0
   after the count has been stored) */
int set_polygonal_light_vertex_count(polygonal_light_t* light, uint32_t vertex_count) {
	if (vertex_count == light->vertex_count && light->vertices_plane_space && light->vertices_world_space && light->fan_areas)
		return 0;
	float* plane = zeroed_floats(4 * (size_t) vertex_count);
	if (light->vertices_plane_space) {
		uint32_t keep = vertex_count < light->vertex_count ? vertex_count : light->vertex_count;
		memcpy(plane, light->vertices_plane_space, sizeof(float) * 4 * keep);
	}
	free(light->vertices_plane_space);
	free(light->vertices_world_space);
	free(light->fan_areas);
	light->vertices_plane_space = plane;
	light->vertices_world_space = zeroed_floats(4 * (size_t) vertex_count);
	light->fan_areas = zeroed_floats(4 * (size_t) (vertex_count > 2 ? vertex_count - 2 : 0));
	light->vertex_count = vertex_count;
	return vertex_count != light->vertex_count;
}

/* reference polygonal_light.c:46-104 */
void update_polygonal_light(polygonal_light_t* light) {
	light->inv_scaling_x = 1.0f / light->scaling_x;
	light->inv_scaling_y = 1.0f / light->scaling_y;
	/* Euler angles -> plane-to-world rotation (rows) */
	const float* angle = light->rotation_angles;
	float cx = cosf(angle[0]), sx = sinf(angle[0]);
	float cy = cosf(angle[1]), sy = sinf(angle[1]);
	float cz = cosf(angle[2]), sz = sinf(angle[2]);
	float cxsy = cx * sy, sxsy = sx * sy;
	float rot[3][4] = {
		{cy * cz, -cy * sz, -sy, 0.0f},
		{-sxsy * cz + cx * sz, sxsy * sz + cx * cz, -sx * cy, 0.0f},
		{cxsy * cz + sx * sz, -cxsy * sz + sx * cz, cx * cy, 0.0f},
	};
	memcpy(light->rotation, rot, sizeof(rot));
	float scale[2] = {light->scaling_x, light->scaling_y};
	for (uint32_t v = 0; v != light->vertex_count; ++v)
		for (uint32_t axis = 0; axis != 3; ++axis) {
			float w = light->translation[axis];
			for (uint32_t k = 0; k != 2; ++k)
				w += scale[k] * rot[axis][k] * light->vertices_plane_space[4 * v + k];
			light->vertices_world_space[4 * v + axis] = w;
		}
	/* plane through the translation with the rotated z-axis as normal */
	for (uint32_t axis = 0; axis != 3; ++axis) light->plane[axis] = rot[axis][2];
	light->plane[3] = -(rot[0][2] * light->translation[0] + rot[1][2] * light->translation[1] + rot[2][2] * light->translation[2]);
	/* triangle fan around vertex 0 */
	const float* p = light->vertices_plane_space;
	float signed_area = 0.0f;
	for (uint32_t t = 0; t + 2 < light->vertex_count + 0u && light->vertex_count >= 3; ++t) {
		float ax = p[4 * (t + 2) + 0] - p[0], bx = p[4 * (t + 1) + 0] - p[0];
		float ay = p[4 * (t + 2) + 1] - p[1], by = p[4 * (t + 1) + 1] - p[1];
		float triangle_area = 0.5f * (ax * by - bx * ay);
		signed_area += triangle_area;
		float flip = (triangle_area < 0.0f) ? -1.0f : 1.0f;
		light->fan_areas[4 * t + 0] = scale[0] * scale[1] * triangle_area;
		light->fan_areas[4 * t + 1] = scale[0] * scale[1] * signed_area;
		light->fan_areas[4 * t + 0] *= flip;
		light->fan_areas[4 * t + 1] *= flip;
	}
	signed_area *= scale[0] * scale[1];
	float abs_area = (signed_area < 0.0f) ? -signed_area : signed_area;
	light->area = abs_area;
	light->rcp_area = 1.0f / abs_area;
	float flux_to_radiance = 1.0f / (abs_area * VKR_PI_F);
	for (uint32_t c = 0; c != 3; ++c) light->surface_radiance[c] = light->radiant_flux[c] * flux_to_radiance;
	/* the plane normal must face the side from which the winding is counter-clockwise */
	for (uint32_t i = 0; i != 4; ++i) light->plane[i] = (signed_area > 0.0f) ? light->plane[i] : (-light->plane[i]);
}

/* reference polygonal_light.c:107-117 */
polygonal_light_t duplicate_polygonal_light(const polygonal_light_t* light) {
	polygonal_light_t copy = *light;
	copy.texture_file_path = vkr_copy_string(light->texture_file_path);
	copy.vertex_count = 0;
	copy.vertices_plane_space = copy.vertices_world_space = copy.fan_areas = NULL;
	set_polygonal_light_vertex_count(&copy, light->vertex_count);
	memcpy(copy.vertices_plane_space, light->vertices_plane_space, sizeof(float) * 4 * light->vertex_count);
	return copy;
}

/* reference polygonal_light.c:120-126 */
void destroy_polygonal_light(polygonal_light_t* light) {
	free(light->vertices_plane_space);
	free(light->vertices_world_space);
	free(light->fan_areas);
	free(light->texture_file_path);
	memset(light, 0, sizeof(*light));
}
