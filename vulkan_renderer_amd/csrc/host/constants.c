/* Per-frame constants, render settings and quicksaves: the parts of reference
 * src/main.c that feed the shading pass (write_constants :2114-2188, vertex-count
 * helpers :173-216, defaults :232-249, quick_save/quick_load :49-130). */
#include "vkr_internal.h"
#include "vkr_experiments.h"
#include "vkr_slab_exchange.h"

/* Cofactor inverse with the term order of reference math_utilities.h:24-47:
   entry (i, j) is (-1)^(i+j) times the 3x3 minor without row j and column i,
   expanded along its first column; products and sums run left to right. */
void vkr_matrix_inverse(float inverse[4][4], const float matrix[4][4]) {
	for (uint32_t i = 0; i != 4; ++i)
		for (uint32_t j = 0; j != 4; ++j) {
			uint32_t rows[3], cols[3];
			for (uint32_t k = 0, n = 0; k != 4; ++k) if (k != j) rows[n++] = k;
			for (uint32_t k = 0, n = 0; k != 4; ++k) if (k != i) cols[n++] = k;
#define A(r, c) matrix[rows[r]][cols[c]]
			float minor = A(0, 0) * A(1, 1) * A(2, 2) - A(0, 0) * A(1, 2) * A(2, 1) - A(1, 0) * A(0, 1) * A(2, 2)
				+ A(1, 0) * A(0, 2) * A(2, 1) + A(2, 0) * A(0, 1) * A(1, 2) - A(2, 0) * A(0, 2) * A(1, 1);
#undef A
			inverse[i][j] = ((i + j) & 1) ? -minor : minor;
		}
	float det = matrix[0][0] * inverse[0][0] + matrix[0][1] * inverse[1][0] + matrix[0][2] * inverse[2][0] + matrix[0][3] * inverse[3][0];
	float rcp_det = 1.0f / det;
	for (uint32_t i = 0; i != 4; ++i)
		for (uint32_t j = 0; j != 4; ++j)
			inverse[i][j] = inverse[i][j] * rcp_det;
}

void specify_default_render_settings(render_settings_t* settings) {
	memset(settings, 0, sizeof(*settings));
	settings->exposure_factor = 8.0f;
	settings->roughness_factor = 1.0f;
	settings->sample_count = 1;
	settings->sampling_strategies = sampling_strategies_diffuse_specular_mis;
	settings->mis_heuristic = mis_heuristic_optimal_clamped;
	settings->mis_visibility_estimate = 0.5f;
	settings->polygon_sampling_technique = sample_polygon_projected_solid_angle;
	settings->error_display = error_display_none;
	settings->error_min_exponent = -7.0f;
	settings->trace_shadow_rays = VK_TRUE;
	settings->show_polygonal_lights = VK_TRUE;
	settings->noise_type = noise_type_ahmed;
	settings->animate_noise = VK_TRUE;
	settings->v_sync = VK_TRUE;
	settings->show_gui = VK_TRUE;
}

/* The attic scene, the camera of the reference's start-up and one unit square light of unit flux
   that faces sideways; a quicksave, if there is one, replaces camera and lights. */
void specify_default_scene(scene_specification_t* scene) {
	memset(scene, 0, sizeof(*scene));
	scene->file_path = vkr_copy_string(g_scene_paths[scene_attic][1]);
	scene->texture_path = vkr_copy_string(g_scene_paths[scene_attic][2]);
	scene->quick_save_path = vkr_copy_string(g_scene_paths[scene_attic][3]);
	scene->camera.near = 0.05f;
	scene->camera.far = 1.0e3f;
	scene->camera.vertical_fov = 0.33f * VKR_PI_F;
	scene->camera.rotation_x = 0.43f * VKR_PI_F;
	scene->camera.rotation_z = 1.3f * VKR_PI_F;
	scene->camera.position_world_space[0] = -3.0f;
	scene->camera.position_world_space[1] = -2.0f;
	scene->camera.position_world_space[2] = 1.65f;
	scene->camera.speed = 2.0f;
	scene->polygonal_lights = (polygonal_light_t*) calloc(1, sizeof(polygonal_light_t));
	scene->polygonal_light_count = 1;
	polygonal_light_t* light = &scene->polygonal_lights[0];
	light->rotation_angles[0] = 0.5f * VKR_PI_F;
	light->scaling_x = light->scaling_y = 1.0f;
	for (uint32_t i = 0; i != 3; ++i) light->radiant_flux[i] = 1.0f;
	set_polygonal_light_vertex_count(light, 4);
	const float corners[4][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {1.0f, 1.0f}, {0.0f, 1.0f}};
	for (uint32_t i = 0; i != 4; ++i) {
		light->vertices_plane_space[4 * i] = corners[i][0];
		light->vertices_plane_space[4 * i + 1] = corners[i][1];
	}
	quick_load(scene, NULL);
}

uint32_t get_min_polygonal_light_vertex_count(const scene_specification_t* spec) {
	if (!spec->polygonal_light_count) return 3;
	uint32_t minimum = 0x7FFFFFFF;
	for (uint32_t i = 0; i != spec->polygonal_light_count; ++i)
		if (spec->polygonal_lights[i].vertex_count < minimum) minimum = spec->polygonal_lights[i].vertex_count;
	return minimum;
}

uint32_t get_max_polygonal_light_vertex_count(const scene_specification_t* spec) {
	uint32_t maximum = 3;
	for (uint32_t i = 0; i != spec->polygonal_light_count; ++i)
		if (spec->polygonal_lights[i].vertex_count > maximum) maximum = spec->polygonal_lights[i].vertex_count;
	return maximum;
}

uint32_t get_max_polygon_vertex_count(const scene_specification_t* spec, const render_settings_t* settings) {
	uint32_t light_maximum = get_max_polygonal_light_vertex_count(spec);
	switch (settings->polygon_sampling_technique) {
	/* techniques that clip against the horizon may gain one vertex */
	case sample_polygon_clipped_solid_angle:
	case sample_polygon_bilinear_cosine_warp_clipping_hart:
	case sample_polygon_biquadratic_cosine_warp_clipping_hart:
	case sample_polygon_projected_solid_angle_arvo:
	case sample_polygon_projected_solid_angle:
	case sample_polygon_projected_solid_angle_biased:
		return light_maximum + 1;
	default:
		return light_maximum;
	}
}

void destroy_scene_specification(scene_specification_t* scene) {
	free(scene->file_path);
	free(scene->texture_path);
	free(scene->quick_save_path);
	for (uint32_t i = 0; i != scene->polygonal_light_count; ++i)
		destroy_polygonal_light(&scene->polygonal_lights[i]);
	free(scene->polygonal_lights);
	memset(scene, 0, sizeof(*scene));
}

/* Quicksave layout (64-bit ABI): camera struct, u32 legacy count, u32 light
   count, then per light 88 bytes of the struct, size_t path size, path bytes, two
   8-byte null pointers and 4 floats per plane-space vertex. */
void quick_save(scene_specification_t* scene) {
	FILE* file = fopen(scene->quick_save_path, "wb");
	if (!file) {
		printf("Quick save failed. Please check path and permissions: %s\n", scene->quick_save_path);
		return;
	}
	fwrite(&scene->camera, sizeof(scene->camera), 1, file);
	uint32_t legacy_count = 0;
	fwrite(&legacy_count, sizeof(uint32_t), 1, file);
	fwrite(&scene->polygonal_light_count, sizeof(uint32_t), 1, file);
	for (uint32_t i = 0; i != scene->polygonal_light_count; ++i) {
		const polygonal_light_t* light = &scene->polygonal_lights[i];
		fwrite(light, POLYGONAL_LIGHT_QUICKSAVE_SIZE, 1, file);
		size_t path_size = light->texture_file_path ? strlen(light->texture_file_path) + 1 : 0;
		fwrite(&path_size, sizeof(path_size), 1, file);
		if (path_size) fwrite(light->texture_file_path, 1, path_size, file);
		const void* null_pointers[2] = {NULL, NULL};
		fwrite(null_pointers, sizeof(void*), 2, file);
		fwrite(light->vertices_plane_space, sizeof(float), 4 * (size_t) light->vertex_count, file);
	}
	fclose(file);
}

void quick_load(scene_specification_t* scene, VkBool32* light_count_changed) {
	FILE* file = fopen(scene->quick_save_path, "rb");
	if (!file) {
		printf("Failed to load a quick save. Please check path and permissions: %s\n", scene->quick_save_path);
		return;
	}
	int ok = fread(&scene->camera, sizeof(scene->camera), 1, file) == 1;
	uint32_t legacy_count = 0, new_count = 0;
	ok = ok && fread(&legacy_count, sizeof(uint32_t), 1, file) == 1 && fread(&new_count, sizeof(uint32_t), 1, file) == 1;
	if (!ok || new_count > 65536) {
		printf("The quick save at %s is damaged.\n", scene->quick_save_path);
		fclose(file);
		return;
	}
	uint32_t old_count = scene->polygonal_light_count;
	polygonal_light_t* old_lights = scene->polygonal_lights;
	VkBool32 vertex_count_changed = VK_FALSE;
	polygonal_light_t* lights = (polygonal_light_t*) calloc(new_count ? new_count : 1, sizeof(polygonal_light_t));
	uint32_t loaded = 0;
	for (; loaded != new_count && ok; ++loaded) {
		polygonal_light_t* light = &lights[loaded];
		ok = fread(light, POLYGONAL_LIGHT_QUICKSAVE_SIZE, 1, file) == 1;
		if (!ok) break;
		uint32_t vertex_count = light->vertex_count;
		if (loaded < old_count && vertex_count != old_lights[loaded].vertex_count) vertex_count_changed = VK_TRUE;
		/* legacy files stored a single scaling */
		if (light->scaling_y <= 0.0f) light->scaling_y = light->scaling_x;
		size_t path_size = 0;
		ok = fread(&path_size, sizeof(path_size), 1, file) == 1 && path_size < 65536 && vertex_count >= 3 && vertex_count < 4096;
		if (!ok) break;
		if (path_size) {
			light->texture_file_path = (char*) malloc(path_size);
			ok = fread(light->texture_file_path, 1, path_size, file) == path_size;
			if (ok) light->texture_file_path[path_size - 1] = 0;
		}
		void* stale_pointers[2];
		ok = ok && fread(stale_pointers, sizeof(void*), 2, file) == 2;
		light->vertex_count = 0;
		set_polygonal_light_vertex_count(light, vertex_count);
		ok = ok && fread(light->vertices_plane_space, sizeof(float), 4 * (size_t) vertex_count, file) == 4 * (size_t) vertex_count;
	}
	fclose(file);
	if (!ok) {
		printf("The quick save at %s is damaged.\n", scene->quick_save_path);
		for (uint32_t i = 0; i <= loaded && i < new_count; ++i) destroy_polygonal_light(&lights[i]);
		free(lights);
		return;
	}
	for (uint32_t i = 0; i != old_count; ++i) destroy_polygonal_light(&old_lights[i]);
	free(old_lights);
	scene->polygonal_lights = lights;
	scene->polygonal_light_count = new_count;
	if (light_count_changed)
		*light_count_changed |= (old_count != new_count) || vertex_count_changed;
}

size_t get_constant_buffer_size(const application_t* app) {
	const scene_specification_t* spec = &app->scene_specification;
	size_t vmax = get_max_polygonal_light_vertex_count(spec);
	size_t per_light = POLYGONAL_LIGHT_FIXED_CONSTANT_BUFFER_SIZE + sizeof(float) * 4 * (vmax * 2 + (vmax - 2));
	size_t count = spec->polygonal_light_count ? spec->polygonal_light_count : 1;
	return sizeof(per_frame_constants_t) + per_light * count;
}

void write_constants(void* data, application_t* app) {
	const scene_t* scene = &app->scene;
	const first_person_camera_t* camera = &app->scene_specification.camera;
	const render_settings_t* settings = &app->render_settings;
	per_frame_constants_t constants;
	memset(&constants, 0, sizeof(constants));
	for (uint32_t i = 0; i != 3; ++i) {
		constants.mesh_dequantization_factor[i] = scene->mesh.dequantization_factor[i];
		constants.mesh_dequantization_summand[i] = scene->mesh.dequantization_summand[i];
		constants.camera_position_world_space[i] = camera->position_world_space[i];
	}
	constants.mis_visibility_estimate = settings->mis_visibility_estimate;
	constants.viewport_size = app->swapchain.extent;
	constants.ltc_constants = app->ltc_table.constants;
	constants.error_factor = powf(10.0f, -settings->error_min_exponent);
	constants.exposure_factor = settings->exposure_factor;
	constants.roughness_factor = settings->roughness_factor;
	constants.frame_bits = app->screenshot.frame_bits;
	set_noise_constants(constants.noise_resolution_mask, &constants.noise_texture_index_mask, constants.noise_random_numbers,
		&app->noise_table, settings->animate_noise && (app->screenshot.frame_bits == 0));
	float width = (float) app->swapchain.extent.width, height = (float) app->swapchain.extent.height;
	get_world_to_projection_space(constants.world_to_projection_space, camera, width / height);
	/* pixel index -> world-space direction through the pixel centre (main.c:2134-2157) */
	float to_ndc[4];
	to_ndc[0] = 2.0f / app->swapchain.extent.width;
	to_ndc[1] = 2.0f / app->swapchain.extent.height;
	to_ndc[2] = 0.5f * to_ndc[0] - 1.0f;
	to_ndc[3] = 0.5f * to_ndc[1] - 1.0f;
	float rotation_only[4][4], back_to_world[4][4];
	memcpy(rotation_only, constants.world_to_projection_space, sizeof(rotation_only));
	rotation_only[0][3] = rotation_only[1][3] = rotation_only[2][3] = 0.0f;
	vkr_matrix_inverse(back_to_world, rotation_only);
	float pixel_to_projection[4][3] = {
		{to_ndc[0], 0.0f, to_ndc[2]},
		{0.0f, to_ndc[1], to_ndc[3]},
		{0.0f, 0.0f, 1.0f},
		{0.0f, 0.0f, 1.0f},
	};
	for (uint32_t i = 0; i != 3; ++i)
		for (uint32_t j = 0; j != 3; ++j)
			for (uint32_t k = 0; k != 4; ++k)
				constants.pixel_to_ray_direction_world_space[i][j] += back_to_world[i][k] * pixel_to_projection[k][j];
	memcpy(data, &constants, sizeof(constants));
	/* packed light array (main.c:2159-2187) */
	char* cursor = ((char*) data) + sizeof(per_frame_constants_t);
	uint32_t vmax = get_max_polygonal_light_vertex_count(&app->scene_specification);
	/* keeps texture_index of every light current (reference main.c:2167 does this per light) */
	create_and_assign_light_textures(NULL, &app->device, &app->scene_specification);
	for (uint32_t i = 0; i != app->scene_specification.polygonal_light_count; ++i) {
		polygonal_light_t* light = &app->scene_specification.polygonal_lights[i];
		update_polygonal_light(light);
		memcpy(cursor, light, POLYGONAL_LIGHT_FIXED_CONSTANT_BUFFER_SIZE);
		cursor += POLYGONAL_LIGHT_FIXED_CONSTANT_BUFFER_SIZE;
		const float* vertex_arrays[2] = {light->vertices_plane_space, light->vertices_world_space};
		for (uint32_t j = 0; j != 2; ++j) {
			memset(cursor, 0, sizeof(float) * 4 * vmax);
			memcpy(cursor, vertex_arrays[j], sizeof(float) * 4 * light->vertex_count);
			if (light->vertex_count < vmax)
				memcpy(cursor + sizeof(float) * 4 * light->vertex_count, vertex_arrays[j], sizeof(float) * 4);
			cursor += sizeof(float) * 4 * vmax;
		}
		/* fan areas; the tail is padded with the last entry.  (The reference pads with
		   a mis-scaled pointer offset, main.c:2183-2185, an out-of-bounds read that only
		   the out-of-scope area sampler would consume.) */
		uint32_t fan_count = light->vertex_count - 2;
		memcpy(cursor, light->fan_areas, sizeof(float) * 4 * fan_count);
		cursor += sizeof(float) * 4 * fan_count;
		for (uint32_t k = light->vertex_count; k != vmax; ++k) {
			memcpy(cursor, light->fan_areas + 4 * (fan_count - 1), sizeof(float) * 4);
			cursor += sizeof(float) * 4;
		}
	}
}

/* sizeof() of every struct that crosses the C-ABI, in declaration order of the
   headers, so that foreign-language bindings can verify their mirrors. */
VKR_API uint32_t get_abi_struct_sizes(uint64_t* sizes, uint32_t capacity) {
	const uint64_t all[] = {
		sizeof(device_t), sizeof(polygonal_light_t), sizeof(first_person_camera_t), sizeof(ltc_constants_t),
		sizeof(ltc_table_t), sizeof(noise_table_t), sizeof(mesh_t), sizeof(materials_t), sizeof(acceleration_structure_t),
		sizeof(scene_t), sizeof(scene_specification_t), sizeof(render_settings_t), sizeof(per_frame_constants_t),
		sizeof(swapchain_t), sizeof(render_targets_t), sizeof(screenshot_t), sizeof(tile_schedule_t), sizeof(light_textures_t),
		sizeof(shading_pass_t), sizeof(application_t), sizeof(experiment_t), sizeof(experiment_list_t),
		sizeof(slab_exchange_id_t), sizeof(slab_exchange_t)};
	uint32_t count = (uint32_t) VKR_COUNT_OF(all);
	for (uint32_t i = 0; i != count && i != capacity; ++i) sizes[i] = all[i];
	return count;
}

/* Host-side description of the slab layout that render_shading_pass() writes for
   app->tile_schedule with the given rank: pixel (x, y) of every slab slot, or
   0xFFFFFFFF twice for padding slots.  Same arithmetic as locate_pixel() in
   csrc/shading_kernel.h; lets hosts (and the CPU tests of the multi-process path)
   scatter gathered slabs without touching a GPU. */
VKR_API uint64_t get_slab_pixel_coordinates(const application_t* app, uint32_t rank, uint32_t* out_xy, uint64_t capacity) {
	uint32_t width = app->swapchain.extent.width, height = app->swapchain.extent.height;
	uint32_t rank_count = app->tile_schedule.rank_count > 1 ? app->tile_schedule.rank_count : 1;
	uint32_t tile_size = app->tile_schedule.tile_size < 16 ? 16 : app->tile_schedule.tile_size;
	tile_size = (tile_size + 15) & ~15u;
	if (rank_count == 1) rank = 0;
	uint32_t tiles_x = (width + tile_size - 1) / tile_size, tiles_y = (height + tile_size - 1) / tile_size;
	uint32_t tile_count = tiles_x * tiles_y;
	uint32_t own_tiles = (tile_count + rank_count - 1 - rank) / rank_count;
	uint64_t slots = (uint64_t) own_tiles * tile_size * tile_size;
	for (uint64_t s = 0; s != slots && s != capacity; ++s) {
		uint32_t local_tile = (uint32_t) (s / ((uint64_t) tile_size * tile_size));
		uint32_t within = (uint32_t) (s % ((uint64_t) tile_size * tile_size));
		uint32_t tile = local_tile * rank_count + rank;
		uint32_t px = (tile % tiles_x) * tile_size + within % tile_size;
		uint32_t py = (tile / tiles_x) * tile_size + within / tile_size;
		int valid = tile < tile_count && px < width && py < height;
		out_xy[2 * s + 0] = valid ? px : 0xFFFFFFFFu;
		out_xy[2 * s + 1] = valid ? py : 0xFFFFFFFFu;
	}
	return slots;
}
