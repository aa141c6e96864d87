/* The experiment table of the reference (src/experiment_list.c:22-544) as data.
 *
 * The reference fills the list with long runs of struct copies; here every figure of
 * the paper is one row group: a template (scene, resolution, settings) plus the list
 * of variations.  tests/test_experiments.py checks all 321 entries field by field
 * against the reference's own create_experiment_list (oracle/_ref) and against the
 * committed fixture.  Switches of the reference that are fixed here: all_figs and
 * all_timings on, html_figs and take_hdr_screenshots off (experiment_list.c:46-56). */
#include "vkr_internal.h"
#include "vkr_experiments.h"
#include <stdarg.h>

const char* const g_scene_paths[scene_count][4] = {
	[scene_cornell_box] = {"Cornell box", "data/cornell_box.vks", "data/cornell_box_textures", "data/quicksaves/cornell_box.save"},
	[scene_mis_plane] = {"MIS plane", "data/mis_plane.vks", "data/mis_plane_textures", "data/quicksaves/mis_plane.save"},
	[scene_roughness_planes] = {"Roughness planes", "data/roughness_planes.vks", "data/roughness_planes_textures", "data/quicksaves/roughness_planes.save"},
	[scene_shadowed_plane] = {"Shadowed plane", "data/shadowed_plane.vks", "data/shadowed_plane_textures", "data/quicksaves/shadowed_plane.save"},
	[scene_arcade] = {"Arcade", "data/Arcade.vks", "data/Arcade_textures", "data/quicksaves/Arcade.save"},
	[scene_living_room] = {"Living room", "data/living_room.vks", "data/living_room_textures", "data/quicksaves/living_room.save"},
	[scene_attic] = {"Attic", "data/attic.vks", "data/attic_textures", "data/quicksaves/attic.save"},
	[scene_bistro_inside] = {"Bistro inside", "data/Bistro_inside.vks", "data/Bistro_textures", "data/quicksaves/Bistro_inside.save"},
	[scene_bistro_outside] = {"Bistro outside", "data/Bistro_outside.vks", "data/Bistro_textures", "data/quicksaves/Bistro_outside.save"},
};

/* names of the sampling techniques inside file names (experiment_list.c:30-43) */
static const char* const k_technique_names[sample_polygon_count] = {
	[sample_polygon_baseline] = "baseline",
	[sample_polygon_area_turk] = "area_turk",
	[sample_polygon_rectangle_solid_angle_urena] = "rectangle_solid_angle_urena",
	[sample_polygon_solid_angle_arvo] = "solid_angle_arvo",
	[sample_polygon_solid_angle] = "solid_angle_ours",
	[sample_polygon_clipped_solid_angle] = "clipped_solid_angle_ours",
	[sample_polygon_bilinear_cosine_warp_hart] = "bilinear_cosine_warp_hart",
	[sample_polygon_bilinear_cosine_warp_clipping_hart] = "bilinear_cosine_warp_clipping_hart",
	[sample_polygon_biquadratic_cosine_warp_hart] = "biquadratic_cosine_warp_hart",
	[sample_polygon_biquadratic_cosine_warp_clipping_hart] = "biquadratic_cosine_warp_clipping_hart",
	[sample_polygon_projected_solid_angle_arvo] = "projected_solid_angle_arvo",
	[sample_polygon_projected_solid_angle] = "projected_solid_angle_ours",
	[sample_polygon_projected_solid_angle_biased] = "projected_solid_angle_biased_ours",
};

/* names of the MIS heuristics inside file names (experiment_list.c:182-187) */
static const char* const k_heuristic_names[mis_heuristic_count] = {
	[mis_heuristic_balance] = "balance_veach",
	[mis_heuristic_power] = "power_veach",
	[mis_heuristic_weighted] = "weighted_ours",
	[mis_heuristic_optimal_clamped] = "clamped_optimal_ours",
	[mis_heuristic_optimal] = "optimal_ours",
};

static char* format_path(const char* format, ...) {
	va_list arguments;
	va_start(arguments, format);
	int length = vsnprintf(NULL, 0, format, arguments);
	va_end(arguments);
	char* result = (char*) malloc((size_t) length + 1);
	va_start(arguments, format);
	vsnprintf(result, (size_t) length + 1, format, arguments);
	va_end(arguments);
	return result;
}

typedef struct {
	experiment_t* entries;
	uint32_t count, capacity;
} table_t;

/* Appends a copy of the template and returns it for modification.  The screenshot path
   is "data/experiments/" + name + "_%.3f.png"; save is the quicksave path or NULL. */
static experiment_t* add(table_t* table, const experiment_t* template, char* name, char* save) {
	if (table->count == table->capacity) {
		table->capacity = table->capacity ? 2 * table->capacity : 512;
		table->entries = (experiment_t*) realloc(table->entries, sizeof(experiment_t) * table->capacity);
	}
	experiment_t* e = &table->entries[table->count++];
	*e = *template;
	e->screenshot_path = format_path("data/experiments/%s_%%.3f.png", name);
	e->quick_save_path = save;
	free(name);
	return e;
}

/* What all experiments share (every settings block of experiment_list.c): */
static experiment_t make_template(scene_index_t scene, uint32_t width, uint32_t height, float exposure, sampling_strategies_t strategies, VkBool32 rays_and_lights) {
	experiment_t t;
	memset(&t, 0, sizeof(t));
	t.scene_index = scene;
	t.width = width; t.height = height;
	t.render_settings.exposure_factor = exposure;
	t.render_settings.roughness_factor = 1.0f;
	t.render_settings.sample_count = 1;
	t.render_settings.sampling_strategies = strategies;
	t.render_settings.error_min_exponent = -7.0f;
	t.render_settings.noise_type = noise_type_ahmed;
	t.render_settings.animate_noise = VK_FALSE;
	t.render_settings.trace_shadow_rays = rays_and_lights;
	t.render_settings.show_polygonal_lights = rays_and_lights;
	return t;
}

void create_experiment_list(experiment_list_t* list) {
	memset(list, 0, sizeof(*list));
	table_t table = {NULL, 0, 0};
	experiment_t* e;

	/* attic, sampling strategies side by side (experiment_list.c:59-101) */
	{
		experiment_t t = make_template(scene_attic, 1440, 1440, 8.0f, sampling_strategies_diffuse_only, VK_TRUE);
		t.render_settings.mis_heuristic = mis_heuristic_balance;
		t.render_settings.mis_visibility_estimate = 0.5f;
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		e = add(&table, &t, format_path("attic_solid_angle_and_ggx_mis_2spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_ggx_mis;
		e->render_settings.polygon_sampling_technique = sample_polygon_solid_angle;
		e = add(&table, &t, format_path("attic_projected_solid_angle_ours_and_ggx_mis_2spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_ggx_mis;
		e = add(&table, &t, format_path("attic_projected_solid_angle_ours_2spp"), NULL);
		e->render_settings.sample_count = 2;
		e = add(&table, &t, format_path("attic_diffuse_and_specular_ours_clamped_optimal_mis_ours_2spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_specular_mis;
		e->render_settings.mis_heuristic = mis_heuristic_optimal_clamped;
		e = add(&table, &t, format_path("attic_reference_128spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_specular_mis;
		e->render_settings.sample_count = 64;
	}
	/* attic, sampling error (experiment_list.c:104-129).  The reference initialises
	   sampling_strategies with a polygon-sampling enumerator there (value 11); the value
	   is kept because the table is compared entry by entry. */
	{
		experiment_t t = make_template(scene_attic, 1440, 1440, 8.0f, (sampling_strategies_t) sample_polygon_projected_solid_angle, VK_FALSE);
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		e = add(&table, &t, format_path("error_attic_backward"), NULL);
		e->render_settings.error_display = error_display_diffuse_backward;
		e = add(&table, &t, format_path("error_attic_backward_times_psa"), NULL);
		e->render_settings.error_display = error_display_diffuse_backward_scaled;
	}
	/* bistro with small distant lights, every technique (experiment_list.c:132-167) */
	{
		experiment_t t = make_template(scene_bistro_outside, 1920, 1080, 14.0f, sampling_strategies_diffuse_only, VK_TRUE);
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		static const char* const sizes[] = {"small", "tiny"};
		for (uint32_t i = 0; i != VKR_COUNT_OF(sizes); ++i) {
			for (uint32_t j = 0; j != sample_polygon_count; ++j) {
				if (j == sample_polygon_bilinear_cosine_warp_clipping_hart || j == sample_polygon_biquadratic_cosine_warp_clipping_hart) continue;
				e = add(&table, &t, format_path("bistro_%s_polygon_%s_1spp", sizes[i], k_technique_names[j]),
					format_path("data/quicksaves/Bistro_outside_%s_light.save", sizes[i]));
				e->render_settings.polygon_sampling_technique = (sample_polygon_technique_t) j;
			}
			e = add(&table, &t, format_path("bistro_%s_polygon_reference_128spp", sizes[i]),
				format_path("data/quicksaves/Bistro_outside_%s_light.save", sizes[i]));
			e->render_settings.polygon_sampling_technique = sample_polygon_area_turk;
			e->render_settings.sample_count = 128;
		}
	}
	/* MIS plane, every heuristic (experiment_list.c:170-214) */
	{
		experiment_t t = make_template(scene_mis_plane, 1024, 1024, 8.0f, sampling_strategies_diffuse_specular_mis, VK_TRUE);
		t.render_settings.mis_visibility_estimate = 0.5f;
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		for (uint32_t j = 0; j != mis_heuristic_count; ++j) {
			e = add(&table, &t, format_path("mis_plane_%s_2spp", k_heuristic_names[j]), NULL);
			e->render_settings.mis_heuristic = (mis_heuristic_t) j;
		}
		e = add(&table, &t, format_path("mis_plane_solid_angle_and_ggx_balance_veach_2spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_ggx_mis;
		e->render_settings.mis_heuristic = mis_heuristic_balance;
		e = add(&table, &t, format_path("mis_plane_diffuse_and_specular_random_ours_1spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_specular_random;
		e = add(&table, &t, format_path("mis_plane_reference_128spp"), NULL);
		e->render_settings.mis_heuristic = mis_heuristic_balance;
		e->render_settings.sample_count = 64;
	}
	/* Cornell box, every technique (experiment_list.c:217-257) */
	{
		experiment_t t = make_template(scene_cornell_box, 1024, 1024, 8.0f, sampling_strategies_diffuse_only, VK_TRUE);
		for (uint32_t j = 0; j != sample_polygon_count; ++j) {
			e = add(&table, &t, format_path("cornell_box_%s_1spp", k_technique_names[j]), NULL);
			e->render_settings.polygon_sampling_technique = (sample_polygon_technique_t) j;
		}
		e = add(&table, &t, format_path("cornell_box_projected_solid_angle_arvo_tilted_1spp"), format_path("data/quicksaves/cornell_box_tilted_light.save"));
		e->render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle_arvo;
		e = add(&table, &t, format_path("cornell_box_reference_tilted_128spp"), format_path("data/quicksaves/cornell_box_tilted_light.save"));
		e->render_settings.polygon_sampling_technique = sample_polygon_solid_angle;
		e->render_settings.sample_count = 128;
		e = add(&table, &t, format_path("cornell_box_reference_128spp"), NULL);
		e->render_settings.polygon_sampling_technique = sample_polygon_solid_angle;
		e->render_settings.sample_count = 128;
	}
	/* shadowed plane, bias of the biased sampler (experiment_list.c:260-285) */
	{
		experiment_t t = make_template(scene_shadowed_plane, 1024, 1024, 10.0f, sampling_strategies_diffuse_specular_mis, VK_TRUE);
		t.render_settings.sample_count = 2048;
		t.render_settings.mis_heuristic = mis_heuristic_optimal_clamped;
		t.render_settings.mis_visibility_estimate = 0.5f;
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		add(&table, &t, format_path("shadowed_plane_reference_4096spp"), NULL);
		e = add(&table, &t, format_path("shadowed_plane_biased_4096spp"), NULL);
		e->render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle_biased;
	}
	/* attic with an IES profile (experiment_list.c:288-308) */
	{
		experiment_t t = make_template(scene_attic, 1280, 1024, 8.0f, sampling_strategies_diffuse_specular_mis, VK_TRUE);
		t.render_settings.mis_heuristic = mis_heuristic_optimal_clamped;
		t.render_settings.mis_visibility_estimate = 0.5f;
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		add(&table, &t, format_path("ies_profile_attic_2spp"), format_path("data/quicksaves/attic_ies_profile.save"));
	}
	/* roughness planes, Lambertian emitter (experiment_list.c:311-335) */
	{
		experiment_t t = make_template(scene_roughness_planes, 2048 + 256, 1024, 8.0f, sampling_strategies_diffuse_specular_mis, VK_TRUE);
		t.render_settings.mis_heuristic = mis_heuristic_weighted;
		t.render_settings.mis_visibility_estimate = 0.5f;
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		add(&table, &t, format_path("roughness_planes_lambertian_2spp"), NULL);
		e = add(&table, &t, format_path("roughness_planes_lambertian_diffuse_only_1spp"), NULL);
		e->render_settings.sampling_strategies = sampling_strategies_diffuse_only;
	}
	/* roughness planes, textured screen (experiment_list.c:338-358) */
	{
		experiment_t t = make_template(scene_roughness_planes, 1280, 1024, 8.0f, sampling_strategies_diffuse_specular_mis, VK_TRUE);
		t.render_settings.mis_heuristic = mis_heuristic_optimal_clamped;
		t.render_settings.mis_visibility_estimate = 0.5f;
		t.render_settings.polygon_sampling_technique = sample_polygon_projected_solid_angle;
		add(&table, &t, format_path("roughness_planes_screen_2spp"), format_path("data/quicksaves/roughness_planes_screen.save"));
	}
	/* the timing matrix: vertex counts 3..7 x central/decentral x (128 lights, 1 sample |
	   1 light, 128 samples) x every technique (experiment_list.c:361-399) */
	{
		experiment_t t = make_template(scene_roughness_planes, 1920, 1080, 8.0f, sampling_strategies_diffuse_only, VK_FALSE);
		for (uint32_t vertex_count = 3; vertex_count != 8; ++vertex_count)
			for (uint32_t decentral = 0; decentral != 2; ++decentral)
				for (uint32_t many_samples = 0; many_samples != 2; ++many_samples)
					for (uint32_t technique = 0; technique != sample_polygon_count; ++technique) {
						const char* configuration = decentral ? "decentral_" : "central_";
						const char* light_count_suffix = many_samples ? "" : "_128";
						uint32_t light_count = many_samples ? 1 : 128;
						e = add(&table, &t,
							format_path("timings_%s%u%s_%s", configuration, vertex_count, light_count_suffix, k_technique_names[technique]),
							format_path("data/quicksaves/roughness_planes_%s%u%s.save", configuration, vertex_count, light_count_suffix));
						e->render_settings.polygon_sampling_technique = (sample_polygon_technique_t) technique;
						e->render_settings.sample_count = many_samples ? 128 : 1;
						e->render_settings.exposure_factor /= (float) light_count;
					}
	}
	printf("Defined %u experiments to reproduce.\n", table.count);
	list->experiments = table.entries;
	list->count = table.count;
	/* greater than count: no experiment is running (main.h:225-227) */
	list->next = table.count + 1;
	list->next_setup_time = 0.0;
}

void destroy_experiment_list(experiment_list_t* list) {
	for (uint32_t i = 0; i != list->count; ++i) {
		free(list->experiments[i].quick_save_path);
		free(list->experiments[i].screenshot_path);
	}
	free(list->experiments);
	memset(list, 0, sizeof(*list));
}

static char* join_path(const char* root, const char* path) {
	if (!root || !root[0]) return vkr_copy_string(path);
	return format_path("%s/%s", root, path);
}

int apply_experiment(application_t* app, const experiment_t* experiment, const char* data_root) {
	if (!experiment || experiment->scene_index >= scene_count) {
		printf("Invalid experiment.\n");
		return 1;
	}
	scene_specification_t* spec = &app->scene_specification;
	char* file_path = join_path(data_root, g_scene_paths[experiment->scene_index][1]);
	char* texture_path = join_path(data_root, g_scene_paths[experiment->scene_index][2]);
	char* quick_save_path = join_path(data_root, experiment->quick_save_path ? experiment->quick_save_path : g_scene_paths[experiment->scene_index][3]);
	free(spec->file_path); free(spec->texture_path); free(spec->quick_save_path);
	spec->file_path = file_path;
	spec->texture_path = texture_path;
	spec->quick_save_path = quick_save_path;
	/* camera and lights come from the quicksave when there is one (main.c:1914-1918) */
	FILE* probe = fopen(quick_save_path, "rb");
	if (probe) {
		fclose(probe);
		quick_load(spec, NULL);
	}
	app->render_settings = experiment->render_settings;
	if (experiment->width && experiment->height) {
		app->swapchain.extent.width = experiment->width;
		app->swapchain.extent.height = experiment->height;
	}
	return 0;
}

char* format_screenshot_path(const char* format_string, float frame_time_milliseconds) {
	return format_path(format_string, (double) frame_time_milliseconds);
}
