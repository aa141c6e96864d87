/* The reference's `vulkan_renderer -e<N>` (src/main.c:2262-2296 parses the flag,
 * startup_application :1896-1946 applies the experiment, advance_experiments :1948-2016
 * renders, times and takes the screenshot) as a plain C99 program on top of the C-ABI of
 * libvkr_shading.so: what a maintainer of the reference would write after swapping the Vulkan
 * back end for this library.  No Python, no PyTorch.
 *
 *     vkr_experiment -e25 [--frames 64] [--white-noise] [--fresnel 51] [--hdr] [data root]
 *
 * The data root is the directory that holds data/ (default: the working directory).  Without
 * -e it renders the default scene with the default settings once and stores data/default.png. */
#include "vkr_experiments.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int compare_floats(const void* a, const void* b) {
	float x = *(const float*) a, y = *(const float*) b;
	return (x > y) - (x < y);
}

int main(int argc, char** argv) {
	int experiment_index = -1, frame_count = 64, white_noise = 0, hdr = 0;
	uint32_t fresnel_count = 51;
	const char* data_root = ".";
	for (int i = 1; i < argc; ++i) {
		if (strncmp(argv[i], "-e", 2) == 0 && argv[i][2]) experiment_index = atoi(argv[i] + 2);
		else if (strcmp(argv[i], "--frames") == 0 && i + 1 < argc) frame_count = atoi(argv[++i]);
		else if (strcmp(argv[i], "--fresnel") == 0 && i + 1 < argc) fresnel_count = (uint32_t) atoi(argv[++i]);
		else if (strcmp(argv[i], "--white-noise") == 0) white_noise = 1;
		else if (strcmp(argv[i], "--hdr") == 0) hdr = 1;
		else if (strcmp(argv[i], "--help") == 0 || strcmp(argv[i], "-h") == 0) {
			printf("usage: %s [-e<experiment index>] [--frames n] [--white-noise] [--fresnel n] [--hdr] [data root]\n", argv[0]);
			return 0;
		}
		else data_root = argv[i];
	}
	if (frame_count < 1) frame_count = 1;
	if (frame_count > 256) frame_count = 256;
	/* the loaders address data/... relative to the working directory, like the reference */
	if (chdir(data_root)) {
		printf("Cannot enter the data root %s.\n", data_root);
		return 1;
	}
	application_t app;
	memset(&app, 0, sizeof(app));
	experiment_list_t list;
	create_experiment_list(&list);
	char* screenshot_format = NULL;
	if (create_hip_device(&app.device, 0, NULL)) {
		destroy_experiment_list(&list);
		return 1;
	}
	app.swapchain.extent.width = 1280;
	app.swapchain.extent.height = 1024;
	if (experiment_index >= 0) {
		if ((uint32_t) experiment_index >= list.count) {
			printf("There are %u experiments, %d is not one of them.\n", list.count, experiment_index);
			return 1;
		}
		const experiment_t* experiment = &list.experiments[experiment_index];
		if (apply_experiment(&app, experiment, ".")) return 1;
		screenshot_format = strdup(experiment->screenshot_path);
		hdr |= (int) experiment->use_hdr;
	}
	else {
		specify_default_scene(&app.scene_specification);
		specify_default_render_settings(&app.render_settings);
		screenshot_format = strdup("data/default_%.3f.png");
	}
	if (white_noise) app.render_settings.noise_type = noise_type_white;
	app.render_settings.show_gui = VK_FALSE;
	scene_specification_t* spec = &app.scene_specification;
	int failed = load_noise_table(&app.noise_table, &app.device, get_default_noise_resolution(app.render_settings.noise_type), app.render_settings.noise_type)
		|| load_ltc_table(&app.ltc_table, &app.device, "data/ggx_ltc_fit", fresnel_count)
		|| load_scene(&app.scene, &app.device, spec->file_path, spec->texture_path, VK_TRUE)
		|| create_and_assign_light_textures(&app.light_textures, &app.device, spec)
		|| create_render_targets(&app.render_targets, &app.device, &app.swapchain)
		|| create_shading_pass(&app.shading_pass, &app)
		|| render_visibility_pass(&app);
	float times[256];
	for (int i = 0; i != 8 && !failed; ++i) failed = render_shading_pass(&app, NULL);
	for (int i = 0; i != frame_count && !failed; ++i) failed = render_shading_pass(&app, NULL);
	if (!failed) {
		failed = wait_for_device(&app.device);
		uint32_t timed = get_dispatch_milliseconds(&app, times, (uint32_t) frame_count);
		qsort(times, timed, sizeof(float), compare_floats);
		float frame_ms = timed ? times[timed / 2] : 0.0f;
		if (hdr && strlen(screenshot_format) > 3) memcpy(screenshot_format + strlen(screenshot_format) - 3, "hdr", 3);
		char* path = format_screenshot_path(screenshot_format, frame_ms);
		failed |= take_screenshot(&app, hdr ? NULL : path, hdr ? path : NULL);
		uint64_t pixels = (uint64_t) app.swapchain.extent.width * app.swapchain.extent.height;
		printf("%ux%u, %u light(s), %u spp, %s shadow rays: %.4f ms per frame (median of %u), %.1f Msamples/s, %llu rays, screenshot %s\n",
			app.swapchain.extent.width, app.swapchain.extent.height, spec->polygonal_light_count, app.render_settings.sample_count,
			app.shading_pass.use_ray_tracing ? "with" : "without", frame_ms, timed,
			frame_ms > 0.0f ? (double) pixels * app.render_settings.sample_count / (frame_ms * 1.0e3) : 0.0,
			(unsigned long long) get_last_ray_count(&app), path);
		free(path);
	}
	free(screenshot_format);
	destroy_shading_pass(&app.shading_pass, &app.device);
	destroy_render_targets(&app.render_targets, &app.device);
	destroy_light_textures(&app.light_textures, &app.device);
	destroy_scene(&app.scene, &app.device);
	destroy_ltc_table(&app.ltc_table, &app.device);
	destroy_noise_table(&app.noise_table, &app.device);
	destroy_scene_specification(&app.scene_specification);
	destroy_experiment_list(&list);
	destroy_hip_device(&app.device);
	return failed ? 1 : 0;
}
