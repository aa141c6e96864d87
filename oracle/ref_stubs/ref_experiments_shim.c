/* TEST INFRASTRUCTURE.  Exposes two more pieces of the reference through
 * oracle/_ref/libref_host.so, built from the sources where they lie:
 *  - create_experiment_list() of src/experiment_list.c (compiled unmodified against
 *    opaque Vulkan type stubs), flattened into plain records, and
 *  - the PNG / Radiance HDR writers of the reference's vendored stb_image_write.h
 *    that implement_screenshot() (src/main.c:1719-1770) calls. */
#include "vulkan_stub/vulkan_types_stub.h"
#include "main.h"
#include <string.h>

#define STB_IMAGE_WRITE_IMPLEMENTATION
#define STBI_WRITE_NO_STDIO_UNUSED
#include "stb_image_write.h"

typedef struct {
	uint32_t width, height, scene_index, use_hdr;
	const char* quick_save_path;
	const char* screenshot_path;
	float exposure_factor, roughness_factor;
	uint32_t sample_count, sampling_strategies, mis_heuristic;
	float mis_visibility_estimate;
	uint32_t polygon_sampling_technique, error_display;
	float error_min_exponent;
	uint32_t noise_type, animate_noise, trace_shadow_rays, show_polygonal_lights, show_gui, v_sync;
} ref_experiment_view_t;

static experiment_list_t g_list;
static int g_list_ready = 0;

uint32_t ref_experiment_count(void) {
	if (!g_list_ready) { create_experiment_list(&g_list); g_list_ready = 1; }
	return g_list.count;
}

int ref_experiment_get(uint32_t index, ref_experiment_view_t* out) {
	if (index >= ref_experiment_count()) return 1;
	const experiment_t* e = &g_list.experiments[index];
	const render_settings_t* s = &e->render_settings;
	out->width = e->width; out->height = e->height; out->scene_index = (uint32_t) e->scene_index; out->use_hdr = e->use_hdr;
	out->quick_save_path = e->quick_save_path; out->screenshot_path = e->screenshot_path;
	out->exposure_factor = s->exposure_factor; out->roughness_factor = s->roughness_factor;
	out->sample_count = s->sample_count; out->sampling_strategies = (uint32_t) s->sampling_strategies;
	out->mis_heuristic = (uint32_t) s->mis_heuristic; out->mis_visibility_estimate = s->mis_visibility_estimate;
	out->polygon_sampling_technique = (uint32_t) s->polygon_sampling_technique; out->error_display = (uint32_t) s->error_display;
	out->error_min_exponent = s->error_min_exponent; out->noise_type = (uint32_t) s->noise_type;
	out->animate_noise = s->animate_noise; out->trace_shadow_rays = s->trace_shadow_rays;
	out->show_polygonal_lights = s->show_polygonal_lights; out->show_gui = s->show_gui; out->v_sync = s->v_sync;
	return 0;
}

int ref_write_png_rgb8(const char* path, int width, int height, const uint8_t* rgb) {
	return stbi_write_png(path, width, height, 3, rgb, width * 3) ? 0 : 1;
}

int ref_write_hdr_rgb32f(const char* path, int width, int height, const float* rgb) {
	return stbi_write_hdr(path, width, height, 3, rgb) ? 0 : 1;
}
