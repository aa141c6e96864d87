/* Experiment table and screenshot writers: the reproducible-measurement harness of
 * the reference (src/experiment_list.c, src/main.c:1601-1770, :1896-1945, :2277-2284)
 * behind the same names.  An experiment = scene + quicksave (camera, lights) + render
 * settings + resolution + the path of the screenshot that records the result; the
 * screenshot path is a format string that consumes the frame time in milliseconds. */
#ifndef VKR_EXPERIMENTS_H
#define VKR_EXPERIMENTS_H
#include "vkr_shading_pass.h"

/*! reference main.h:161-172 */
typedef enum scene_index_e {
	scene_cornell_box,
	scene_mis_plane,
	scene_roughness_planes,
	scene_shadowed_plane,
	scene_arcade,
	scene_living_room,
	scene_attic,
	scene_bistro_inside,
	scene_bistro_outside,
	scene_count
} scene_index_t;

/*! Display name, *.vks path, texture directory, default quicksave per scene
	(reference main.c:34-44) */
VKR_API extern const char* const g_scene_paths[scene_count][4];

/*! reference main.h:184-201, same member order */
typedef struct experiment_s {
	uint32_t width, height;
	scene_index_t scene_index;
	char* quick_save_path;
	VkBool32 use_hdr;
	char* screenshot_path;
	render_settings_t render_settings;
} experiment_t;

/*! reference main.h:205-214 */
typedef enum experiment_state_e {
	experiment_state_rendering,
	experiment_state_screenshot_frame_0,
	experiment_state_screenshot_frame_1,
	experiment_state_new_experiment,
} experiment_state_t;

/*! reference main.h:218-238, same member order */
typedef struct experiment_list_s {
	experiment_t* experiments;
	const experiment_t* experiment;
	uint32_t count;
	uint32_t next;
	double next_setup_time;
	uint32_t next_setup_frame;
	uint32_t frame_index;
	experiment_state_t state;
} experiment_list_t;

/*! reference experiment_list.c:22-544: the 321 experiments of the paper (figures and
	the 260-entry timing matrix), in the same order with the same paths and settings */
VKR_API void create_experiment_list(experiment_list_t* list);
/*! reference experiment_list.c:547-554 */
VKR_API void destroy_experiment_list(experiment_list_t* list);

/*! The experiment branch of startup_application (reference main.c:1909-1925): replaces
	the scene specification (paths from g_scene_paths, prefixed with data_root + "/" when
	data_root is not NULL; quicksave of the experiment or the scene's default, loaded
	with quick_load when the file exists), the render settings and the swapchain extent.
	Returns 0 on success.  Scene, tables, render targets and shading pass have to be
	(re)created by the caller afterwards, as in the reference's update_application. */
VKR_API int apply_experiment(application_t* app, const experiment_t* experiment, const char* data_root);

/*! reference math_utilities.h:70-84 */
VKR_API float half_to_float(uint16_t half);
/*! Formats a screenshot path of an experiment with the frame time in milliseconds
	(reference string_utilities.h:77-82, main.c:2006).  malloc'ed. */
VKR_API char* format_screenshot_path(const char* format_string, float frame_time_milliseconds);

/*! 8-bit RGB PNG (reference: stbi_write_png at main.c:1731).  Returns 0 on success. */
VKR_API int write_png_rgb8(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb);
/*! Radiance RGBE *.hdr with run-length encoded scanlines from linear float RGB
	(reference: stbi_write_hdr at main.c:1752).  Returns 0 on success. */
VKR_API int write_hdr_rgb32f(const char* path, uint32_t width, uint32_t height, const float* rgb);

/*! implement_screenshot (reference main.c:1719-1770) for an already rendered radiance
	target: path_png gets the sRGB-encoded LDR frame (alpha dropped); path_hdr gets the
	frame that the reference assembles from two renders with frame_bits 1 and 2, i.e.
	every channel rounded to half precision and widened again (main.c:1700-1711).
	Either path may be NULL.  Returns 0 on success; prints the reason otherwise. */
VKR_API int take_screenshot(application_t* app, const char* path_png, const char* path_hdr);

#endif
