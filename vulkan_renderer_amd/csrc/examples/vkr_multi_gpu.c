/* The shading pass of one experiment of the reference's table tiled over the GPUs of one node:
 * plain C99 on the C-ABI of libvkr_shading.so (include/vkr_slab_exchange.h), one thread per GPU,
 * RCCL all-gather of the tile slabs over xGMI, no Python, no MPI.
 *
 *     vkr_multi_gpu --gpus 8 -e34 [--frames 64] [--format rgb8] [--tile 32] [--white-noise] [--fresnel 51] [data root]
 *
 * Every rank loads the scene, builds its BVH and renders the visibility buffer (replicated
 * inputs: a few MB), then shades tiles t with t % N == rank, gathers all slabs and assembles
 * the frame; rank 0 compares the assembled frame with a single-GPU render of the whole frame
 * bit for bit, prints the timing and stores the screenshot.  A process-per-GPU launcher (MPI,
 * torchrun) does the same with get_slab_exchange_id() on rank 0 and a broadcast of the token. */
#define _GNU_SOURCE
#include "vkr_experiments.h"
#include "vkr_slab_exchange.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef struct options_s {
	int gpu_count, experiment_index, frame_count, white_noise;
	uint32_t fresnel_count, tile_size;
	slab_format_t format;
} options_t;

typedef struct rank_state_s {
	const options_t* options;
	const experiment_list_t* list;
	const slab_exchange_id_t* id;
	pthread_barrier_t* barrier;
	int rank, failed;
	double seconds;
	float stage_ms[3];
} rank_state_t;

static double now_seconds(void) {
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1.0e-9 * (double) t.tv_nsec;
}

static void* run_rank(void* argument) {
	rank_state_t* state = (rank_state_t*) argument;
	const options_t* options = state->options;
	application_t app;
	slab_exchange_t exchange;
	memset(&app, 0, sizeof(app));
	memset(&exchange, 0, sizeof(exchange));
	int failed = create_hip_device(&app.device, state->rank, NULL);
	if (!failed) {
		if (options->experiment_index >= 0) failed = apply_experiment(&app, &state->list->experiments[options->experiment_index], ".");
		else {
			specify_default_scene(&app.scene_specification);
			specify_default_render_settings(&app.render_settings);
			app.swapchain.extent.width = 1920;
			app.swapchain.extent.height = 1080;
		}
	}
	if (options->white_noise) app.render_settings.noise_type = noise_type_white;
	app.shading_pass.frames_in_flight = 2;
	scene_specification_t* spec = &app.scene_specification;
	failed = failed
		|| load_noise_table(&app.noise_table, &app.device, get_default_noise_resolution(app.render_settings.noise_type), app.render_settings.noise_type)
		|| load_ltc_table(&app.ltc_table, &app.device, "data/ggx_ltc_fit", options->fresnel_count)
		|| load_scene(&app.scene, &app.device, spec->file_path, spec->texture_path, VK_TRUE)
		|| create_and_assign_light_textures(&app.light_textures, &app.device, spec)
		|| create_render_targets(&app.render_targets, &app.device, &app.swapchain)
		|| create_shading_pass(&app.shading_pass, &app)
		|| render_visibility_pass(&app);
	/* single-GPU reference frame on rank 0, before the schedule is set */
	size_t pixels = (size_t) app.swapchain.extent.width * app.swapchain.extent.height;
	float* single = NULL;
	if (!failed && state->rank == 0) {
		single = (float*) malloc(sizeof(float) * 4 * pixels);
		failed = !single || render_shading_pass(&app, NULL) || read_back_radiance(&app, single);
	}
	app.tile_schedule.tile_size = options->tile_size;
	app.tile_schedule.rank = (uint32_t) state->rank;
	app.tile_schedule.rank_count = (uint32_t) options->gpu_count;
	app.tile_schedule.slab_layout = VK_TRUE;
	/* every thread arrives here, failed or not: the communicator needs all ranks */
	state->failed = failed;
	pthread_barrier_wait(state->barrier);
	int any_failed = 0;
	for (int r = 0; r != options->gpu_count; ++r) any_failed |= state[r - state->rank].failed;
	if (!any_failed) failed = create_slab_exchange(&exchange, &app, state->id, options->format);
	state->failed = failed;
	pthread_barrier_wait(state->barrier);
	for (int r = 0; r != options->gpu_count; ++r) any_failed |= state[r - state->rank].failed;
	if (!any_failed) {
		for (int i = 0; i != 8 && !failed; ++i) failed = render_and_exchange_frame(&app, &exchange, NULL);
		failed = failed || finish_slab_exchange(&app, &exchange) || wait_for_device(&app.device);
		pthread_barrier_wait(state->barrier);
		double start = now_seconds();
		for (int i = 0; i != options->frame_count && !failed; ++i) failed = render_and_exchange_frame(&app, &exchange, NULL);
		failed = failed || finish_slab_exchange(&app, &exchange) || wait_for_device(&app.device);
		state->seconds = now_seconds() - start;
		(void) get_slab_exchange_milliseconds(&exchange, state->stage_ms);
		pthread_barrier_wait(state->barrier);
		if (!failed && state->rank == 0) {
			/* every rank holds the whole frame now; rank 0 checks and stores it */
			size_t different = 0;
			if (options->format == slab_format_rgba32f) {
				float* assembled = (float*) malloc(sizeof(float) * 4 * pixels);
				failed = !assembled || read_back_radiance(&app, assembled);
				for (size_t i = 0; i != 4 * pixels && !failed; ++i) different += memcmp(&assembled[i], &single[i], sizeof(float)) != 0;
				free(assembled);
			}
			else {
				uint8_t* assembled = (uint8_t*) malloc(4 * pixels);
				uint8_t* expected = (uint8_t*) malloc(4 * pixels);
				/* (read the assembled codes first: encode_output() overwrites render_targets.encoded) */
				failed = !assembled || !expected || read_back_encoded(&app, assembled);
				app.tile_schedule.rank_count = 1; app.tile_schedule.slab_layout = VK_FALSE;
				failed = failed || render_shading_pass(&app, NULL) || encode_output(&app, VK_FALSE) || read_back_encoded(&app, expected);
				for (size_t i = 0; i != 4 * pixels && !failed; ++i) different += assembled[i] != expected[i];
				free(assembled); free(expected);
			}
			double seconds = 0.0;
			for (int r = 0; r != options->gpu_count; ++r) seconds = state[r].seconds > seconds ? state[r].seconds : seconds;
			double frame_ms = 1.0e3 * seconds / options->frame_count;
			printf("%d GPU(s), %ux%u, %u light(s), %u spp, %s slabs: %.4f ms per frame (max over ranks, %d frames), %.1f Msamples/s; rank 0 stages: shade %.3f ms, all-gather %.3f ms, scatter %.3f ms; %llu values differ from the single-GPU frame\n",
				options->gpu_count, app.swapchain.extent.width, app.swapchain.extent.height, spec->polygonal_light_count, app.render_settings.sample_count,
				options->format == slab_format_rgba32f ? "RGBA32F" : "RGB8", frame_ms, options->frame_count,
				(double) pixels * app.render_settings.sample_count / (frame_ms * 1.0e3), state->stage_ms[0], state->stage_ms[1], state->stage_ms[2], (unsigned long long) different);
			if (different) failed = 1;
			if (!failed && options->format == slab_format_rgba32f) failed = take_screenshot(&app, "data/multi_gpu.png", NULL);
		}
	}
	free(single);
	destroy_slab_exchange(&exchange, &app);
	destroy_shading_pass(&app.shading_pass, &app.device);
	destroy_render_targets(&app.render_targets, &app.device);
	destroy_light_textures(&app.light_textures, &app.device);
	destroy_scene(&app.scene, &app.device);
	destroy_ltc_table(&app.ltc_table, &app.device);
	destroy_noise_table(&app.noise_table, &app.device);
	destroy_scene_specification(&app.scene_specification);
	destroy_hip_device(&app.device);
	state->failed = failed || any_failed;
	return NULL;
}

int main(int argc, char** argv) {
	options_t options = {1, -1, 64, 0, 51, 32, slab_format_rgba32f};
	const char* data_root = ".";
	for (int i = 1; i < argc; ++i) {
		if (strncmp(argv[i], "-e", 2) == 0 && argv[i][2]) options.experiment_index = atoi(argv[i] + 2);
		else if (strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) options.gpu_count = atoi(argv[++i]);
		else if (strcmp(argv[i], "--frames") == 0 && i + 1 < argc) options.frame_count = atoi(argv[++i]);
		else if (strcmp(argv[i], "--fresnel") == 0 && i + 1 < argc) options.fresnel_count = (uint32_t) atoi(argv[++i]);
		else if (strcmp(argv[i], "--tile") == 0 && i + 1 < argc) options.tile_size = (uint32_t) atoi(argv[++i]);
		else if (strcmp(argv[i], "--format") == 0 && i + 1 < argc) options.format = strcmp(argv[++i], "rgb8") == 0 ? slab_format_rgb8 : slab_format_rgba32f;
		else if (strcmp(argv[i], "--white-noise") == 0) options.white_noise = 1;
		else if (strcmp(argv[i], "--help") == 0 || strcmp(argv[i], "-h") == 0) {
			printf("usage: %s --gpus n [-e<experiment index>] [--frames n] [--format rgba32f|rgb8] [--tile n] [--white-noise] [--fresnel n] [data root]\n", argv[0]);
			return 0;
		}
		else data_root = argv[i];
	}
	if (options.gpu_count < 1 || options.gpu_count > 64 || options.frame_count < 1) {
		printf("Need 1 to 64 GPUs and at least one frame.\n");
		return 1;
	}
	if (chdir(data_root)) {
		printf("Cannot enter the data root %s.\n", data_root);
		return 1;
	}
	experiment_list_t list;
	create_experiment_list(&list);
	if (options.experiment_index >= 0 && (uint32_t) options.experiment_index >= list.count) {
		printf("There are %u experiments, %d is not one of them.\n", list.count, options.experiment_index);
		return 1;
	}
	slab_exchange_id_t id;
	if (get_slab_exchange_id(&id)) return 1;
	pthread_barrier_t barrier;
	pthread_barrier_init(&barrier, NULL, (unsigned) options.gpu_count);
	rank_state_t* states = (rank_state_t*) calloc((size_t) options.gpu_count, sizeof(rank_state_t));
	pthread_t* threads = (pthread_t*) calloc((size_t) options.gpu_count, sizeof(pthread_t));
	for (int r = 0; r != options.gpu_count; ++r) {
		states[r].options = &options; states[r].list = &list; states[r].id = &id; states[r].barrier = &barrier; states[r].rank = r;
		pthread_create(&threads[r], NULL, run_rank, &states[r]);
	}
	int failed = 0;
	for (int r = 0; r != options.gpu_count; ++r) {
		pthread_join(threads[r], NULL);
		failed |= states[r].failed;
	}
	pthread_barrier_destroy(&barrier);
	free(states); free(threads);
	destroy_experiment_list(&list);
	return failed ? 1 : 0;
}
