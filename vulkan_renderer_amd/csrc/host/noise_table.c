/* Noise tables: reference src/noise_table.c:23-168. */
#include "vkr_internal.h"

uint32_t vkr_wang_random_number(uint32_t seed) {
	seed = (seed ^ 61u) ^ (seed >> 16);
	seed *= 9u;
	seed ^= seed >> 4;
	seed *= 0x27d4eb2du;
	seed ^= seed >> 15;
	return seed;
}

VkExtent3D get_default_noise_resolution(noise_type_t noise_type) {
	VkExtent3D r = {256, 256, 64};
	if (noise_type == noise_type_blue) { r.width = r.height = r.depth = 64; }
	else if (noise_type == noise_type_blue_noise_dithered) { r.width = r.height = 128; r.depth = 1; }
	return r;
}

int load_noise_table(noise_table_t* noise, const device_t* device, VkExtent3D resolution, noise_type_t noise_type) {
	memset(noise, 0, sizeof(*noise));
	noise->random_seed = 3124705;
	if (resolution.width > 9999 || resolution.height > 9999 || resolution.depth > 9999
		|| resolution.width == 0 || resolution.height == 0 || resolution.depth == 0) {
		printf("Invalid noise resolution or slice count.\n");
		return 1;
	}
	noise->resolution = resolution;
	uint32_t cell_count = resolution.width * resolution.height * resolution.depth * 4;
	noise->host_data = (uint16_t*) malloc(sizeof(uint16_t) * cell_count);
	if (noise_type == noise_type_white) {
		for (uint32_t i = 0; i != cell_count; ++i)
			noise->host_data[i] = (uint16_t) (vkr_wang_random_number(i + 243708) & 0xFFFF);
	}
	else {
		const char* stem = NULL;
		switch (noise_type) {
		case noise_type_blue: stem = "data/noise/blue_noise_rgba_%02dx%02d_%02d.blob"; break;
		case noise_type_sobol: stem = "data/noise/sobol_2d_rgba_%02dx%02d_%02d.blob"; break;
		case noise_type_owen: stem = "data/noise/owen_2d_rgba_%02dx%02d_%02d.blob"; break;
		case noise_type_burley_owen: stem = "data/noise/burley_owen_2d_rgba_%02dx%02d_%02d.blob"; break;
		case noise_type_ahmed: stem = "data/noise/ahmed_2d_rgba_%02dx%02d_%02d.blob"; break;
		case noise_type_blue_noise_dithered: stem = "data/noise/dithered_2d_rgba_%02dx%02d_%02d.blob"; break;
		default: break;
		}
		if (!stem) {
			printf("Failed to load a noise table. The given type is unknown.\n");
			destroy_noise_table(noise, device);
			return 1;
		}
		char path[256];
		snprintf(path, sizeof(path), stem, resolution.width, resolution.height, resolution.depth);
		FILE* file = fopen(path, "rb");
		if (!file) {
			printf("Failed to open the noise file at path %s. Please check path and permissions?\n", path);
			destroy_noise_table(noise, device);
			return 1;
		}
		size_t got = fread(noise->host_data, sizeof(uint16_t), cell_count, file);
		fclose(file);
		if (got != cell_count) {
			printf("The noise file at path %s is too short for resolution %ux%ux%u.\n", path, resolution.width, resolution.height, resolution.depth);
			destroy_noise_table(noise, device);
			return 1;
		}
	}
	if (device && vkr_device_upload(&noise->device_data, device, noise->host_data, sizeof(uint16_t) * cell_count, "the noise table")) {
		destroy_noise_table(noise, device);
		return 1;
	}
	return 0;
}

void destroy_noise_table(noise_table_t* noise, const device_t* device) {
	free(noise->host_data);
	vkr_device_free(noise->device_data, device);
	memset(noise, 0, sizeof(*noise));
}

void set_noise_constants(uint32_t resolution_mask[2], uint32_t* texture_index_mask, uint32_t random_numbers[4], noise_table_t* noise, VkBool32 animate_noise) {
	resolution_mask[0] = noise->resolution.width - 1;
	resolution_mask[1] = noise->resolution.height - 1;
	(*texture_index_mask) = noise->resolution.depth - 1;
	for (uint32_t i = 0; i != 4; ++i)
		random_numbers[i] = animate_noise ? vkr_wang_random_number(noise->random_seed * 4 + i) : (i * 0x123456);
	if (animate_noise) ++noise->random_seed;
}
