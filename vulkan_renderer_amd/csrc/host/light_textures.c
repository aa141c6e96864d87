/* Light textures (reference create_and_assign_light_textures / destroy_light_textures,
 * main.c:364-417): the textures that polygonal_light_t.texture_file_path names, loaded once each
 * and indexed by polygonal_light_t.texture_index.
 *
 * The shader reads them with textureLod(..., 0.0f) (shading_pass.frag.glsl:182), so only the
 * finest level is kept, as RGBA fp32 whatever the file holds (the texture conversion tool writes
 * half / float formats for light probes and IES profiles, tools/texture_conversion/main.c:31-39;
 * the 8-bit and block formats of material textures are accepted too).  The reference falls back to
 * data/white.vkt for lights without a texture; here white is built in (a descriptor of width 0)
 * and that file is never opened. */
#include "vkr_internal.h"
#include "vkr_experiments.h"

enum {
	format_r16g16b16_sfloat = 90, format_r16g16b16a16_sfloat = 97,
	format_r32g32b32_sfloat = 106, format_r32g32b32a32_sfloat = 109,
};

static const char* const g_default_path = "data/white.vkt";

/* Finest level of a *.vkt as RGBA fp32.  0 on success, 1 absent, 2 invalid / unsupported. */
static int load_level_0_rgba32f(float** out_texels, uint32_t* out_width, uint32_t* out_height, const char* path) {
	(*out_texels) = NULL;
	FILE* file = fopen(path, "rb");
	if (!file) return 1;
	int32_t header[6];
	uint64_t payload_size = 0;
	int ok = fread(header, sizeof(int32_t), 6, file) == 6 && fread(&payload_size, sizeof(uint64_t), 1, file) == 1
		&& header[0] == 0xbc1bc1 && header[1] == 1 && header[2] >= 1 && header[2] <= 32 && header[3] >= 1 && header[4] >= 1;
	int32_t format = ok ? header[5] : 0;
	uint32_t channels = (format == format_r16g16b16_sfloat || format == format_r32g32b32_sfloat) ? 3 : 4;
	uint32_t channel_bytes = (format == format_r16g16b16_sfloat || format == format_r16g16b16a16_sfloat) ? 2 : 4;
	int is_float = format == format_r16g16b16_sfloat || format == format_r16g16b16a16_sfloat || format == format_r32g32b32_sfloat || format == format_r32g32b32a32_sfloat;
	if (ok && !is_float) {
		/* the 8-bit and block-compressed formats go through the material texture decoder */
		fclose(file);
		vkr_host_texture_t texture;
		int result = vkr_load_texture_rgba8(&texture, path);
		if (result) {
			if (result == 3) printf("The light texture at path %s has format %d, which is not supported.\n", path, format);
			return 2;
		}
		float srgb_table[256];
		vkr_fill_srgb_table(srgb_table);
		uint64_t count = (uint64_t) texture.width * texture.height;
		float* texels = (float*) malloc(count * 4 * sizeof(float));
		if (!texels) {
			vkr_free_host_texture(&texture);
			return 2;
		}
		for (uint64_t i = 0; i != count; ++i) {
			for (uint32_t c = 0; c != 3; ++c)
				texels[4 * i + c] = texture.srgb ? srgb_table[texture.texels[4 * i + c]] : (float) texture.texels[4 * i + c] / 255.0f;
			texels[4 * i + 3] = (float) texture.texels[4 * i + 3] / 255.0f;
		}
		(*out_texels) = texels; (*out_width) = texture.width; (*out_height) = texture.height;
		vkr_free_host_texture(&texture);
		return 0;
	}
	int32_t extent[2] = {0, 0};
	uint64_t size_offset[2] = {0, 0};
	ok = ok && fread(extent, sizeof(int32_t), 2, file) == 2 && fread(size_offset, sizeof(uint64_t), 2, file) == 2
		&& extent[0] == header[3] && extent[1] == header[4];
	uint64_t count = ok ? (uint64_t) extent[0] * (uint64_t) extent[1] : 0;
	uint64_t needed = count * channels * channel_bytes;
	ok = ok && size_offset[0] >= needed && size_offset[1] + needed <= payload_size
		&& fseek(file, (long) (32 + 24 * (uint64_t) header[2] + size_offset[1]), SEEK_SET) == 0;
	uint8_t* raw = ok ? (uint8_t*) malloc(needed ? needed : 1) : NULL;
	ok = ok && raw && fread(raw, 1, needed, file) == needed;
	/* the end of file marker follows the payload */
	uint32_t eof_marker = 0;
	ok = ok && fseek(file, (long) (32 + 24 * (uint64_t) header[2] + payload_size), SEEK_SET) == 0
		&& fread(&eof_marker, sizeof(eof_marker), 1, file) == 1 && eof_marker == 0xE0FE0F;
	fclose(file);
	if (!ok) {
		printf("The light texture at path %s is not a valid *.vkt file.\n", path);
		free(raw);
		return 2;
	}
	float* texels = (float*) malloc(count * 4 * sizeof(float));
	if (!texels) {
		free(raw);
		return 2;
	}
	for (uint64_t i = 0; i != count; ++i) {
		for (uint32_t c = 0; c != 4; ++c) {
			float value = 1.0f;
			if (c < channels) {
				const uint8_t* source = raw + (i * channels + c) * channel_bytes;
				if (channel_bytes == 2) {
					uint16_t half;
					memcpy(&half, source, 2);
					value = half_to_float(half);
				}
				else memcpy(&value, source, 4);
			}
			texels[4 * i + c] = value;
		}
	}
	free(raw);
	(*out_texels) = texels; (*out_width) = (uint32_t) extent[0]; (*out_height) = (uint32_t) extent[1];
	return 0;
}

void destroy_light_textures(light_textures_t* light_textures, const device_t* device) {
	vkr_device_free(light_textures->descriptors, device);
	vkr_device_free(light_textures->texels, device);
	free(light_textures->host_descriptors);
	free(light_textures->host_texels);
	memset(light_textures, 0, sizeof(*light_textures));
}

int create_and_assign_light_textures(light_textures_t* light_textures, const device_t* device, scene_specification_t* scene_specification) {
	/* the list of distinct paths; absent files fall back to white like in the reference */
	uint32_t light_count = scene_specification->polygonal_light_count;
	const char** unique_paths = (const char**) malloc(sizeof(char*) * (light_count + 1));
	if (!unique_paths) return 1;
	uint32_t unique_count = 0;
	for (uint32_t i = 0; i != light_count; ++i) {
		polygonal_light_t* light = &scene_specification->polygonal_lights[i];
		const char* new_path = light->texture_file_path;
		if (!new_path || strlen(new_path) == 0) new_path = g_default_path;
		else {
			FILE* file = fopen(new_path, "rb");
			if (file) fclose(file);
			else {
				printf("The light texture at path %s does not exist. Using a white texture instead.\n", new_path);
				new_path = g_default_path;
			}
		}
		light->texture_index = unique_count;
		for (uint32_t j = 0; j != unique_count; ++j)
			if (strcmp(new_path, unique_paths[j]) == 0) light->texture_index = j;
		if (light->texture_index == unique_count) unique_paths[unique_count++] = new_path;
	}
	if (!light_textures) {
		free(unique_paths);
		return 0;
	}
	memset(light_textures, 0, sizeof(*light_textures));
	if (unique_count == 0) unique_paths[unique_count++] = g_default_path;
	light_textures->texture_count = unique_count;
	light_textures->host_descriptors = (uint32_t(*)[4]) calloc(unique_count, sizeof(uint32_t[4]));
	if (!light_textures->host_descriptors) {
		free(unique_paths);
		destroy_light_textures(light_textures, device);
		return 1;
	}
	for (uint32_t i = 0; i != unique_count; ++i) {
		if (unique_paths[i] == g_default_path) continue;
		float* texels;
		uint32_t width, height;
		if (load_level_0_rgba32f(&texels, &width, &height, unique_paths[i])) {
			printf("Failed to load the light texture at path %s.\n", unique_paths[i]);
			free(unique_paths);
			destroy_light_textures(light_textures, device);
			return 1;
		}
		uint64_t count = (uint64_t) width * height;
		float* grown = (float*) realloc(light_textures->host_texels, (light_textures->texel_count + count) * 4 * sizeof(float));
		if (!grown) {
			printf("Out of memory for the light texture at path %s.\n", unique_paths[i]);
			free(texels);
			free(unique_paths);
			destroy_light_textures(light_textures, device);
			return 1;
		}
		light_textures->host_texels = grown;
		memcpy(light_textures->host_texels + 4 * light_textures->texel_count, texels, count * 4 * sizeof(float));
		free(texels);
		light_textures->host_descriptors[i][0] = (uint32_t) light_textures->texel_count;
		light_textures->host_descriptors[i][1] = width;
		light_textures->host_descriptors[i][2] = height;
		light_textures->texel_count += count;
	}
	free(unique_paths);
	if (light_textures->texel_count >> 32) {
		printf("The light textures hold more than 2^32 texels.\n");
		destroy_light_textures(light_textures, device);
		return 1;
	}
	if (vkr_device_upload((void**) &light_textures->descriptors, device, light_textures->host_descriptors, sizeof(uint32_t[4]) * unique_count, "light texture descriptors")
		|| (light_textures->texel_count && vkr_device_upload((void**) &light_textures->texels, device, light_textures->host_texels, light_textures->texel_count * 4 * sizeof(float), "light texture texels")))
	{
		destroy_light_textures(light_textures, device);
		return 1;
	}
	return 0;
}
