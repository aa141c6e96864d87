/* Scene loader: .vks parser of reference src/scene.c:409-559 with device
 * buffers instead of Vulkan objects, constant material texels instead of
 * filtered textures and an LBVH instead of the driver's acceleration structure. */
#include "vkr_internal.h"

const char* get_material_texture_suffix(material_texture_type_t type) {
	/* reference scene.c:24-31 */
	switch (type) {
	case material_texture_type_base_color: return "BaseColor";
	case material_texture_type_specular: return "Specular";
	case material_texture_type_normal: return "Normal";
	default: return NULL;
	}
}

static float half_bits_to_float(uint16_t h) {
	uint32_t sign = (uint32_t) (h & 0x8000) << 16, exponent = (h >> 10) & 0x1F, mantissa = h & 0x3FF;
	uint32_t bits;
	if (exponent == 0) {
		if (mantissa == 0) bits = sign;
		else {
			int shift = 0;
			while (!(mantissa & 0x400)) { mantissa <<= 1; ++shift; }
			bits = sign | ((uint32_t) (113 - shift) << 23) | ((mantissa & 0x3FF) << 13);
		}
	}
	else if (exponent == 31) bits = sign | 0x7F800000u | (mantissa << 13);
	else bits = sign | ((exponent + 112) << 23) | (mantissa << 13);
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

/* Reads the first texel of the smallest mip level of a .vkt file (format:
   reference textures.c:101-172, writer tools/texture_conversion/main.c:41-64).
   Returns 0 and fills up to four channels on success, 1 if the file is absent,
   2 if it exists but cannot be reduced to a constant here. */
static int read_constant_texel(float out[4], const char* path) {
	FILE* file = fopen(path, "rb");
	if (!file) return 1;
	int32_t header[6];
	uint64_t payload_size;
	if (fread(header, sizeof(int32_t), 6, file) != 6 || fread(&payload_size, sizeof(uint64_t), 1, file) != 1
		|| header[0] != 0xbc1bc1 || header[1] != 1 || header[2] < 1 || header[2] > 32) {
		printf("The texture at path %s does not have the .vkt format.\n", path);
		fclose(file);
		return 2;
	}
	int32_t mip_count = header[2], format = header[5];
	uint64_t last_size = 0, last_offset = 0;
	for (int32_t m = 0; m != mip_count; ++m) {
		int32_t extent[2];
		uint64_t size_offset[2];
		if (fread(extent, sizeof(int32_t), 2, file) != 2 || fread(size_offset, sizeof(uint64_t), 2, file) != 2) { fclose(file); return 2; }
		last_size = size_offset[0];
		last_offset = size_offset[1];
	}
	/* VkFormat numbers: 90/97 half RGB/RGBA, 106/109 float RGB/RGBA */
	uint32_t channels = (format == 90 || format == 106) ? 3 : ((format == 97 || format == 109) ? 4 : 0);
	int is_half = format == 90 || format == 97;
	if (!channels || last_size < channels * (is_half ? 2u : 4u)) {
		printf("The texture at path %s uses VkFormat %d. Only uncompressed half/float textures can be reduced to material constants.\n", path, format);
		fclose(file);
		return 2;
	}
	long payload_start = ftell(file);
	fseek(file, payload_start + (long) last_offset, SEEK_SET);
	out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f;
	int ok = 1;
	for (uint32_t c = 0; c != channels; ++c) {
		if (is_half) { uint16_t h; ok &= fread(&h, 2, 1, file) == 1; out[c] = half_bits_to_float(h); }
		else ok &= fread(&out[c], 4, 1, file) == 1;
	}
	fseek(file, payload_start + (long) payload_size, SEEK_SET);
	uint32_t eof_marker = 0;
	ok &= fread(&eof_marker, sizeof(eof_marker), 1, file) == 1 && eof_marker == 0xE0FE0F;
	fclose(file);
	if (!ok) {
		printf("The texture file at path %s seems to be invalid. The texture data is not followed by the expected end of file marker.\n", path);
		return 2;
	}
	return 0;
}

/* The sRGB decoding table of the software sampler: the same formula and the same libm as the
   oracle's oracle_srgb_table(), so that both sides filter identical floats. */
void vkr_fill_srgb_table(float table[256]) {
	for (int i = 0; i != 256; ++i) {
		float v = (float) i / 255.0f;
		table[i] = (v <= 0.04045f) ? (v / 12.92f) : powf((v + 0.055f) / 1.055f, 2.4f);
	}
}

static int vkr_upload_srgb_table(void** out, const device_t* device) {
	float table[256];
	vkr_fill_srgb_table(table);
	return vkr_device_upload(out, device, table, sizeof(table), "the sRGB table");
}

static void free_mesh(mesh_t* mesh, const device_t* device) {
	free(mesh->host_positions);
	free(mesh->host_normals_and_tex_coords);
	free(mesh->host_material_indices);
	vkr_device_free(mesh->positions, device);
	vkr_device_free(mesh->normals_and_tex_coords, device);
	vkr_device_free(mesh->material_indices, device);
	memset(mesh, 0, sizeof(*mesh));
}

void destroy_scene(scene_t* scene, const device_t* device) {
	free_mesh(&scene->mesh, device);
	free(scene->materials.host_texture_descriptors);
	free(scene->materials.host_texels);
	vkr_device_free(scene->materials.texture_descriptors, device);
	vkr_device_free(scene->materials.texels, device);
	vkr_device_free(scene->materials.srgb_table, device);
	if (scene->materials.material_names)
		for (uint64_t i = 0; i != scene->materials.material_count; ++i) free(scene->materials.material_names[i]);
	free(scene->materials.material_names);
	free(scene->materials.host_constants);
	vkr_device_free(scene->materials.constants, device);
	vkr_destroy_acceleration_structure(&scene->acceleration_structure, device);
	memset(scene, 0, sizeof(*scene));
}

int load_scene(scene_t* scene, const device_t* device, const char* file_path, const char* texture_path, VkBool32 request_acceleration_structure) {
	memset(scene, 0, sizeof(*scene));
	FILE* file = fopen(file_path, "rb");
	if (!file) {
		printf("Failed to open the scene file at %s.\n", file_path);
		return 1;
	}
	uint32_t marker = 0, version = 0;
	int header_ok = fread(&marker, sizeof(marker), 1, file) == 1 && fread(&version, sizeof(version), 1, file) == 1;
	if (!header_ok || marker != 0xabcabc || version != 1) {
		printf("The scene file at path %s is invalid or unsupported. The format marker is 0x%x, the version is %d.\n", file_path, marker, version);
		fclose(file);
		return 1;
	}
	mesh_t* mesh = &scene->mesh;
	header_ok = fread(&scene->materials.material_count, sizeof(uint64_t), 1, file) == 1
		&& fread(&mesh->triangle_count, sizeof(uint64_t), 1, file) == 1
		&& fread(mesh->dequantization_factor, sizeof(float), 3, file) == 3
		&& fread(mesh->dequantization_summand, sizeof(float), 3, file) == 3;
	if (!header_ok || mesh->triangle_count == 0 || mesh->triangle_count > 0x50000000ull || scene->materials.material_count > 256) {
		if (header_ok && mesh->triangle_count == 0)
			printf("The scene file at path %s is completely empty, i.e. it holds 0 triangles.\n", file_path);
		else
			printf("The scene file at path %s has a damaged header.\n", file_path);
		fclose(file);
		destroy_scene(scene, device);
		return 1;
	}
	printf("Triangle count: %llu\n", (unsigned long long) mesh->triangle_count);
	uint64_t material_count = scene->materials.material_count;
	scene->materials.material_names = (char**) calloc(material_count ? material_count : 1, sizeof(char*));
	if (!scene->materials.material_names) { header_ok = 0; material_count = 0; }
	for (uint64_t i = 0; i != material_count; ++i) {
		uint64_t length = 0;
		if (fread(&length, sizeof(length), 1, file) != 1 || length > 4096) { header_ok = 0; break; }
		scene->materials.material_names[i] = (char*) calloc(length + 1, 1);
		if (!scene->materials.material_names[i]) { header_ok = 0; break; }
		if (fread(scene->materials.material_names[i], 1, length + 1, file) != length + 1) { header_ok = 0; break; }
		scene->materials.material_names[i][length] = 0;
	}
	/* the three mesh buffers follow in the layout the kernels consume */
	size_t vertex_count = (size_t) mesh->triangle_count * 3;
	mesh->host_positions = (uint32_t*) malloc(sizeof(uint32_t) * 2 * vertex_count);
	mesh->host_normals_and_tex_coords = (uint16_t*) malloc(sizeof(uint16_t) * 4 * vertex_count);
	mesh->host_material_indices = (uint8_t*) malloc(mesh->triangle_count);
	uint32_t eof_marker = 0;
	/* (sizes come from the file: an allocation may fail) */
	header_ok = header_ok && mesh->host_positions && mesh->host_normals_and_tex_coords && mesh->host_material_indices
		&& fread(mesh->host_positions, sizeof(uint32_t) * 2, vertex_count, file) == vertex_count
		&& fread(mesh->host_normals_and_tex_coords, sizeof(uint16_t) * 4, vertex_count, file) == vertex_count
		&& fread(mesh->host_material_indices, 1, mesh->triangle_count, file) == mesh->triangle_count
		&& fread(&eof_marker, sizeof(eof_marker), 1, file) == 1;
	fclose(file);
	if (!header_ok || eof_marker != 0xE0FE0F) {
		printf("The scene file at path %s seems to be invalid. The geometry data is not followed by the expected end of file marker.\n", file_path);
		destroy_scene(scene, device);
		return 1;
	}
	for (uint64_t t = 0; t != mesh->triangle_count; ++t)
		if (mesh->host_material_indices[t] >= material_count) {
			printf("The scene file at path %s references material %u but only has %llu materials.\n", file_path, mesh->host_material_indices[t], (unsigned long long) material_count);
			destroy_scene(scene, device);
			return 1;
		}
	/* material constants */
	scene->materials.host_constants = (float*) malloc(sizeof(float) * 8 * (material_count ? material_count : 1));
	scene->materials.host_texture_descriptors = (uint32_t*) calloc(12 * (material_count ? material_count : 1), sizeof(uint32_t));
	if (!scene->materials.host_constants || !scene->materials.host_texture_descriptors) {
		printf("Out of memory for the materials of the scene file at path %s.\n", file_path);
		destroy_scene(scene, device);
		return 1;
	}
	for (uint64_t i = 0; i != material_count; ++i) {
		float* k = scene->materials.host_constants + 8 * i;
		const float defaults[8] = {0.8f, 0.8f, 0.8f, 1.0f, 0.5f, 0.0f, 0.5f, 0.5f};
		memcpy(k, defaults, sizeof(defaults));
		for (uint32_t type = 0; type != material_texture_count && texture_path; ++type) {
			const char* pieces[] = {texture_path, "/", scene->materials.material_names[i], "_", get_material_texture_suffix((material_texture_type_t) type), ".vkt"};
			char* path = vkr_concatenate(VKR_COUNT_OF(pieces), pieces);
			/* 8-bit and block-compressed textures become RGBA8 mip chains, half / float ones a constant */
			vkr_host_texture_t image;
			int status = vkr_load_texture_rgba8(&image, path);
			if (status == 0) {
				uint32_t* descriptor = scene->materials.host_texture_descriptors + 4 * (3 * i + type);
				uint8_t* grown = (uint8_t*) realloc(scene->materials.host_texels, 4 * (scene->materials.texel_count + image.texel_count));
				if (!grown || scene->materials.texel_count + image.texel_count > 0xFFFFFFFFull) {
					printf("Out of memory for the material textures of the scene file at path %s.\n", file_path);
					free(path); vkr_free_host_texture(&image);
					destroy_scene(scene, device);
					return 1;
				}
				scene->materials.host_texels = grown;
				memcpy(grown + 4 * scene->materials.texel_count, image.texels, 4 * image.texel_count);
				descriptor[0] = (uint32_t) scene->materials.texel_count;
				descriptor[1] = image.width; descriptor[2] = image.height;
				descriptor[3] = image.mip_count | (image.srgb << 16);
				scene->materials.texel_count += image.texel_count;
				scene->materials.textured = VK_TRUE;
				vkr_free_host_texture(&image);
				free(path);
				continue;
			}
			float texel[4];
			if (status == 3) status = read_constant_texel(texel, path);
			free(path);
			if (status == 2) {
				printf("Failed to load material textures for the scene file at path %s using texture path %s.\n", file_path, texture_path);
				destroy_scene(scene, device);
				return 1;
			}
			if (status == 0) {
				uint32_t channels = (type == material_texture_type_normal) ? 2 : 3;
				memcpy(k + 3 * type, texel, sizeof(float) * channels);
			}
		}
	}
	if (device) {
		if (vkr_device_upload(&mesh->positions, device, mesh->host_positions, sizeof(uint32_t) * 2 * vertex_count, "vertex positions")
			|| vkr_device_upload(&mesh->normals_and_tex_coords, device, mesh->host_normals_and_tex_coords, sizeof(uint16_t) * 4 * vertex_count, "normals and texture coordinates")
			|| vkr_device_upload(&mesh->material_indices, device, mesh->host_material_indices, mesh->triangle_count, "material indices")
			|| vkr_device_upload(&scene->materials.constants, device, scene->materials.host_constants, sizeof(float) * 8 * (material_count ? material_count : 1), "material constants")
			|| (scene->materials.textured && (
				vkr_device_upload(&scene->materials.texture_descriptors, device, scene->materials.host_texture_descriptors, sizeof(uint32_t) * 12 * material_count, "texture descriptors")
				|| vkr_device_upload(&scene->materials.texels, device, scene->materials.host_texels, 4 * scene->materials.texel_count, "material textures")
				|| vkr_upload_srgb_table(&scene->materials.srgb_table, device))))
		{
			printf("Failed to copy mesh data of the scene file at path %s to the device. It has %llu triangles.\n", file_path, (unsigned long long) mesh->triangle_count);
			destroy_scene(scene, device);
			return 1;
		}
		if (request_acceleration_structure && device->ray_tracing_supported) {
			if (vkr_build_acceleration_structure(&scene->acceleration_structure, device, mesh, (int) request_acceleration_structure)) {
				printf("Failed to construct an acceleration structure for the scene file at path %s.\n", file_path);
				destroy_scene(scene, device);
				return 1;
			}
		}
	}
	return 0;
}
