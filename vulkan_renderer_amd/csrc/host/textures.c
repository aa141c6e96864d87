/* Material textures: the *.vkt container of the reference (src/textures.c:95-241, written by
 * tools/texture_conversion) decoded to RGBA8 mip chains on the host.
 *
 * The reference uploads the blocks as they are and lets the GPU's texture unit decode BC1 / BC5
 * and filter anisotropically (scene.c:486-559); here the blocks are decoded once at load time and
 * the filtering is done in software by the material resolve kernel (anisotropic: up to 16
 * trilinear taps along the footprint's longer axis, sample_texture() in csrc/shading_kernel.h).  Block decoding follows the format definitions (endpoint expansion
 * by bit replication, interpolants rounded to nearest); what a given GPU does in the last bit
 * is not specified by Vulkan, so this is a documented choice, not a pinned one. */
#include "vkr_internal.h"

enum {
	format_r8g8b8a8_unorm = 37, format_r8g8b8a8_srgb = 43,
	format_bc1_rgb_unorm = 131, format_bc1_rgb_srgb = 132, format_bc1_rgba_unorm = 133, format_bc1_rgba_srgb = 134,
	format_bc5_unorm = 141,
};

void vkr_free_host_texture(vkr_host_texture_t* texture) {
	free(texture->texels);
	memset(texture, 0, sizeof(*texture));
}

static inline uint8_t expand_5(uint32_t v) { return (uint8_t) ((v << 3) | (v >> 2)); }
static inline uint8_t expand_6(uint32_t v) { return (uint8_t) ((v << 2) | (v >> 4)); }

/* one BC1 block (8 bytes) -> 16 RGBA8 texels, row-major within the block */
void vkr_decode_bc1_block(const uint8_t block[8], uint8_t out_rgba[64], int has_alpha) {
	uint32_t c0 = block[0] | (block[1] << 8), c1 = block[2] | (block[3] << 8);
	uint8_t colors[4][4] = {
		{expand_5(c0 >> 11), expand_6((c0 >> 5) & 63), expand_5(c0 & 31), 255},
		{expand_5(c1 >> 11), expand_6((c1 >> 5) & 63), expand_5(c1 & 31), 255}};
	for (int c = 0; c != 3; ++c) {
		if (c0 > c1) {
			colors[2][c] = (uint8_t) ((2 * colors[0][c] + colors[1][c] + 1) / 3);
			colors[3][c] = (uint8_t) ((colors[0][c] + 2 * colors[1][c] + 1) / 3);
		}
		else {
			colors[2][c] = (uint8_t) ((colors[0][c] + colors[1][c] + 1) / 2);
			colors[3][c] = 0;
		}
	}
	colors[2][3] = 255;
	colors[3][3] = (c0 > c1 || !has_alpha) ? 255 : 0;
	uint32_t indices = block[4] | (block[5] << 8) | (block[6] << 16) | ((uint32_t) block[7] << 24);
	for (int i = 0; i != 16; ++i) memcpy(out_rgba + 4 * i, colors[(indices >> (2 * i)) & 3], 4);
}

/* one BC4 half block (8 bytes) -> 16 values */
static void decode_bc4_block(const uint8_t block[8], uint8_t out[16]) {
	uint32_t e0 = block[0], e1 = block[1];
	uint8_t values[8] = {(uint8_t) e0, (uint8_t) e1};
	if (e0 > e1)
		for (uint32_t i = 1; i != 7; ++i) values[i + 1] = (uint8_t) (((7 - i) * e0 + i * e1 + 3) / 7);
	else {
		for (uint32_t i = 1; i != 5; ++i) values[i + 1] = (uint8_t) (((5 - i) * e0 + i * e1 + 2) / 5);
		values[6] = 0;
		values[7] = 255;
	}
	uint64_t indices = 0;
	for (int i = 0; i != 6; ++i) indices |= (uint64_t) block[2 + i] << (8 * i);
	for (int i = 0; i != 16; ++i) out[i] = values[(indices >> (3 * i)) & 7];
}

/* one BC5 block (16 bytes) -> 16 RGBA8 texels with blue 0 and alpha 255 */
void vkr_decode_bc5_block(const uint8_t block[16], uint8_t out_rgba[64]) {
	uint8_t red[16], green[16];
	decode_bc4_block(block, red);
	decode_bc4_block(block + 8, green);
	for (int i = 0; i != 16; ++i) {
		out_rgba[4 * i + 0] = red[i]; out_rgba[4 * i + 1] = green[i]; out_rgba[4 * i + 2] = 0; out_rgba[4 * i + 3] = 255;
	}
}

/* Returns 0 on success, 1 if the file is absent, 2 if it is invalid, 3 if it is a valid *.vkt of
 * a format that is not decoded here (half / float: the caller reduces those to constants). */
int vkr_load_texture_rgba8(vkr_host_texture_t* out, const char* path) {
	memset(out, 0, sizeof(*out));
	FILE* file = fopen(path, "rb");
	if (!file) return 1;
	int32_t header[6];
	uint64_t payload_size;
	if (fread(header, sizeof(int32_t), 6, file) != 6 || fread(&payload_size, sizeof(uint64_t), 1, file) != 1
		|| header[0] != 0xbc1bc1 || header[1] != 1 || header[2] < 1 || header[2] > 32 || header[3] < 1 || header[4] < 1)
	{
		printf("The texture at path %s does not seem to have the correct format. It is supposed to be converted to a custom format for the renderer using the texture conversion utility. Aborting.\n", path);
		fclose(file);
		return 2;
	}
	int32_t mip_count = header[2], format = header[5];
	int is_bc1 = format >= format_bc1_rgb_unorm && format <= format_bc1_rgba_srgb;
	int is_bc5 = format == format_bc5_unorm, is_rgba8 = format == format_r8g8b8a8_unorm || format == format_r8g8b8a8_srgb;
	if (!is_bc1 && !is_bc5 && !is_rgba8) {
		fclose(file);
		return 3;
	}
	uint32_t widths[32], heights[32];
	uint64_t sizes[32], offsets[32], texel_count = 0;
	for (int32_t m = 0; m != mip_count; ++m) {
		int32_t extent[2];
		uint64_t size_offset[2];
		if (fread(extent, sizeof(int32_t), 2, file) != 2 || fread(size_offset, sizeof(uint64_t), 2, file) != 2 || extent[0] < 1 || extent[1] < 1) { fclose(file); return 2; }
		widths[m] = (uint32_t) extent[0]; heights[m] = (uint32_t) extent[1];
		sizes[m] = size_offset[0]; offsets[m] = size_offset[1];
		/* the sampler walks the chain by halving: the file has to store exactly that chain */
		uint32_t expected_w = (uint32_t) header[3] >> m, expected_h = (uint32_t) header[4] >> m;
		if (widths[m] != (expected_w ? expected_w : 1) || heights[m] != (expected_h ? expected_h : 1)) {
			printf("The texture at path %s has an unexpected mipmap chain.\n", path);
			fclose(file);
			return 2;
		}
		texel_count += (uint64_t) widths[m] * heights[m];
	}
	uint8_t* payload = (uint8_t*) malloc(payload_size ? payload_size : 1);
	uint32_t eof_marker = 0;
	int ok = payload && fread(payload, 1, payload_size, file) == payload_size && fread(&eof_marker, sizeof(eof_marker), 1, file) == 1 && eof_marker == 0xE0FE0F;
	fclose(file);
	if (!ok) {
		printf("The texture file at path %s seems to be invalid. The texture data is not followed by the expected end of file marker.\n", path);
		free(payload);
		return 2;
	}
	out->texels = (uint8_t*) malloc(texel_count ? 4 * texel_count : 1);
	if (!out->texels) {
		printf("Out of memory while decoding the texture file at path %s.\n", path);
		free(payload);
		return 2;
	}
	out->width = widths[0]; out->height = heights[0]; out->mip_count = (uint32_t) mip_count;
	out->srgb = format == format_r8g8b8a8_srgb || format == format_bc1_rgb_srgb || format == format_bc1_rgba_srgb;
	out->texel_count = texel_count;
	uint8_t* target = out->texels;
	for (int32_t m = 0; m != mip_count && ok; ++m) {
		uint32_t w = widths[m], h = heights[m];
		const uint8_t* source = payload + offsets[m];
		uint64_t block_bytes = is_bc5 ? 16 : 8;
		uint64_t needed = is_rgba8 ? 4ull * w * h : block_bytes * ((w + 3) / 4) * ((h + 3) / 4);
		/* (written so that a crafted offset near 2^64 cannot wrap the sum) */
		if (offsets[m] > payload_size || needed > payload_size - offsets[m] || sizes[m] < needed) { ok = 0; break; }
		if (is_rgba8) memcpy(target, source, 4ull * w * h);
		else
			for (uint32_t by = 0; by != (h + 3) / 4; ++by)
				for (uint32_t bx = 0; bx != (w + 3) / 4; ++bx) {
					uint8_t texels[64];
					const uint8_t* block = source + block_bytes * ((uint64_t) by * ((w + 3) / 4) + bx);
					if (is_bc5) vkr_decode_bc5_block(block, texels);
					else vkr_decode_bc1_block(block, texels, format == format_bc1_rgba_unorm || format == format_bc1_rgba_srgb);
					for (uint32_t y = 0; y != 4 && 4 * by + y < h; ++y)
						for (uint32_t x = 0; x != 4 && 4 * bx + x < w; ++x)
							memcpy(target + 4 * ((uint64_t) (4 * by + y) * w + 4 * bx + x), texels + 4 * (4 * y + x), 4);
				}
		target += 4ull * w * h;
	}
	free(payload);
	if (!ok) {
		printf("The texture file at path %s is truncated.\n", path);
		vkr_free_host_texture(out);
		return 2;
	}
	return 0;
}
