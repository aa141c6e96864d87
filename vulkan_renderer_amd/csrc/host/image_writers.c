/* Screenshot writers: 8-bit RGB PNG and Radiance RGBE *.hdr, and the screenshot
 * procedure of the reference (src/main.c:1601-1770) on top of them.  The reference
 * calls the vendored stb_image_write.h; these are independent implementations of the
 * two file formats (PNG: RFC 1950/1951/2083; RGBE: Ward, Graphics Gems II), checked in
 * tests/test_experiments.py to decode to the same pixels as the files the reference's
 * writer produces. */
#include "vkr_internal.h"
#include "vkr_experiments.h"

/* ---- byte buffer ----------------------------------------------------------------- */

typedef struct {
	uint8_t* data;
	size_t size, capacity;
} bytes_t;

static int bytes_reserve(bytes_t* b, size_t extra) {
	if (b->size + extra <= b->capacity) return 0;
	size_t capacity = b->capacity ? b->capacity : 4096;
	while (capacity < b->size + extra) capacity *= 2;
	uint8_t* data = (uint8_t*) realloc(b->data, capacity);
	if (!data) return 1;
	b->data = data;
	b->capacity = capacity;
	return 0;
}

static int bytes_append(bytes_t* b, const void* source, size_t count) {
	if (bytes_reserve(b, count)) return 1;
	memcpy(b->data + b->size, source, count);
	b->size += count;
	return 0;
}

static int bytes_put(bytes_t* b, uint8_t value) { return bytes_append(b, &value, 1); }

/* ---- checksums ------------------------------------------------------------------- */

/* CRC-32 (polynomial 0xEDB88320) of one byte, four bits at a time: a 16-entry constant table,
 * so there is no lazily initialised state that two contexts could race on. */
static const uint32_t crc32_nibble_table[16] = {
	0x00000000u, 0x1DB71064u, 0x3B6E20C8u, 0x26D930ACu, 0x76DC4190u, 0x6B6B51F4u, 0x4DB26158u, 0x5005713Cu,
	0xEDB88320u, 0xF00F9344u, 0xD6D6A3E8u, 0xCB61B38Cu, 0x9B64C2B0u, 0x86D3D2D4u, 0xA00AE278u, 0xBDBDF21Cu};

static uint32_t crc32_update(uint32_t crc, const uint8_t* data, size_t count) {
	crc = ~crc;
	for (size_t i = 0; i != count; ++i) {
		crc ^= data[i];
		crc = crc32_nibble_table[crc & 0xF] ^ (crc >> 4);
		crc = crc32_nibble_table[crc & 0xF] ^ (crc >> 4);
	}
	return ~crc;
}

static uint32_t adler32(const uint8_t* data, size_t count) {
	uint32_t a = 1, b = 0;
	while (count) {
		size_t block = count < 5552 ? count : 5552;
		for (size_t i = 0; i != block; ++i) { a += data[i]; b += a; }
		a %= 65521; b %= 65521;
		data += block; count -= block;
	}
	return (b << 16) | a;
}

/* ---- deflate with the fixed Huffman code and a greedy hash-chain matcher ----------- */

typedef struct {
	bytes_t* out;
	uint32_t bit_buffer;
	int bit_count;
} bit_writer_t;

static void put_bits(bit_writer_t* w, uint32_t bits, int count) {
	w->bit_buffer |= bits << w->bit_count;
	w->bit_count += count;
	while (w->bit_count >= 8) {
		bytes_put(w->out, (uint8_t) (w->bit_buffer & 0xFF));
		w->bit_buffer >>= 8;
		w->bit_count -= 8;
	}
}

/* Huffman codes are sent most significant bit first */
static void put_code(bit_writer_t* w, uint32_t code, int length) {
	uint32_t reversed = 0;
	for (int i = 0; i != length; ++i) reversed |= ((code >> i) & 1u) << (length - 1 - i);
	put_bits(w, reversed, length);
}

static void put_literal_or_length_symbol(bit_writer_t* w, uint32_t symbol) {
	if (symbol < 144) put_code(w, 0x30 + symbol, 8);
	else if (symbol < 256) put_code(w, 0x190 + (symbol - 144), 9);
	else if (symbol < 280) put_code(w, symbol - 256, 7);
	else put_code(w, 0xC0 + (symbol - 280), 8);
}

static void put_match(bit_writer_t* w, uint32_t length, uint32_t distance) {
	static const uint16_t length_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
	static const uint8_t length_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
	static const uint16_t distance_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
	static const uint8_t distance_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
	int l = 28;
	while (length_base[l] > length) --l;
	put_literal_or_length_symbol(w, 257 + (uint32_t) l);
	if (length_extra[l]) put_bits(w, length - length_base[l], length_extra[l]);
	int d = 29;
	while (distance_base[d] > distance) --d;
	put_code(w, (uint32_t) d, 5);
	if (distance_extra[d]) put_bits(w, distance - distance_base[d], distance_extra[d]);
}

/* zlib stream (RFC 1950) around one fixed-Huffman deflate block (RFC 1951) */
static int zlib_compress(bytes_t* out, const uint8_t* data, size_t count) {
	enum { hash_bits = 15, hash_size = 1 << hash_bits, window = 32768, max_chain = 24 };
	int32_t* head = (int32_t*) malloc(sizeof(int32_t) * hash_size);
	int32_t* previous = (int32_t*) malloc(sizeof(int32_t) * window);
	if (!head || !previous) { free(head); free(previous); return 1; }
	memset(head, 0xFF, sizeof(int32_t) * hash_size);
	bytes_put(out, 0x78);
	bytes_put(out, 0x5E);
	bit_writer_t w = {out, 0, 0};
	put_bits(&w, 1, 1); /* final block */
	put_bits(&w, 1, 2); /* fixed Huffman codes */
	size_t i = 0;
	while (i < count) {
		uint32_t best_length = 0, best_distance = 0;
		if (i + 3 <= count) {
			uint32_t h = ((uint32_t) data[i] << 16 | (uint32_t) data[i + 1] << 8 | data[i + 2]) * 2654435761u >> (32 - hash_bits);
			int32_t candidate = head[h];
			size_t limit = count - i < 258 ? count - i : 258;
			for (int chain = 0; candidate >= 0 && i - (size_t) candidate <= window - 1 && chain != max_chain; ++chain) {
				const uint8_t* a = data + candidate;
				const uint8_t* b = data + i;
				if (a[best_length] == b[best_length]) {
					uint32_t length = 0;
					while (length < limit && a[length] == b[length]) ++length;
					if (length > best_length) { best_length = length; best_distance = (uint32_t) (i - (size_t) candidate); if (length == limit) break; }
				}
				int32_t next = previous[candidate & (window - 1)];
				if (next >= candidate) break;
				candidate = next;
			}
			previous[i & (window - 1)] = head[h];
			head[h] = (int32_t) i;
		}
		if (best_length >= 3) {
			put_match(&w, best_length, best_distance);
			/* index the skipped positions so that later matches can start inside them */
			for (size_t k = i + 1; k < i + best_length && k + 3 <= count; ++k) {
				uint32_t h = ((uint32_t) data[k] << 16 | (uint32_t) data[k + 1] << 8 | data[k + 2]) * 2654435761u >> (32 - hash_bits);
				previous[k & (window - 1)] = head[h];
				head[h] = (int32_t) k;
			}
			i += best_length;
		}
		else {
			put_literal_or_length_symbol(&w, data[i]);
			++i;
		}
	}
	put_literal_or_length_symbol(&w, 256);
	if (w.bit_count) put_bits(&w, 0, 8 - w.bit_count);
	uint32_t checksum = adler32(data, count);
	uint8_t trailer[4] = {(uint8_t) (checksum >> 24), (uint8_t) (checksum >> 16), (uint8_t) (checksum >> 8), (uint8_t) checksum};
	free(head);
	free(previous);
	return bytes_append(out, trailer, 4);
}

/* ---- PNG --------------------------------------------------------------------------- */

static int png_chunk(FILE* file, const char type[4], const uint8_t* data, uint32_t size) {
	uint8_t header[8] = {(uint8_t) (size >> 24), (uint8_t) (size >> 16), (uint8_t) (size >> 8), (uint8_t) size, (uint8_t) type[0], (uint8_t) type[1], (uint8_t) type[2], (uint8_t) type[3]};
	uint32_t crc = crc32_update(0, header + 4, 4);
	if (size) crc = crc32_update(crc, data, size);
	uint8_t tail[4] = {(uint8_t) (crc >> 24), (uint8_t) (crc >> 16), (uint8_t) (crc >> 8), (uint8_t) crc};
	return fwrite(header, 1, 8, file) != 8 || (size && fwrite(data, 1, size, file) != size) || fwrite(tail, 1, 4, file) != 4;
}

static inline int paeth(int a, int b, int c) {
	int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int write_png_rgb8(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb) {
	if (!width || !height || !rgb) {
		printf("Cannot write an empty image to %s.\n", path);
		return 1;
	}
	size_t stride = 3 * (size_t) width;
	uint8_t* filtered = (uint8_t*) malloc((stride + 1) * height);
	uint8_t* attempt = (uint8_t*) malloc(stride);
	if (!filtered || !attempt) { free(filtered); free(attempt); printf("Out of memory writing %s.\n", path); return 1; }
	/* per scanline the filter with the smallest sum of absolute residuals */
	for (uint32_t y = 0; y != height; ++y) {
		const uint8_t* row = rgb + stride * y;
		const uint8_t* above = y ? row - stride : NULL;
		uint8_t* out = filtered + (stride + 1) * y;
		uint64_t best_cost = ~0ull;
		for (int filter = 0; filter != 5; ++filter) {
			uint64_t cost = 0;
			for (size_t x = 0; x != stride; ++x) {
				int left = x >= 3 ? row[x - 3] : 0, up = above ? above[x] : 0, up_left = (above && x >= 3) ? above[x - 3] : 0;
				int predicted = filter == 0 ? 0 : filter == 1 ? left : filter == 2 ? up : filter == 3 ? (left + up) / 2 : paeth(left, up, up_left);
				attempt[x] = (uint8_t) (row[x] - predicted);
				cost += (uint64_t) abs((int) (int8_t) attempt[x]);
			}
			if (cost < best_cost) {
				best_cost = cost;
				out[0] = (uint8_t) filter;
				memcpy(out + 1, attempt, stride);
			}
		}
	}
	free(attempt);
	bytes_t compressed = {NULL, 0, 0};
	int failed = zlib_compress(&compressed, filtered, (stride + 1) * height);
	free(filtered);
	FILE* file = failed ? NULL : fopen(path, "wb");
	if (!file) {
		printf("Failed to store a screenshot to the *.png file at %s. Please check path and permissions.\n", path);
		free(compressed.data);
		return 1;
	}
	static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	uint8_t header[13] = {(uint8_t) (width >> 24), (uint8_t) (width >> 16), (uint8_t) (width >> 8), (uint8_t) width,
		(uint8_t) (height >> 24), (uint8_t) (height >> 16), (uint8_t) (height >> 8), (uint8_t) height, 8, 2, 0, 0, 0};
	failed = fwrite(signature, 1, 8, file) != 8
		|| png_chunk(file, "IHDR", header, 13)
		|| png_chunk(file, "IDAT", compressed.data, (uint32_t) compressed.size)
		|| png_chunk(file, "IEND", NULL, 0);
	failed |= fclose(file) != 0;
	free(compressed.data);
	if (failed) printf("Failed to store a screenshot to the *.png file at %s. Please check path and permissions.\n", path);
	return failed;
}

/* ---- Radiance RGBE ------------------------------------------------------------------- */

static void float_to_rgbe(uint8_t rgbe[4], const float* rgb) {
	float largest = rgb[0] > rgb[1] ? rgb[0] : rgb[1];
	if (rgb[2] > largest) largest = rgb[2];
	if (!(largest >= 1.0e-32f)) {
		rgbe[0] = rgbe[1] = rgbe[2] = rgbe[3] = 0;
		return;
	}
	int exponent;
	float scale = (float) frexp(largest, &exponent) * 256.0f / largest;
	rgbe[0] = (uint8_t) (rgb[0] * scale);
	rgbe[1] = (uint8_t) (rgb[1] * scale);
	rgbe[2] = (uint8_t) (rgb[2] * scale);
	rgbe[3] = (uint8_t) (exponent + 128);
}

/* One component plane of a scanline: runs of 3..127 equal bytes become (128 + n, value),
   everything else literal chunks (n <= 128, bytes) */
static int rle_component(bytes_t* out, const uint8_t* values, uint32_t count) {
	uint32_t x = 0;
	while (x < count) {
		/* find the next run of at least three */
		uint32_t run_start = x;
		while (run_start + 2 < count && !(values[run_start] == values[run_start + 1] && values[run_start] == values[run_start + 2])) ++run_start;
		if (run_start + 2 >= count) run_start = count;
		while (x < run_start) {
			uint32_t chunk = run_start - x > 128 ? 128 : run_start - x;
			if (bytes_put(out, (uint8_t) chunk) || bytes_append(out, values + x, chunk)) return 1;
			x += chunk;
		}
		if (run_start + 2 < count) {
			uint32_t run_end = run_start + 3;
			while (run_end < count && values[run_end] == values[run_start]) ++run_end;
			while (x < run_end) {
				uint32_t chunk = run_end - x > 127 ? 127 : run_end - x;
				if (bytes_put(out, (uint8_t) (128 + chunk)) || bytes_put(out, values[run_start])) return 1;
				x += chunk;
			}
		}
	}
	return 0;
}

int write_hdr_rgb32f(const char* path, uint32_t width, uint32_t height, const float* rgb) {
	if (!width || !height || !rgb) {
		printf("Cannot write an empty image to %s.\n", path);
		return 1;
	}
	bytes_t out = {NULL, 0, 0};
	char header[160];
	int header_size = snprintf(header, sizeof(header), "#?RADIANCE\n# Written by libvkr_shading\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %u +X %u\n", height, width);
	int failed = bytes_append(&out, header, (size_t) header_size);
	uint8_t* planes = (uint8_t*) malloc(4 * (size_t) width);
	if (!planes) failed = 1;
	for (uint32_t y = 0; y != height && !failed; ++y) {
		const float* row = rgb + 3 * (size_t) width * y;
		if (width < 8 || width >= 32768) {
			/* the format allows run-length encoding only for these widths */
			for (uint32_t x = 0; x != width && !failed; ++x) {
				uint8_t rgbe[4];
				float_to_rgbe(rgbe, row + 3 * x);
				failed = bytes_append(&out, rgbe, 4);
			}
			continue;
		}
		for (uint32_t x = 0; x != width; ++x) {
			uint8_t rgbe[4];
			float_to_rgbe(rgbe, row + 3 * x);
			for (int c = 0; c != 4; ++c) planes[(size_t) c * width + x] = rgbe[c];
		}
		uint8_t scanline_header[4] = {2, 2, (uint8_t) (width >> 8), (uint8_t) (width & 0xFF)};
		failed = bytes_append(&out, scanline_header, 4);
		for (int c = 0; c != 4 && !failed; ++c) failed = rle_component(&out, planes + (size_t) c * width, width);
	}
	free(planes);
	FILE* file = failed ? NULL : fopen(path, "wb");
	if (file) {
		failed = fwrite(out.data, 1, out.size, file) != out.size;
		failed |= fclose(file) != 0;
	}
	else failed = 1;
	free(out.data);
	if (failed) printf("Failed to store a screenshot to the *.hdr file at %s. Please check path and permissions.\n", path);
	return failed;
}

/* ---- screenshots --------------------------------------------------------------------- */

float half_to_float(uint16_t half) {
	/* exact widening: sign, 5-bit exponent (bias 15), 10-bit mantissa; subnormals and
	   inf/NaN handled explicitly (same values as reference math_utilities.h:70-84) */
	uint32_t sign = (uint32_t) (half & 0x8000) << 16;
	uint32_t exponent = (half >> 10) & 0x1F, mantissa = half & 0x3FF;
	uint32_t bits;
	if (exponent == 0x1F) bits = sign | 0x7F800000u | (mantissa << 13);
	else if (exponent != 0) bits = sign | ((exponent + 112) << 23) | (mantissa << 13);
	else if (mantissa == 0) bits = sign;
	else {
		/* subnormal half: normalise */
		int shift = 0;
		while (!(mantissa & 0x400)) { mantissa <<= 1; ++shift; }
		bits = sign | ((uint32_t) (113 - shift) << 23) | ((mantissa & 0x3FF) << 13);
	}
	float result;
	memcpy(&result, &bits, sizeof(result));
	return result;
}

int take_screenshot(application_t* app, const char* path_png, const char* path_hdr) {
	if (!path_png && !path_hdr) return 0;
	uint32_t width = app->swapchain.extent.width, height = app->swapchain.extent.height;
	size_t pixel_count = (size_t) width * height;
	uint8_t* rgba = (uint8_t*) malloc(4 * pixel_count);
	uint8_t* ldr = (uint8_t*) malloc(3 * pixel_count * (path_hdr ? 2 : 1));
	float* hdr = path_hdr ? (float*) malloc(sizeof(float) * 3 * pixel_count) : NULL;
	uint32_t frame_bits_before = app->screenshot.frame_bits;
	int failed = !rgba || !ldr || (path_hdr && !hdr);
	if (failed) printf("Out of memory taking a screenshot.\n");
	if (!failed && path_png) {
		/* LDR frame: sRGB transfer function, 8 bits, alpha dropped (main.c:1664-1674) */
		app->screenshot.frame_bits = 0;
		failed = encode_output(app, VK_FALSE) || read_back_encoded(app, rgba);
		for (size_t i = 0; i != pixel_count && !failed; ++i) memcpy(ldr + 3 * i, rgba + 4 * i, 3);
		if (!failed) failed = write_png_rgb8(path_png, width, height, ldr);
		if (!failed) printf("Wrote screenshot to %s.\n", path_png);
	}
	if (!failed && path_hdr) {
		/* HDR frame: low bytes, then high bytes of the half-precision colour (main.c:1700-1711) */
		for (uint32_t pass = 0; pass != 2 && !failed; ++pass) {
			app->screenshot.frame_bits = 1 + pass;
			failed = encode_output(app, VK_FALSE) || read_back_encoded(app, rgba);
			for (size_t i = 0; i != pixel_count && !failed; ++i) memcpy(ldr + 3 * (pass * pixel_count + i), rgba + 4 * i, 3);
		}
		for (size_t i = 0; i != 3 * pixel_count && !failed; ++i)
			hdr[i] = half_to_float((uint16_t) (ldr[i] | ((uint16_t) ldr[i + 3 * pixel_count] << 8)));
		if (!failed) failed = write_hdr_rgb32f(path_hdr, width, height, hdr);
		if (!failed) printf("Wrote screenshot to %s.\n", path_hdr);
	}
	app->screenshot.frame_bits = frame_bits_before;
	free(rgba); free(ldr); free(hdr);
	return failed;
}
