/* TEST INFRASTRUCTURE — not part of the product.
 *
 * CPU restatement (C99, scalar fp32, -ffp-contract=off) of the per-pixel shading
 * program of the reference.  Every function names the GLSL it follows; paths are
 * relative to /root/reference/src/shaders/.  Vertex arrays have capacity
 * O_CAP and the reference's compile-time MAX_POLYGON_VERTEX_COUNT is the runtime
 * argument `cap` ("max_count" at the API), which is legal because results do not
 * depend on it as long as the repeated-first-vertex convention holds
 * (polygon_sampling.glsl:514-515). */
#include "oracle.h"
#include "oracle_math.h"
#include <stdlib.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define O_CAP 9

int g_oracle_math_mode = 0;
static uint64_t g_ray_count = 0;
static int g_log_rays = 0;
void oracle_set_ray_log(int on) { g_log_rays = on; }

void oracle_set_math_mode(int mode) { g_oracle_math_mode = mode; }
int g_oracle_system_libm = 0;
void oracle_set_libm_source(int use_system_library) { g_oracle_system_libm = use_system_library; }
int oracle_get_libm_source(void) { return g_oracle_system_libm; }
uint64_t oracle_last_ray_count(void) { return g_ray_count; }

/* GLSL min/max as the specification words them */
static inline float g_max(float x, float y) { return (x < y) ? y : x; }
static inline float g_min(float x, float y) { return (y < x) ? y : x; }
static inline float g_clamp(float x, float lo, float hi) { return g_min(g_max(x, lo), hi); }
/* max(+0.0f, x) as used for determinants: NaN and -0 become +0 */
static inline float positive_part(float x) { return (x > 0.0f) ? x : 0.0f; }

/* ------------------------------------------------------------------------ */
/* constants buffer views                                                   */

typedef struct {
	v3 dequant_factor, dequant_summand;
	float pixel_to_ray[3][4];
	v3 camera_position;
	float mis_visibility_estimate, exposure_factor, roughness_factor, error_factor;
	uint32_t noise_resolution_mask[2], noise_texture_index_mask, frame_bits;
	uint32_t noise_random_numbers[4];
	float ltc_fresnel_factor, ltc_fresnel_summand, ltc_roughness_factor, ltc_roughness_summand, ltc_inclination_factor, ltc_inclination_summand;
} frame_constants_t;

typedef struct {
	v3 translation, surface_radiance;
	float scaling_x, scaling_y, inv_scaling_x, inv_scaling_y;
	v4 plane;
	uint32_t vertex_count, texturing_technique, texture_index;
	v3 rotation_columns[3];
	v3 vertices_world[O_CAP];
	/* world-space area; per fan triangle (area of the triangle, area of the fan so far) */
	float area;
	v2 fan_areas[O_CAP];
} light_view_t;

static float rd_f(const uint8_t* p, size_t off) { float f; memcpy(&f, p + off, 4); return f; }
static uint32_t rd_u(const uint8_t* p, size_t off) { uint32_t u; memcpy(&u, p + off, 4); return u; }

/* byte offsets: reference main.h:488-505 / shared_constants.glsl:20-66 */
static frame_constants_t read_frame_constants(const uint8_t* c) {
	frame_constants_t k;
	k.dequant_factor = mk3(rd_f(c, 0), rd_f(c, 4), rd_f(c, 8));
	k.dequant_summand = mk3(rd_f(c, 16), rd_f(c, 20), rd_f(c, 24));
	for (int i = 0; i != 3; ++i)
		for (int j = 0; j != 4; ++j)
			k.pixel_to_ray[i][j] = rd_f(c, 96 + 16 * i + 4 * j);
	k.camera_position = mk3(rd_f(c, 144), rd_f(c, 148), rd_f(c, 152));
	k.mis_visibility_estimate = rd_f(c, 156);
	k.error_factor = rd_f(c, 28);
	k.exposure_factor = rd_f(c, 176);
	k.roughness_factor = rd_f(c, 180);
	k.noise_resolution_mask[0] = rd_u(c, 184);
	k.noise_resolution_mask[1] = rd_u(c, 188);
	k.noise_texture_index_mask = rd_u(c, 192);
	k.frame_bits = rd_u(c, 196);
	for (int i = 0; i != 4; ++i) k.noise_random_numbers[i] = rd_u(c, 208 + 4 * i);
	k.ltc_fresnel_factor = rd_f(c, 224);
	k.ltc_fresnel_summand = rd_f(c, 228);
	k.ltc_roughness_factor = rd_f(c, 232);
	k.ltc_roughness_summand = rd_f(c, 236);
	k.ltc_inclination_factor = rd_f(c, 240);
	k.ltc_inclination_summand = rd_f(c, 244);
	return k;
}

/* light record: polygonal_light_utility.glsl:26-83, host mirror polygonal_light.h:100-129 */
static light_view_t read_light(const uint8_t* constants, uint32_t index, uint32_t vmax) {
	size_t stride = 160 + 16 * (size_t) vmax * 2 + 16 * ((size_t) vmax - 2);
	const uint8_t* p = constants + 256 + stride * index;
	light_view_t l;
	memset(&l, 0, sizeof(l));
	l.scaling_x = rd_f(p, 12);
	l.translation = mk3(rd_f(p, 16), rd_f(p, 20), rd_f(p, 24));
	l.scaling_y = rd_f(p, 28);
	l.inv_scaling_x = rd_f(p, 44);
	l.surface_radiance = mk3(rd_f(p, 48), rd_f(p, 52), rd_f(p, 56));
	l.inv_scaling_y = rd_f(p, 60);
	l.plane.x = rd_f(p, 64); l.plane.y = rd_f(p, 68); l.plane.z = rd_f(p, 72); l.plane.w = rd_f(p, 76);
	l.vertex_count = rd_u(p, 80);
	l.texturing_technique = rd_u(p, 84);
	l.texture_index = rd_u(p, 88);
	/* the UBO is row_major, so GLSL column k is (row0[k], row1[k], row2[k]) */
	for (int k = 0; k != 3; ++k)
		l.rotation_columns[k] = mk3(rd_f(p, 96 + 4 * k), rd_f(p, 112 + 4 * k), rd_f(p, 128 + 4 * k));
	const uint8_t* world = p + 160 + 16 * (size_t) vmax;
	for (uint32_t i = 0; i != vmax && i != O_CAP; ++i)
		l.vertices_world[i] = mk3(rd_f(world, 16 * i), rd_f(world, 16 * i + 4), rd_f(world, 16 * i + 8));
	l.area = rd_f(p, 144);
	const uint8_t* fans = p + 160 + 32 * (size_t) vmax;
	for (uint32_t i = 0; i + 2 < vmax && i != O_CAP; ++i)
		l.fan_areas[i] = mk2(rd_f(fans, 16 * i), rd_f(fans, 16 * i + 4));
	return l;
}

/* ------------------------------------------------------------------------ */
/* noise_utility.glsl                                                       */

typedef struct {
	float noise[4];
	uint32_t available, pixel[2], sample_index;
} noise_accessor_t;

/* get_noise_sample, noise_utility.glsl:63-73 */
static void fetch_noise(const oracle_frame_t* f, const frame_constants_t* k, const uint32_t pixel[2], uint32_t sample_index, float out[4]) {
	uint32_t r[4];
	const uint32_t* n = k->noise_random_numbers;
	if (sample_index & 2) { r[0] = n[2]; r[1] = n[3]; r[2] = n[0]; r[3] = n[1]; }
	else { r[0] = n[0]; r[1] = n[1]; r[2] = n[2]; r[3] = n[3]; }
	if (sample_index & 1) { r[0] = r[1]; r[1] = r[2]; r[2] = r[3]; }
	uint32_t shift = (sample_index & 124) >> 2;
	uint32_t ox = r[0] >> shift, oy = r[1] >> shift;
	uint32_t layer = (r[2] + sample_index) & k->noise_texture_index_mask;
	uint32_t sx = (pixel[0] + ox) & k->noise_resolution_mask[0];
	uint32_t sy = (pixel[1] + oy) & k->noise_resolution_mask[1];
	const uint16_t* texel = f->noise + 4 * (((size_t) layer * f->noise_height + sy) * f->noise_width + sx);
	for (int i = 0; i != 4; ++i) out[i] = (float) texel[i] / 65535.0f;
}

/* get_noise_2, noise_utility.glsl:93-103 */
static v2 next_noise_2(const oracle_frame_t* f, const frame_constants_t* k, noise_accessor_t* a) {
	if (a->available <= 1) {
		fetch_noise(f, k, a->pixel, a->sample_index, a->noise);
		a->available = 4;
		++a->sample_index;
	}
	a->available -= 2;
	v2 r = mk2(a->noise[0], a->noise[1]);
	a->noise[0] = a->noise[2];
	a->noise[1] = a->noise[3];
	return r;
}

/* ------------------------------------------------------------------------ */
/* mesh_quantization.glsl                                                   */

/* decode_position_64_bit, mesh_quantization.glsl:38-45 */
static v3 decode_position(uint32_t q0, uint32_t q1, v3 factor, v3 summand) {
	float px = (float) (q0 & 0x1FFFFF);
	float py = (float) (((q0 & 0xFFE00000u) >> 21) | ((q1 & 0x3FF) << 11));
	float pz = (float) ((q1 & 0x7FFFFC00u) >> 10);
	return mk3(fmaf(px, factor.x, summand.x), fmaf(py, factor.y, summand.y), fmaf(pz, factor.z, summand.z));
}

/* decode_normal_32_bit, mesh_quantization.glsl:19-33 */
static v3 decode_normal(float ox, float oy) {
	const float factor = 2.0f * (65534.0f / 65535.0f);
	const float summand = -(32768.0f / 65535.0f) * factor;
	ox = fmaf(ox, factor, summand);
	oy = fmaf(oy, factor, summand);
	v3 n = mk3(ox, oy, 1.0f - fabsf(ox) - fabsf(oy));
	float sx = (ox >= 0.0f) ? 1.0f : -1.0f;
	float sy = (oy >= 0.0f) ? 1.0f : -1.0f;
	if (n.z < 0.0f) {
		float nx = (1.0f - fabsf(n.y)) * sx;
		float ny = (1.0f - fabsf(n.x)) * sy;
		n.x = nx; n.y = ny;
	}
	return normalize3(n);
}

/* ------------------------------------------------------------------------ */
/* brdfs.glsl                                                               */

typedef struct {
	v3 position, normal, outgoing;
	float lambert_outgoing;
	v3 diffuse_albedo, fresnel_0;
	float roughness;
} shading_data_t;

/* fresnel_schlick, brdfs.glsl:42-46 (scalar channel) */
static float schlick(float f0, float f90, float cos_theta) {
	float flipped = 1.0f - cos_theta;
	float flipped_squared = flipped * flipped;
	return f0 + (f90 - f0) * (flipped_squared * flipped * flipped_squared);
}

/* evaluate_brdf, brdfs.glsl:57-88 */
static v3 evaluate_brdf(const shading_data_t* d, v3 incoming, int diffuse, int specular) {
	v3 half_vector = normalize3(add3(incoming, d->outgoing));
	float lambert_incoming = dot3(d->normal, incoming);
	float outgoing_dot_half = dot3(d->outgoing, half_vector);
	v3 brdf = mk3(0.0f, 0.0f, 0.0f);
	if (diffuse) {
		float f90 = fmaf(outgoing_dot_half * outgoing_dot_half, 2.0f * d->roughness, 0.5f);
		float product = schlick(1.0f, f90, d->lambert_outgoing) * schlick(1.0f, f90, lambert_incoming);
		brdf = add3(brdf, scale3(d->diffuse_albedo, product));
	}
	if (specular) {
		float normal_dot_half = dot3(d->normal, half_vector);
		float a2 = d->roughness * d->roughness;
		float ggx = fmaf(fmaf(normal_dot_half, a2, -normal_dot_half), normal_dot_half, 1.0f);
		ggx = a2 / (ggx * ggx);
		float masking = lambert_incoming * sqrtf(fmaf(fmaf(-d->lambert_outgoing, a2, d->lambert_outgoing), d->lambert_outgoing, a2));
		float shadowing = d->lambert_outgoing * sqrtf(fmaf(fmaf(-lambert_incoming, a2, lambert_incoming), lambert_incoming, a2));
		float smith = 0.5f / (masking + shadowing);
		float c = g_clamp(outgoing_dot_half, 0.0f, 1.0f);
		float gs = ggx * smith;
		brdf = add3(brdf, mk3(gs * schlick(d->fresnel_0.x, 1.0f, c), gs * schlick(d->fresnel_0.y, 1.0f, c), gs * schlick(d->fresnel_0.z, 1.0f, c)));
	}
	return scale3(brdf, O_INV_PI);
}

/* get_ggx_visible_normal_density, brdfs.glsl:180-191 */
static float ggx_vndf_density(float out_dot_n, float micro_dot_n, float micro_dot_out, float roughness) {
	float a2 = roughness * roughness;
	float ggx = fmaf(fmaf(micro_dot_n, a2, -micro_dot_n), micro_dot_n, 1.0f);
	ggx = a2 / (ggx * ggx);
	ggx *= O_INV_PI;
	float masking = sqrtf(fmaf(fmaf(-out_dot_n, a2, out_dot_n), out_dot_n, a2));
	masking = 2.0f / (out_dot_n + masking);
	return masking * micro_dot_out * ggx;
}

/* sample_ggx_visible_normal_distribution, brdfs.glsl:122-162 (isotropic use) */
static v3 sample_ggx_vndf(v3 out_shading, float rx, float ry, v2 u) {
	m3 e2h;
	e2h.c[2] = normalize3(mk3(rx * out_shading.x, ry * out_shading.y, 1.0f * out_shading.z));
	float length_sq = e2h.c[2].x * e2h.c[2].x + e2h.c[2].y * e2h.c[2].y;
	float inv_len = rsqrt_f(length_sq);
	e2h.c[0] = mk3(-e2h.c[2].y * inv_len, e2h.c[2].x * inv_len, 0.0f * inv_len);
	if (length_sq <= 0.0f) e2h.c[0] = mk3(1.0f, 0.0f, 0.0f);
	e2h.c[1] = cross3(e2h.c[2], e2h.c[0]);
	float radius = sqrtf(u.x);
	float azimuth = (2.0f * O_PI) * u.y;
	float sn, cs;
	o_sincos(azimuth, &sn, &cs);
	v2 disk = mk2(radius * cs, radius * sn);
	v3 s;
	s.x = disk.x;
	float lerp = fmaf(0.5f, e2h.c[2].z, 0.5f);
	float a = sqrtf(fmaf(-disk.x, disk.x, 1.0f));
	/* mix(a, b, t) = a * (1 - t) + b * t */
	s.y = a * (1.0f - lerp) + disk.y * lerp;
	s.z = sqrtf(g_max(0.0f, 1.0f - (s.x * s.x + s.y * s.y)));
	v3 hemi = m3_mul(&e2h, s);
	return normalize3(mk3(rx * hemi.x, ry * hemi.y, 1.0f * hemi.z));
}

/* sample_ggx_reflected_direction, brdfs.glsl:200-210 */
static v3 sample_ggx_reflected(float* out_density, v3 out_shading, float roughness, v2 u) {
	v3 micro = sample_ggx_vndf(out_shading, roughness, roughness, u);
	float micro_dot_out = dot3(micro, out_shading);
	float density = ggx_vndf_density(out_shading.z, micro.z, micro_dot_out, roughness);
	v3 incoming = fma3s(2.0f * micro_dot_out, micro, neg3(out_shading));
	density /= 4.0f * micro_dot_out;
	*out_density = density;
	return incoming;
}

/* get_ggx_reflected_direction_density, brdfs.glsl:214-224 */
static float ggx_reflected_density(float out_dot_n, v3 out_dir, v3 in_dir, v3 normal, float roughness) {
	v3 micro = normalize3(add3(out_dir, in_dir));
	float micro_dot_out = dot3(micro, out_dir);
	float micro_dot_n = dot3(micro, normal);
	float density = ggx_vndf_density(out_dot_n, micro_dot_n, micro_dot_out, roughness);
	density /= 4.0f * micro_dot_out;
	return density;
}

/* ------------------------------------------------------------------------ */
/* get_shading_data, shading_pass.frag.glsl:721-822                          */

/* ---- material textures: the software sampler (unpinned by the reference, see oracle.h) ---- */

const float* oracle_srgb_table(void) {
	static float table[256];
	static int ready = 0;
	if (!ready) {
		for (int i = 0; i != 256; ++i) {
			float v = (float) i / 255.0f;
			table[i] = (v <= 0.04045f) ? (v / 12.92f) : l_powf((v + 0.055f) / 1.055f, 2.4f);
		}
		ready = 1;
	}
	return table;
}

static void fetch_texel(const oracle_texture_t* t, const uint8_t* level, int width, int height, int x, int y, float out[4]) {
	x = ((x % width) + width) % width;
	y = ((y % height) + height) % height;
	const uint8_t* texel = level + 4 * ((size_t) y * (size_t) width + (size_t) x);
	const float* srgb = oracle_srgb_table();
	for (int c = 0; c != 3; ++c) out[c] = t->srgb ? srgb[texel[c]] : (float) texel[c] * (1.0f / 255.0f);
	out[3] = (float) texel[3] * (1.0f / 255.0f);
}

static void sample_level(const oracle_texture_t* t, uint32_t level, float u, float v, float out[4]) {
	const uint8_t* texels = t->texels;
	int width = (int) t->width, height = (int) t->height;
	for (uint32_t l = 0; l != level; ++l) {
		texels += 4 * (size_t) width * (size_t) height;
		width = width > 1 ? width / 2 : 1;
		height = height > 1 ? height / 2 : 1;
	}
	float x = u * (float) width - 0.5f, y = v * (float) height - 0.5f;
	float x0 = floorf(x), y0 = floorf(y);
	float fx = x - x0, fy = y - y0;
	float t00[4], t10[4], t01[4], t11[4];
	fetch_texel(t, texels, width, height, (int) x0, (int) y0, t00);
	fetch_texel(t, texels, width, height, (int) x0 + 1, (int) y0, t10);
	fetch_texel(t, texels, width, height, (int) x0, (int) y0 + 1, t01);
	fetch_texel(t, texels, width, height, (int) x0 + 1, (int) y0 + 1, t11);
	for (int c = 0; c != 4; ++c) {
		float top = t00[c] * (1.0f - fx) + t10[c] * fx;
		float bottom = t01[c] * (1.0f - fx) + t11[c] * fx;
		out[c] = top * (1.0f - fy) + bottom * fy;
	}
}

/* textureGrad with the sampler of src/scene.c:546-552 (linear filters, repeat, 16x anisotropy).  What a driver does with
 * "anisotropy" is its own business; this build does what the Vulkan specification sketches (Texel Anisotropic Filtering; the
 * example of GL_EXT_texture_filter_anisotropic): the footprint's longer axis P_max and shorter axis P_min in texels,
 * N = min(ceil(P_max / P_min), 16) taps - never more than the footprint is texels long -, each a trilinear sample at
 * level log2(P_max / N), spread evenly along the longer axis at uv + (i / (N + 1) - 1 / 2) d(uv), i = 1 ... N, and averaged
 * in that order.  N = 1 is the isotropic trilinear sample of rounds 1 - 4 in every bit. */
#define ORACLE_MAX_ANISOTROPY 16.0f
void oracle_sample_texture(const oracle_texture_t* t, const float uv[2], const float duv_dx[2], const float duv_dy[2], float out_rgba[4]) {
	float w = (float) t->width, h = (float) t->height;
	float ax = duv_dx[0] * w, ay = duv_dx[1] * h, bx = duv_dy[0] * w, by = duv_dy[1] * h;
	float px = sqrtf(ax * ax + ay * ay), py = sqrtf(bx * bx + by * by);
	/* (a NaN length makes y the longer axis and N = 1) */
	int x_major = px >= py;
	float p_max = g_max(px, py), p_min = x_major ? py : px;
	float taps = ceilf(p_max / p_min);
	taps = g_min(g_min(taps, ORACLE_MAX_ANISOTROPY), g_max(ceilf(p_max), 1.0f));
	if (!(taps >= 1.0f)) taps = 1.0f;
	float rho = p_max / taps;
	float max_level = (float) (t->mip_count - 1);
	/* rho <= 1 (magnification), NaN and 0 all select the finest level */
	float lambda = (rho > 1.0f) ? g_min(o_log2(rho), max_level) : 0.0f;
	float level_0 = floorf(lambda);
	float fraction = lambda - level_0;
	uint32_t l0 = (uint32_t) level_0;
	uint32_t l1 = (l0 + 1 < t->mip_count) ? l0 + 1 : l0;
	uint32_t count = (uint32_t) taps;
	float du = x_major ? duv_dx[0] : duv_dy[0], dv = x_major ? duv_dx[1] : duv_dy[1];
	float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	for (uint32_t i = 0; i != count; ++i) {
		float u = uv[0], v = uv[1];
		if (count > 1) {
			float offset = (float) (i + 1) / (taps + 1.0f) - 0.5f;
			u = u + du * offset;
			v = v + dv * offset;
		}
		float c0[4], c1[4];
		sample_level(t, l0, u, v, c0);
		sample_level(t, l1, u, v, c1);
		for (int c = 0; c != 4; ++c) {
			float tap = c0[c] * (1.0f - fraction) + c1[c] * fraction;
			sum[c] = (count > 1) ? sum[c] + tap : tap;
		}
	}
	for (int c = 0; c != 4; ++c) out_rgba[c] = (count > 1) ? sum[c] / taps : sum[c];
}

void oracle_sample_light_texture(const oracle_light_texture_t* t, const float uv[2], float out_rgba[4]) {
	if (t->width == 0) {
		out_rgba[0] = out_rgba[1] = out_rgba[2] = out_rgba[3] = 1.0f;
		return;
	}
	float u = uv[0] - floorf(uv[0]), v = uv[1];
	if (!(u >= 0.0f && u <= 1.0f)) u = 0.0f;
	v = (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f;
	int w = (int) t->width, h = (int) t->height;
	float fx = u * (float) w - 0.5f, fy = v * (float) h - 0.5f;
	float flx = floorf(fx), fly = floorf(fy);
	float wx = fx - flx, wy = fy - fly;
	int ix = (int) flx, iy = (int) fly;
	int x0 = (ix < 0) ? w - 1 : ix, x1 = (ix + 1 >= w) ? 0 : ix + 1;
	int y0 = (iy < 0) ? 0 : iy, y1 = (iy + 1 >= h) ? h - 1 : iy + 1;
	const float* t00 = t->texels + 4 * ((size_t) y0 * w + x0), *t10 = t->texels + 4 * ((size_t) y0 * w + x1);
	const float* t01 = t->texels + 4 * ((size_t) y1 * w + x0), *t11 = t->texels + 4 * ((size_t) y1 * w + x1);
	for (int c = 0; c != 4; ++c) {
		float top = t00[c] * (1.0f - wx) + t10[c] * wx;
		float bottom = t01[c] * (1.0f - wx) + t11[c] * wx;
		out_rgba[c] = top * (1.0f - wy) + bottom * wy;
	}
}

static shading_data_t get_shading_data(const oracle_frame_t* f, const frame_constants_t* k, uint32_t primitive, v3 ray_direction) {
	shading_data_t r;
	v3 pos[3], nrm[3];
	v2 uv[3];
	for (int i = 0; i != 3; ++i) {
		size_t vi = (size_t) primitive * 3 + i;
		pos[i] = decode_position(f->quantized_positions[2 * vi], f->quantized_positions[2 * vi + 1], k->dequant_factor, k->dequant_summand);
		const uint16_t* q = f->normals_and_tex_coords + 4 * vi;
		nrm[i] = decode_normal((float) q[0] / 65535.0f, (float) q[1] / 65535.0f);
		uv[i] = mk2(fmaf((float) q[2] / 65535.0f, 8.0f, 0.0f), fmaf((float) q[3] / 65535.0f, -8.0f, 1.0f));
	}
	v3 origin = k->camera_position;
	v3 e0 = sub3(pos[1], pos[0]), e1 = sub3(pos[2], pos[0]);
	v3 ray_cross_e1 = cross3(ray_direction, e1);
	float rcp_det = 1.0f / dot3(e0, ray_cross_e1);
	v3 to0 = sub3(origin, pos[0]);
	float b[3];
	b[1] = rcp_det * dot3(to0, ray_cross_e1);
	v3 e0_cross_to0 = cross3(e0, to0);
	b[2] = -rcp_det * dot3(ray_direction, e0_cross_to0);
	b[0] = 1.0f - (b[1] + b[2]);
	r.position = fma3s(b[0], pos[0], fma3s(b[1], pos[1], scale3(pos[2], b[2])));
	v3 interpolated_normal = normalize3(fma3s(b[0], nrm[0], fma3s(b[1], nrm[1], scale3(nrm[2], b[2]))));
	uint32_t material = f->material_indices[primitive];
	const float* mc = f->material_constants + 8 * (size_t) material;
	float sampled[8];
	if (f->material_textures) {
		/* screen-space derivatives of the barycentrics (:754-766) and of the texture coordinate (:772-777) */
		v3 derivs[2];
		for (int i = 0; i != 2; ++i) {
			v3 ray_deriv = mk3(k->pixel_to_ray[0][i], k->pixel_to_ray[1][i], k->pixel_to_ray[2][i]);
			v3 ray_cross_e1_deriv = cross3(ray_deriv, e1);
			float rcp_det_deriv = -dot3(e0, ray_cross_e1_deriv) * rcp_det * rcp_det;
			float det_0_dir_e1 = dot3(to0, ray_cross_e1);
			float det_0_dir_e1_deriv = dot3(to0, ray_cross_e1_deriv);
			derivs[i].y = rcp_det_deriv * det_0_dir_e1 + rcp_det * det_0_dir_e1_deriv;
			float det_dir_e0_0 = dot3(ray_direction, e0_cross_to0);
			float det_dir_e0_0_deriv = dot3(ray_deriv, e0_cross_to0);
			derivs[i].z = -rcp_det_deriv * det_dir_e0_0 - rcp_det * det_dir_e0_0_deriv;
			derivs[i].x = -(derivs[i].y + derivs[i].z);
		}
		v2 tex_coord = fma2s(b[0], uv[0], fma2s(b[1], uv[1], scale2(uv[2], b[2])));
		v2 tex_derivs[2] = {mk2(0.0f, 0.0f), mk2(0.0f, 0.0f)};
		for (int i = 0; i != 2; ++i) {
			const float weights[3] = {derivs[i].x, derivs[i].y, derivs[i].z};
			for (int j = 0; j != 3; ++j) tex_derivs[i] = add2(tex_derivs[i], scale2(uv[j], weights[j]));
		}
		const float uv_[2] = {tex_coord.x, tex_coord.y}, dx_[2] = {tex_derivs[0].x, tex_derivs[0].y}, dy_[2] = {tex_derivs[1].x, tex_derivs[1].y};
		for (int type = 0; type != 3; ++type) {
			const oracle_texture_t* texture = &f->material_textures[3 * material + type];
			/* width 0: this texture is a constant (half / float *.vkt or absent file) */
			float texel[4] = {mc[3 * type], mc[3 * type + 1], type < 2 ? mc[3 * type + 2] : 0.0f, 1.0f};
			if (texture->width != 0) oracle_sample_texture(texture, uv_, dx_, dy_, texel);
			sampled[3 * type] = texel[0];
			sampled[3 * type + 1] = texel[1];
			if (type < 2) sampled[3 * type + 2] = texel[2];
		}
		mc = sampled;
	}
	v3 base_color = mk3(mc[0], mc[1], mc[2]);
	v3 specular_data = mk3(mc[3], mc[4], mc[5]);
	v3 nt;
	nt.x = fmaf(mc[6], 2.0f, -1.0f);
	nt.y = fmaf(mc[7], 2.0f, -1.0f);
	nt.z = sqrtf(g_max(0.0f, fmaf(-nt.x, nt.x, fmaf(-nt.y, nt.y, 1.0f))));
	float metalicity = specular_data.z;
	r.diffuse_albedo = mk3(fmaf(base_color.x, -metalicity, base_color.x), fmaf(base_color.y, -metalicity, base_color.y), fmaf(base_color.z, -metalicity, base_color.z));
	/* mix(vec3(0.02), base_color, metalicity) */
	r.fresnel_0 = mk3(0.02f * (1.0f - metalicity) + base_color.x * metalicity, 0.02f * (1.0f - metalicity) + base_color.y * metalicity, 0.02f * (1.0f - metalicity) + base_color.z * metalicity);
	float linear_roughness = specular_data.y;
	r.roughness = linear_roughness * linear_roughness;
	r.roughness = g_clamp(r.roughness * k->roughness_factor, 0.0064f, 1.0f);
	v2 uv_e0 = sub2(uv[1], uv[0]), uv_e1 = sub2(uv[2], uv[0]);
	v3 n_cross_e0 = cross3(interpolated_normal, e0);
	v3 e1_cross_n = cross3(e1, interpolated_normal);
	v3 tangent = add3(scale3(e1_cross_n, uv_e0.x), scale3(n_cross_e0, uv_e1.x));
	v3 bitangent = add3(scale3(e1_cross_n, uv_e0.y), scale3(n_cross_e0, uv_e1.y));
	float mean_tangent_length = sqrtf(0.5f * (dot3(tangent, tangent) + dot3(bitangent, bitangent)));
	m3 t2w;
	t2w.c[0] = tangent; t2w.c[1] = bitangent; t2w.c[2] = interpolated_normal;
	nt.z *= g_max(1.0e-10f, mean_tangent_length);
	r.normal = normalize3(m3_mul(&t2w, nt));
	r.outgoing = normalize3(sub3(k->camera_position, r.position));
	float normal_offset = g_max(0.0f, 1.0e-3f - dot3(r.normal, r.outgoing));
	r.normal = fma3s(normal_offset, r.outgoing, r.normal);
	r.normal = normalize3(r.normal);
	r.lambert_outgoing = dot3(r.normal, r.outgoing);
	return r;
}

/* ------------------------------------------------------------------------ */
/* ltc_utility.glsl                                                         */

typedef struct {
	m43 world_to_shading;
	m3 shading_to_cosine;
	m43 world_to_cosine;
	m3 cosine_to_shading;
	float albedo, determinant;
} ltc_t;

/* Bilinear, clamp-to-edge, nearest layer (sampler: ltc_table.c:170-177).  Weights
 * are exact fp32; x is filtered first, then y. */
static void ltc_fetch(const oracle_frame_t* f, float u, float v, float w, float out[6]) {
	int res = (int) f->ltc_resolution;
	float fx = u * (float) res - 0.5f, fy = v * (float) res - 0.5f;
	float flx = floorf(fx), fly = floorf(fy);
	float wx = fx - flx, wy = fy - fly;
	int x0 = (int) flx, y0 = (int) fly;
	int x1 = x0 + 1, y1 = y0 + 1;
	x0 = x0 < 0 ? 0 : (x0 > res - 1 ? res - 1 : x0);
	x1 = x1 < 0 ? 0 : (x1 > res - 1 ? res - 1 : x1);
	y0 = y0 < 0 ? 0 : (y0 > res - 1 ? res - 1 : y0);
	y1 = y1 < 0 ? 0 : (y1 > res - 1 ? res - 1 : y1);
	int layer = (int) rintf(w);
	int layers = (int) f->ltc_fresnel_count;
	layer = layer < 0 ? 0 : (layer > layers - 1 ? layers - 1 : layer);
	size_t base = (size_t) layer * res * res;
	for (int c = 0; c != 6; ++c) {
		const uint16_t* t = (c < 4) ? f->ltc_rgba : f->ltc_rg;
		int ch = (c < 4) ? c : c - 4, nch = (c < 4) ? 4 : 2;
		float t00 = (float) t[(base + (size_t) y0 * res + x0) * nch + ch] / 65535.0f;
		float t10 = (float) t[(base + (size_t) y0 * res + x1) * nch + ch] / 65535.0f;
		float t01 = (float) t[(base + (size_t) y1 * res + x0) * nch + ch] / 65535.0f;
		float t11 = (float) t[(base + (size_t) y1 * res + x1) * nch + ch] / 65535.0f;
		float top = t00 * (1.0f - wx) + t10 * wx;
		float bottom = t01 * (1.0f - wx) + t11 * wx;
		out[c] = top * (1.0f - wy) + bottom * wy;
	}
}

/* get_ltc_coefficients, ltc_utility.glsl:58-91 */
static ltc_t get_ltc_coefficients(const oracle_frame_t* f, const frame_constants_t* k, float fresnel_0, float roughness, v3 position, v3 normal, v3 outgoing) {
	ltc_t l;
	float n_dot_o = dot3(normal, outgoing);
	float inclination = o_acos_unit(g_clamp(n_dot_o, 0.0f, 1.0f));
	float u = fmaf(sqrtf(g_clamp(roughness, 0.0f, 1.0f)), k->ltc_roughness_factor, k->ltc_roughness_summand);
	float v = fmaf(inclination, k->ltc_inclination_factor, k->ltc_inclination_summand);
	float w = fmaf(g_clamp(fresnel_0, 0.0f, 1.0f), k->ltc_fresnel_factor, k->ltc_fresnel_summand);
	float d[6];
	ltc_fetch(f, u, v, w, d);
	/* mat3 constructor fills column by column */
	l.shading_to_cosine.c[0] = mk3(d[0], 0.0f, -d[1]);
	l.shading_to_cosine.c[1] = mk3(0.0f, d[2], 0.0f);
	l.shading_to_cosine.c[2] = mk3(d[3], 0.0f, d[4]);
	l.albedo = d[5];
	float det2 = d[0] * d[4] + d[1] * d[3];
	l.determinant = d[2] * det2;
	float inv_det2 = 1.0f / det2;
	l.cosine_to_shading.c[0] = mk3(d[4] * inv_det2, 0.0f, d[1] * inv_det2);
	l.cosine_to_shading.c[1] = mk3(0.0f, 1.0f / d[2], 0.0f);
	l.cosine_to_shading.c[2] = mk3(-d[3] * inv_det2, 0.0f, d[0] * inv_det2);
	v3 x_axis = normalize3(fma3s(-n_dot_o, normal, outgoing));
	v3 y_axis = cross3(normal, x_axis);
	/* rotation = transpose(mat3(x_axis, y_axis, normal)) */
	v3 r0 = mk3(x_axis.x, y_axis.x, normal.x);
	v3 r1 = mk3(x_axis.y, y_axis.y, normal.y);
	v3 r2 = mk3(x_axis.z, y_axis.z, normal.z);
	l.world_to_shading.c[0] = r0;
	l.world_to_shading.c[1] = r1;
	l.world_to_shading.c[2] = r2;
	m3 neg_rot;
	neg_rot.c[0] = neg3(r0); neg_rot.c[1] = neg3(r1); neg_rot.c[2] = neg3(r2);
	l.world_to_shading.c[3] = m3_mul(&neg_rot, position);
	for (int i = 0; i != 4; ++i)
		l.world_to_cosine.c[i] = m3_mul(&l.shading_to_cosine, l.world_to_shading.c[i]);
	return l;
}

/* evaluate_ltc_density, ltc_utility.glsl:103-108 */
static float evaluate_ltc_density(const ltc_t* l, v3 dir_shading, float rcp_psa) {
	v3 dc = m3_mul(&l->shading_to_cosine, dir_shading);
	float len_sq = dot3(dc, dc);
	float density = g_max(0.0f, dc.z) * l->determinant / (len_sq * len_sq);
	return density * rcp_psa;
}

/* ------------------------------------------------------------------------ */
/* polygon_clipping.glsl                                                    */

/* iz0, polygon_clipping.glsl:19-25 */
static v3 horizon_crossing(v3 a, v3 b) {
	float t = a.z / (a.z - b.z);
	return mk3(fmaf(t, b.x, fmaf(-t, a.x, a.x)), fmaf(t, b.y, fmaf(-t, a.y, a.y)), 0.0f);
}

/* clip_polygon, polygon_clipping.glsl:35-225.  The reference is a generated
 * switch over the sign mask; the generator's rule (verified case by case by
 * oracle/tools/check_clip_rule.py) is: walk the polygon from vertex 0, keep
 * vertices with z > 0, insert the horizon crossing of edge (i, i+1) at every
 * sign change, then rotate the result so that as many kept vertices as possible
 * stay at their original index (first such rotation wins), and repeat out[0]
 * at out[count] when there is room. */
static uint32_t clip_polygon(uint32_t vertex_count, uint32_t min_count, uint32_t cap, v3* v) {
	uint32_t n = vertex_count;
	int above[O_CAP];
	uint32_t kept = 0, changes = 0;
	for (uint32_t i = 0; i + 1 < cap; ++i) {
		int flag = (v[i].z > 0.0f) && (i < min_count || i < n);
		if (i < n) above[i] = flag;
		else if (flag) return 0; /* bit outside the polygon: no case in the switch */
	}
	for (uint32_t i = 0; i != n; ++i) {
		kept += above[i];
		changes += above[i] != above[(i + 1) % n];
	}
	if (kept == 0) return 0;
	if (kept == n) {
		if (n < cap) v[n] = v[0];
		return n;
	}
	if (changes != 2) return 0;
	/* tags: >= 0 original vertex index, -(1+i) crossing of edge (i, i+1) */
	int seq[O_CAP];
	uint32_t count = 0;
	for (uint32_t i = 0; i != n; ++i) {
		if (above[i]) seq[count++] = (int) i;
		if (above[i] != above[(i + 1) % n]) seq[count++] = -(int) (1 + i);
	}
	uint32_t best_rotation = 0, best_writes = 1000;
	for (uint32_t r = 0; r != count; ++r) {
		uint32_t writes = 0;
		for (uint32_t j = 0; j != count; ++j)
			writes += seq[(j + r) % count] != (int) j;
		writes += seq[r % count] != (int) count;
		if (writes < best_writes) { best_writes = writes; best_rotation = r; }
	}
	v3 out[O_CAP];
	for (uint32_t j = 0; j != count; ++j) {
		int tag = seq[(j + best_rotation) % count];
		if (tag >= 0) out[j] = v[tag];
		else {
			uint32_t i = (uint32_t) (-tag - 1);
			out[j] = horizon_crossing(v[i], v[(i + 1) % n]);
		}
	}
	for (uint32_t j = 0; j != count; ++j) v[j] = out[j];
	if (count < cap) v[count] = v[0];
	return count;
}

/* ------------------------------------------------------------------------ */
/* polygon_sampling.glsl                                                    */

/* fast_positive_atan, polygon_sampling.glsl:83-97 */
static float fast_positive_atan(float y) {
	float rx = (fabsf(y) > 1.0f) ? (1.0f / fabsf(y)) : fabsf(y);
	float ry = rx * rx;
	float rz = fmaf(ry, 0.02083509974181652f, -0.08513300120830536f);
	rz = fmaf(ry, rz, 0.18014100193977356f);
	rz = fmaf(ry, rz, -0.3302994966506958f);
	ry = fmaf(ry, rz, 0.9998660087585449f);
	rz = fmaf(-2.0f * ry, rx, O_HALF_PI);
	rz = (fabsf(y) > 1.0f) ? rz : 0.0f;
	rx = fmaf(rx, ry, rz);
	return (y < 0.0f) ? (O_PI - rx) : rx;
}

/* positive_atan, polygon_sampling.glsl:104-111 */
static float positive_atan(float tangent, int biased) {
	if (biased) return fast_positive_atan(tangent);
	float offset = (tangent < 0.0f) ? O_PI : 0.0f;
	return o_atan(tangent) + offset;
}

/* mix_fma, polygon_sampling.glsl:183-185 */
static float mix_fma(float x, float y, float a) { return fmaf(a, y, fmaf(-a, x, x)); }

/* kahan, polygon_sampling.glsl:261-268 */
static float kahan(float a, float b, float c, float d) {
	float cd = c * d;
	float error = fmaf(c, d, -cd);
	float result = fmaf(a, b, -cd);
	return result - error;
}

/* cross_stable, polygon_sampling.glsl:273-279 */
static v3 cross_stable(v3 l, v3 r) {
	return mk3(kahan(l.y, r.z, l.z, r.y), kahan(l.z, r.x, l.x, r.z), kahan(l.x, r.y, l.y, r.x));
}

/* is_inner_ellipse, polygon_sampling.glsl:292-299: the sign bit, so that -0 counts */
static int is_inner(v2 ellipse) { return (f2u(ellipse.x) & 0x80000000u) != 0; }

/* ellipse_from_edge, polygon_sampling.glsl:317-326 */
static v2 ellipse_from_edge(v3 a, v3 b) {
	v3 n = cross_stable(a, b);
	float scaling = 1.0f / n.z;
	scaling = is_inner(mk2(n.x, n.y)) ? -scaling : scaling;
	v2 e = mk2(n.x * scaling, n.y * scaling);
	e.x = (n.z != 0.0f) ? e.x : INFINITY;
	return e;
}

/* ellipse_transform, :332-334 */
static v2 ell_transform(v2 e, v2 p) { return fma2s(dot2(e, p), e, p); }
/* get_ellipse_det / rsqrt_det, :340-348 */
static float ell_det(v2 e) { return fmaf(e.x, e.x, fmaf(e.y, e.y, 1.0f)); }
static float ell_rsqrt_det(v2 e) { return rsqrt_f(ell_det(e)); }
/* get_ellipse_direction_factor_rsq, :351-355 */
static float ell_factor_rsq(v2 e, v2 d) {
	float ed = dot2(e, d);
	float dd = dot2(d, d);
	return fmaf(ed, ed, dd);
}
/* get_ellipse_direction_factor, :363-365 */
static float ell_factor(v2 e, v2 d) { return rsqrt_f(ell_factor_rsq(e, d)); }
/* get_ellipse_normalized_direction_factor, :369-372 */
static float ell_factor_unit(v2 e, v2 d) {
	float ed = dot2(e, d);
	return rsqrt_f(fmaf(ed, ed, 1.0f));
}

/* get_area_between_ellipses_in_sector_from_tangents, :377-382 */
/* positive_atan(n / d); the unbiased form goes through the fused evaluation of mode 1 (oracle_math.h) */
static float positive_atan_ratio(float n, float d, int biased) {
	if (biased) return fast_positive_atan(n / d);
	return o_positive_atan_ratio(n, d);
}

/* the tangents arrive as numerator / denominator pairs */
static float area_from_tangents(float in_rs, float in_n, float in_d, float out_rs, float out_n, float out_d, int biased) {
	float in_area = in_rs * positive_atan_ratio(in_n, in_d, biased);
	float r = fmaf(out_rs, positive_atan_ratio(out_n, out_d, biased), -in_area);
	return (r > 0.0f) ? (0.5f * r) : 0.0f;
}

/* get_area_between_ellipses_in_sector, :390-397 */
static float area_between(v2 ein, float in_rs, v2 eout, float out_rs, v2 d0, v2 d1, int biased) {
	float det_dirs = positive_part(dot2(d1, rot90(d0)));
	float in_dot = in_rs * dot2(d0, ell_transform(ein, d1));
	float out_dot = out_rs * dot2(d0, ell_transform(eout, d1));
	return area_from_tangents(in_rs, det_dirs, in_dot, out_rs, det_dirs, out_dot, biased);
}

/* get_ellipse_area_in_sector, :405-412 */
static float area_in_sector(v2 e, v2 d0, v2 d1, int biased) {
	float rs = ell_rsqrt_det(e);
	float det_dirs = positive_part(dot2(d1, rot90(d0)));
	float ed = rs * dot2(d0, ell_transform(e, d1));
	float area = 0.5f * rs * positive_atan_ratio(det_dirs, ed, biased);
	return (rs > 0.0f) ? area : 0.0f;
}

typedef struct {
	uint32_t vertex_count;
	v2 vertices[O_CAP];
	v2 ellipses[O_CAP];
	v2 inner_ellipse_0;
	float sector_psa[O_CAP];
	float psa;
} psa_polygon_t;

/* compare_and_swap, :421-435 */
static void compare_and_swap(psa_polygon_t* p, uint32_t l, uint32_t r) {
	v2 lv = p->vertices[l], rv = p->vertices[r];
	float normal_z = kahan(lv.x, -rv.y, lv.y, -rv.x);
	int swap = (normal_z == 0.0f) ? (isinf(p->ellipses[r].x) != 0) : (normal_z > 0.0f);
	if (swap) {
		p->vertices[l] = rv; p->vertices[r] = lv;
		v2 le = p->ellipses[l];
		p->ellipses[l] = p->ellipses[r]; p->ellipses[r] = le;
	}
}

/* sort_convex_polygon_vertices, :440-505: one network per vertex count plus a shared tail */
static void sort_vertices(psa_polygon_t* p) {
	static const uint8_t net5[] = {2,4, 1,3, 1,2, 0,3, 3,4};
	static const uint8_t net6[] = {3,5, 2,4, 1,5, 0,4, 4,5, 1,3};
	static const uint8_t net7[] = {2,5, 1,6, 5,6, 3,4, 0,4, 4,6, 1,3, 3,5, 4,5};
	static const uint8_t net8[] = {2,6, 3,7, 1,5, 0,4, 4,6, 5,7, 6,7, 4,5, 1,3};
	const uint8_t* net = NULL;
	uint32_t pairs = 0;
	switch (p->vertex_count) {
	case 3: compare_and_swap(p, 1, 2); break;
	case 4: compare_and_swap(p, 1, 3); break;
	case 5: net = net5; pairs = 5; break;
	case 6: net = net6; pairs = 6; break;
	case 7: net = net7; pairs = 9; break;
	case 8: net = net8; pairs = 9; break;
	default: break;
	}
	for (uint32_t i = 0; i != pairs; ++i) compare_and_swap(p, net[2 * i], net[2 * i + 1]);
	compare_and_swap(p, 0, 2);
	if (p->vertex_count >= 4) compare_and_swap(p, 2, 3);
	compare_and_swap(p, 0, 1);
}

/* prepare_projected_solid_angle_polygon_sampling, :521-589 */
static psa_polygon_t prepare_psa(uint32_t vertex_count, uint32_t cap, const v3* verts, int biased) {
	psa_polygon_t p;
	memset(&p, 0, sizeof(p));
	p.vertex_count = vertex_count;
	p.inner_ellipse_0 = mk2(1.0f, 0.0f);
	p.vertices[0] = mk2(verts[0].x, verts[0].y);
	p.ellipses[0] = ellipse_from_edge(verts[0], verts[1]);
	v2 previous = p.ellipses[0];
	for (uint32_t i = 1; i != cap; ++i) {
		p.vertices[i] = mk2(verts[i].x, verts[i].y);
		if (i > 2 && i == vertex_count) break;
		v2 e = ellipse_from_edge(verts[i], verts[(i + 1) % cap]);
		int e_inner = is_inner(e);
		p.ellipses[i] = e_inner ? previous : e;
		if (is_inner(previous) && !e_inner) p.inner_ellipse_0 = previous;
		previous = e;
	}
	{
		v2 e = p.ellipses[0];
		int e_inner = is_inner(e);
		p.ellipses[0] = e_inner ? previous : e;
		if (is_inner(previous) && !e_inner) p.inner_ellipse_0 = previous;
	}
	p.psa = 0.0f;
	if (p.inner_ellipse_0.x > 0.0f) {
		for (uint32_t i = 0; i != cap; ++i) {
			if (i > 2 && i == vertex_count) break;
			p.sector_psa[i] = area_in_sector(p.ellipses[i], p.vertices[i], p.vertices[(i + 1) % cap], biased);
			p.psa += p.sector_psa[i];
		}
	}
	else {
		sort_vertices(&p);
		v2 ein = p.inner_ellipse_0, eout = mk2(0.0f, 0.0f);
		float in_rs = ell_rsqrt_det(ein), out_rs = 0.0f;
		for (uint32_t i = 0; i + 1 != cap; ++i) {
			if (i > 1 && i + 1 == vertex_count) break;
			v2 ve = p.ellipses[i];
			int v_inner = is_inner(ve);
			float v_rs = ell_rsqrt_det(ve);
			if (i == 0) { eout = ve; out_rs = v_rs; }
			else if (v_inner) { ein = ve; in_rs = v_rs; }
			else { eout = ve; out_rs = v_rs; }
			p.sector_psa[i] = area_between(ein, in_rs, eout, out_rs, p.vertices[i], p.vertices[i + 1], biased);
			p.psa += p.sector_psa[i];
		}
	}
	return p;
}

/* normalize_approx_and_flip, :599-611 */
static v2 normalize_approx_and_flip(v2 rhs, v2 semi_circle) {
	float scaling = fabsf(rhs.x) + fabsf(rhs.y);
	scaling = u2f(f2u(scaling) ^ 0x7F800000u);
	scaling = (dot2(rhs, semi_circle) >= 0.0f) ? scaling : -scaling;
	return mk2(scaling * rhs.x, scaling * rhs.y);
}

/* solve_homogeneous_quadratic, :625-630, with the four entries passed as
 * q[column][row] */
static v2 solve_quadratic(float q00, float q01, float q10, float q11) {
	float cxy = 0.5f * (q01 + q10);
	float sd = sqrtf(g_max(0.0f, cxy * cxy - q00 * q11));
	float root = fabsf(cxy) + sd;
	return (cxy >= 0.0f) ? mk2(root, -q00) : mk2(q11, root);
}

/* sample_sector_between_ellipses, :645-739 */
static v2 sample_between_ellipses(v2 u, float target_area, v2 ein, v2 eout, v2 d0, v2 d1, uint32_t iterations, int biased) {
	v2 q0 = normalize2(d0), q2 = normalize2(d1);
	v2 q1 = add2(q0, q2);
	float fin0 = ell_factor_unit(ein, q0), fin1 = ell_factor(ein, q1), fin2 = ell_factor_unit(ein, q2);
	float fout0 = ell_factor_unit(eout, q0), fout1 = ell_factor(eout, q1), fout2 = ell_factor_unit(eout, q2);
	float area0 = fout0 * fout1 - fin0 * fin1;
	float area1 = fout1 * fout2 - fin1 * fin2;
	float tq = mix_fma(-area0, area1, u.x);
	int first = tq <= 0.0f;
	if (first) { q2 = q0; fin2 = fin0; fout2 = fout0; }
	tq += first ? area0 : -area1;
	tq *= fabsf(q1.x * q2.y - q1.y * q2.x);
	v2 nin = add2(scale2(q1, fin1), scale2(q2, fin2));
	v2 nout = add2(scale2(q1, fout1), scale2(q2, fout2));
	nin = ell_transform(ein, nin);
	nout = ell_transform(eout, nout);
	float off_in = dot2(nin, q1) * fin1;
	float off_out = dot2(nout, q1) * fout1;
	v2 r2 = rot90(q2);
	/* quadratic = outer(a, nin) - outer(b, nout); outer(c, r)[col][row] = c[row] * r[col] */
	v2 a = scale2(r2, off_out * fout2);
	v2 b = add2(scale2(r2, off_in * fin2), scale2(nin, tq));
	v2 cur = solve_quadratic(a.x * nin.x - b.x * nout.x, a.y * nin.x - b.y * nout.x, a.x * nin.y - b.x * nout.y, a.y * nin.y - b.y * nout.y);
	if (!biased) {
		const float acceptable_error = 1.0e-5f;
		uint32_t count = (fabsf(u.x - 0.5f) <= 0.5f - acceptable_error) ? iterations : 0;
		float in_rs = ell_rsqrt_det(ein), out_rs = ell_rsqrt_det(eout);
		for (uint32_t i = 0; i != count; ++i) {
			cur = normalize_approx_and_flip(cur, q1);
			v2 ind = ell_transform(ein, cur), outd = ell_transform(eout, cur);
			float det_dirs = positive_part(dot2(cur, rot90(q0)));
			float error = target_area - area_from_tangents(in_rs, det_dirs, in_rs * dot2(q0, ind), out_rs, det_dirs, out_rs * dot2(q0, outd), biased);
			v2 c = sub2(ind, outd), rc = rot90(cur);
			v2 c2 = scale2(ind, 2.0f * error);
			cur = solve_quadratic(c.x * rc.x - c2.x * outd.x, c.y * rc.x - c2.y * outd.x, c.x * rc.y - c2.x * outd.y, c.y * rc.y - c2.y * outd.y);
		}
	}
	if (!(dot2(cur, q1) >= 0.0f)) cur = neg2(cur);
	float in_factor = 1.0f / ell_factor_rsq(ein, cur);
	float out_factor = 1.0f / ell_factor_rsq(eout, cur);
	return scale2(cur, sqrtf(mix_fma(in_factor, out_factor, u.y)));
}

/* sample_projected_solid_angle_polygon, :749-805 */
static v3 sample_psa(const psa_polygon_t* p, uint32_t cap, v2 u, int biased) {
	float target = u.x * p->psa;
	v2 xy, eout = mk2(0.0f, 0.0f), d0 = mk2(0.0f, 0.0f);
	if (p->inner_ellipse_0.x > 0.0f) {
		for (uint32_t i = 0; i != cap; ++i) {
			if (i > 0) target -= p->sector_psa[i - 1];
			eout = p->ellipses[i];
			d0 = p->vertices[i];
			if ((i >= 2 && i + 1 == p->vertex_count) || target < p->sector_psa[i]) break;
		}
		float sqrt_det = sqrtf(ell_det(eout));
		float angle = 2.0f * target * sqrt_det;
		float sn, cs;
		o_sincos(angle, &sn, &cs);
		v2 t = rot90(ell_transform(eout, d0));
		float cf = cs * sqrt_det;
		xy = mk2(cf * d0.x + sn * t.x, cf * d0.y + sn * t.y);
		xy = scale2(xy, sqrtf(u.y / ell_factor_rsq(eout, xy)));
	}
	else {
		float sector = 0.0f;
		v2 ein = p->inner_ellipse_0, d1 = mk2(0.0f, 0.0f);
		for (uint32_t i = 0; i + 1 != cap; ++i) {
			v2 ve = p->ellipses[i];
			if (i == 0) eout = ve;
			else {
				target -= p->sector_psa[i - 1];
				if (is_inner(ve)) ein = ve; else eout = ve;
			}
			d0 = p->vertices[i];
			d1 = p->vertices[i + 1];
			sector = p->sector_psa[i];
			if ((i >= 1 && i + 2 == p->vertex_count) || target < sector) break;
		}
		u.x = target / sector;
		xy = sample_between_ellipses(u, target, ein, eout, d0, d1, 2, biased);
	}
	float z = sqrtf(g_max(0.0f, fmaf(-xy.x, xy.x, fmaf(-xy.y, xy.y, 1.0f))));
	return mk3(xy.x, xy.y, z);
}

/* compute_projected_solid_angle_polygon_sampling_error, :823-883 */
static v3 psa_sampling_error(const psa_polygon_t* p, uint32_t cap, v2 u, v3 dir) {
	float target = u.x * p->psa;
	if (p->inner_ellipse_0.x > 0.0f) return mk3(0.0f, 0.0f, 0.0f);
	float sector = 0.0f;
	v2 eout = mk2(0.0f, 0.0f), ein = p->inner_ellipse_0, d0 = mk2(0.0f, 0.0f);
	for (uint32_t i = 0; i + 1 != cap; ++i) {
		if ((i > 1 && i + 1 == p->vertex_count) || (i > 0 && target < 0.0f)) break;
		sector = p->sector_psa[i];
		target -= sector;
		v2 ve = p->ellipses[i];
		if (i == 0) eout = ve;
		else if (is_inner(ve)) ein = ve;
		else eout = ve;
		d0 = p->vertices[i];
	}
	target += sector;
	v2 sxy = mk2(dir.x, dir.y);
	float sampled = area_between(ein, ell_rsqrt_det(ein), eout, ell_rsqrt_det(eout), d0, sxy, 0);
	float scaled_backward = target - sampled;
	float backward = scaled_backward / p->psa;
	v2 ind = ell_transform(ein, sxy), outd = ell_transform(eout, sxy);
	float inf = 1.0f / dot2(sxy, ind), outf = 1.0f / dot2(sxy, outd);
	/* constraint matrix columns, then transposed */
	v2 c0 = scale2(rot90(sxy), 0.5f * (inf - outf));
	v2 c1 = scale2(ind, (1.0f - u.y) / (inf * inf));
	c1 = add2(c1, scale2(outd, u.y / (outf * outf)));
	/* after transpose: m[0] = (c0.x, c1.x), m[1] = (c0.y, c1.y) */
	float m00 = c0.x, m01 = c1.x, m10 = c0.y, m11 = c1.y;
	float det = m00 * m11 - m01 * m10;
	float inv = 1.0f / det;
	v3 deriv;
	deriv.x = inv * m11;
	deriv.y = inv * -m01;
	deriv.z = -(sxy.x * deriv.x + sxy.y * deriv.y) / dir.z;
	float forward = sqrtf(dot3(deriv, deriv)) * scaled_backward;
	return mk3(backward, scaled_backward, forward);
}

/* ---- solid angle sampling (polygon_sampling.glsl:61-224) ------------------ */

typedef struct {
	uint32_t vertex_count;
	v3 dirs[O_CAP];
	v3 params[O_CAP];
	float fan[O_CAP];
	float solid_angle;
} sa_polygon_t;

/* prepare_solid_angle_polygon_sampling, :120-175 */
static sa_polygon_t prepare_sa(uint32_t vertex_count, uint32_t cap, const v3* verts, v3 shading_position) {
	sa_polygon_t p;
	memset(&p, 0, sizeof(p));
	p.vertex_count = vertex_count;
	for (uint32_t i = 0; i != cap; ++i) p.dirs[i] = normalize3(sub3(verts[i], shading_position));
	float h_sign = (p.dirs[0].x > 0.0f) ? -1.0f : 1.0f;
	float h_scale = 1.0f / (fabsf(p.dirs[0].x) + 1.0f);
	v2 h_yz = mk2(p.dirs[0].y * h_scale, p.dirs[0].z * h_scale);
	p.solid_angle = 0.0f;
	float prev_dot_1_2 = dot3(p.dirs[0], p.dirs[1]);
	for (uint32_t i = 0; i + 2 != cap; ++i) {
		if (i >= 1 && i + 2 >= vertex_count) break;
		v3 t0 = p.dirs[i + 1], t1 = p.dirs[0], t2 = p.dirs[i + 2];
		float d01 = prev_dot_1_2;
		float d02 = dot3(t0, t2);
		float d12 = dot3(t1, t2);
		prev_dot_1_2 = d12;
		float dh0 = fmaf(-h_sign, t0.x, d01);
		float dh2 = fmaf(-h_sign, t2.x, d12);
		v2 c0 = mk2(fmaf(-dh0, h_yz.x, t0.y), fmaf(-dh0, h_yz.y, t0.z));
		v2 c1 = mk2(fmaf(-dh2, h_yz.x, t2.y), fmaf(-dh2, h_yz.y, t2.z));
		float volume = fabsf(c0.x * c1.y - c0.y * c1.x);
		float d02_plus_12 = d02 + d12;
		float one_plus_01 = 1.0f + d01;
		float tangent = volume / (one_plus_01 + d02_plus_12);
		float tri = 2.0f * positive_atan(tangent, 0);
		p.solid_angle += tri;
		p.fan[i] = p.solid_angle;
		p.params[i] = mk3(volume, d02_plus_12, one_plus_01);
	}
	return p;
}

/* sample_solid_angle_polygon, :194-224 */
static v3 sample_sa(const sa_polygon_t* p, uint32_t cap, v2 u) {
	float target = p->solid_angle * u.x;
	float sub = target;
	v3 prm = p->params[0];
	v3 t0 = p->dirs[1], t1 = p->dirs[0], t2 = p->dirs[2];
	for (uint32_t i = 0; i + 3 < cap; ++i) {
		if (i + 3 >= p->vertex_count || p->fan[i] >= target) break;
		sub = target - p->fan[i];
		t0 = p->dirs[i + 2];
		t2 = p->dirs[i + 3];
		prm = p->params[i + 1];
	}
	float sn, cs;
	o_sincos(0.5f * sub, &sn, &cs);
	float w0 = prm.x * cs - prm.y * sn;
	float w2 = prm.z * sn;
	v3 offset = add3(scale3(t0, w0), scale3(t2, w2));
	float f = 2.0f * (dot3(t0, offset) / dot3(offset, offset));
	v3 new2 = fma3s(f, offset, neg3(t0));
	float s2 = dot3(t1, new2);
	float s = mix_fma(1.0f, s2, u.y);
	float denominator = fmaf(-s2, s2, 1.0f);
	float t_normed = sqrtf(fmaf(-s, s, 1.0f) / denominator);
	t_normed = (denominator > 0.0f) ? t_normed : u.y;
	return add3(scale3(t1, fmaf(-t_normed, s2, s)), scale3(new2, t_normed));
}

/* ------------------------------------------------------------------------ */
/* shading_pass.frag.glsl                                                   */

typedef struct {
	const oracle_frame_t* f;
	const frame_constants_t* k;
	uint64_t rays;
} pixel_ctx_t;

static float dot4_point(v3 p, v4 plane) { return ((p.x * plane.x + p.y * plane.y) + p.z * plane.z) + 1.0f * plane.w; }
static v3 plane_normal(v4 plane) { return mk3(plane.x, plane.y, plane.z); }

/* get_polygon_visibility, shading_pass.frag.glsl:120-138 */
static int polygon_visibility(pixel_ctx_t* ctx, int visibility, v3 dir, v3 position, const light_view_t* light) {
	if (!ctx->f->trace_shadow_rays || !visibility) return visibility;
	float max_t = -dot4_point(position, light->plane) / dot3(dir, plane_normal(light->plane));
	float o[3] = {position.x, position.y, position.z}, d[3] = {dir.x, dir.y, dir.z};
	++ctx->rays;
	int blocked = oracle_bvh_any_hit(ctx->f->bvh, o, d, 1.0e-3f, max_t, ctx->f->brute_force_rays);
	/* (diagnostics: ORACLE_LOG_RAYS=1 prints every shadow ray - used with one-pixel calls to explain a pixel) */
	if (g_log_rays) printf("ray o %.9g %.9g %.9g d %.9g %.9g %.9g tmax %.9g plane %.9g %.9g %.9g %.9g blocked %d\n", o[0], o[1], o[2], d[0], d[1], d[2], max_t, light->plane.x, light->plane.y, light->plane.z, light->plane.w, blocked);
	return !blocked;
}

/* get_polygon_radiance, :151-185 */
static v3 polygon_radiance(const oracle_frame_t* f, v3 dir, v3 position, const light_view_t* light) {
	v3 radiance = light->surface_radiance;
	uint32_t technique = light->texturing_technique;
	if (technique != 0) {
		float uv[2];
		if (technique == 1) {
			/* polygon_texturing_area: plane-space coordinates of the hit point */
			float t = -dot4_point(position, light->plane) / dot3(dir, plane_normal(light->plane));
			v3 x = sub3(add3(position, scale3(dir, t)), light->translation);
			uv[0] = dot3(light->rotation_columns[0], x) * light->inv_scaling_x;
			uv[1] = dot3(light->rotation_columns[1], x) * light->inv_scaling_y;
		}
		else {
			v3 lookup;
			if (technique == 3) {
				/* polygon_texturing_ies_profile: plane space; the profile holds the cosine already */
				lookup = mk3(dot3(light->rotation_columns[0], dir), dot3(light->rotation_columns[1], dir), dot3(light->rotation_columns[2], dir));
				radiance = scale3(radiance, 1.0f / fabsf(lookup.z));
			}
			else
				lookup = mk3(-dir.x, dir.y, dir.z);
			uv[0] = o_atan2(lookup.y, lookup.x) * (0.5f * O_INV_PI);
			uv[1] = o_acos(lookup.z) * O_INV_PI;
		}
		float texel[4] = {1.0f, 1.0f, 1.0f, 1.0f};
		if (f->light_textures && light->texture_index < f->light_texture_count)
			oracle_sample_light_texture(&f->light_textures[light->texture_index], uv, texel);
		radiance = mul3(radiance, mk3(texel[0], texel[1], texel[2]));
	}
	return radiance;
}

/* get_polygon_radiance_visibility_brdf_product, :203-231 */
static v3 radiance_visibility_brdf(pixel_ctx_t* ctx, float* out_lambert, int* out_visibility, v3 dir, const shading_data_t* sd, const light_view_t* light, int diffuse, int specular) {
	float lambert = dot3(sd->normal, dir);
	int visibility = lambert > 0.0f;
	visibility = polygon_visibility(ctx, visibility, dir, sd->position, light);
	if (out_lambert) *out_lambert = lambert;
	if (out_visibility) *out_visibility = visibility;
	if (visibility) return mul3(polygon_radiance(ctx->f, dir, sd->position, light), evaluate_brdf(sd, dir, diffuse, specular));
	return mk3(0.0f, 0.0f, 0.0f);
}

/* get_mis_weight_over_density, :243-252 */
static float mis_weight_over_density(int heuristic, float sampled, float other) {
	if (heuristic == O_MIS_BALANCE) return 1.0f / (sampled + other);
	if (heuristic == O_MIS_POWER) return sampled / (sampled * sampled + other * other);
	return 0.0f;
}

/* get_mis_estimate, :270-293 (per colour channel) */
static v3 mis_estimate(int heuristic, v3 integrand, v3 sw, float sd, v3 ow, float od, float ve) {
	float in[3] = {integrand.x, integrand.y, integrand.z}, s[3] = {sw.x, sw.y, sw.z}, o[3] = {ow.x, ow.y, ow.z}, r[3];
	for (int c = 0; c != 3; ++c) {
		if (heuristic == O_MIS_WEIGHTED) {
			float weighted_sum = s[c] * sd + o[c] * od;
			r[c] = (s[c] * in[c]) / weighted_sum;
		}
		else if (heuristic == O_MIS_OPTIMAL_CLAMPED || heuristic == O_MIS_OPTIMAL) {
			float balance = 1.0f / (sd + od);
			float weighted_sum = s[c] * sd + o[c] * od;
			if (heuristic == O_MIS_OPTIMAL_CLAMPED) {
				float weighted = s[c] / weighted_sum;
				float mixed = fmaf(-ve, balance, balance);
				mixed = fmaf(ve, weighted, mixed);
				r[c] = mixed * in[c];
			}
			else
				r[c] = ve * s[c] + balance * (in[c] - ve * weighted_sum);
		}
		else
			r[c] = mis_weight_over_density(heuristic, sd, od) * in[c];
	}
	return mk3(r[0], r[1], r[2]);
}

/* get_polygonal_light_mis_estimate, :305-323 */
static v3 light_mis_estimate(pixel_ctx_t* ctx, v3 dir, float density, const shading_data_t* sd, const light_view_t* light) {
	float lambert;
	v3 rb = radiance_visibility_brdf(ctx, &lambert, NULL, dir, sd, light, 1, 1);
	int strategy = ctx->f->sampling_strategies;
	if (strategy == O_STRATEGY_DIFFUSE_ONLY)
		return (density > 0.0f) ? scale3(rb, lambert / density) : mk3(0.0f, 0.0f, 0.0f);
	if (strategy == O_STRATEGY_DIFFUSE_GGX_MIS) {
		float ggx_density = ggx_reflected_density(sd->lambert_outgoing, sd->outgoing, dir, sd->normal, sd->roughness);
		return scale3(scale3(rb, lambert), mis_weight_over_density(ctx->f->mis_heuristic, density, ggx_density));
	}
	return mk3(0.0f, 0.0f, 0.0f);
}

/* polygonal_light_ray_intersection, polygonal_light_utility.glsl:93-112 */
static int light_ray_intersection(const light_view_t* light, uint32_t vmax, v3 origin, v3 end_xyz, float end_w) {
	float side_a = dot4_point(origin, light->plane);
	float side_b = ((light->plane.x * end_xyz.x + light->plane.y * end_xyz.y) + light->plane.z * end_xyz.z) + light->plane.w * end_w;
	if (side_a * side_b > 0.0f) return 0;
	v3 dir = sub3(end_xyz, scale3(origin, end_w));
	float previous_sign = 0.0f;
	int result = 1;
	for (uint32_t i = 0; i != vmax; ++i) {
		v3 a = sub3(light->vertices_world[i], origin);
		v3 b = sub3(light->vertices_world[(i + 1) % vmax], origin);
		/* determinant(mat3(dir, a, b)) = dot(dir, cross(a, b)) expanded along the first column */
		float sign = dir.x * (a.y * b.z - b.y * a.z) - a.x * (dir.y * b.z - b.y * dir.z) + b.x * (dir.y * a.z - a.y * dir.z);
		result = result && ((i >= 3 && i >= light->vertex_count) || previous_sign * sign >= 0.0f);
		previous_sign = sign;
	}
	return result;
}

/* error_to_color, shading_pass.frag.glsl:80-114: tab20b colours, five decades */
static v3 error_to_color(const frame_constants_t* k, float error) {
	static const float colors[20][3] = {
		{0.04092f, 0.04374f, 0.19120f}, {0.08438f, 0.08866f, 0.36625f}, {0.14703f, 0.15593f, 0.62396f}, {0.33245f, 0.34191f, 0.73046f},
		{0.12477f, 0.19120f, 0.04092f}, {0.26225f, 0.36131f, 0.08438f}, {0.46208f, 0.62396f, 0.14703f}, {0.61721f, 0.70838f, 0.33245f},
		{0.26225f, 0.15293f, 0.03071f}, {0.50888f, 0.34191f, 0.04092f}, {0.79910f, 0.49102f, 0.08438f}, {0.79910f, 0.59720f, 0.29614f},
		{0.23074f, 0.04519f, 0.04092f}, {0.41789f, 0.06663f, 0.06848f}, {0.67244f, 0.11954f, 0.14703f}, {0.79910f, 0.30499f, 0.33245f},
		{0.19807f, 0.05286f, 0.17144f}, {0.37626f, 0.08228f, 0.29614f}, {0.61721f, 0.15293f, 0.50888f}, {0.73046f, 0.34191f, 0.67244f}};
	const float min_exponent = 0.0f, max_exponent = 5.0f, color_count = 20.0f;
	const float min_error = l_powf(10.0f, min_exponent);
	const float max_error = l_powf(10.0f, max_exponent - 0.01f);
	error = g_min(g_max(fabsf(k->error_factor * error), min_error), max_error);
	/* A NaN error (degenerate sample) passes through the clamp; in the reference the
	 * colour index is then int(NaN), which is undefined.  Defined here: first colour. */
	error = (error == error) ? error : min_error;
	float color_index = fmaf(o_log2(error), color_count / ((max_exponent - min_exponent) * l_log2f(10.0f)), color_count * -min_exponent / (max_exponent - min_exponent));
	int index = (int) color_index;
	return mk3(colors[index][0], colors[index][1], colors[index][2]);
}

/* the ERROR_DISPLAY_* branches, shading_pass.frag.glsl:489-494, :549-563 */
static v3 display_sampling_error(pixel_ctx_t* ctx, const psa_polygon_t* polygon, uint32_t cap, noise_accessor_t* noise, int biased) {
	const oracle_frame_t* f = ctx->f;
	v2 u = next_noise_2(f, ctx->k, noise);
	v3 dir = sample_psa(polygon, cap, u, biased);
	v3 e = psa_sampling_error(polygon, cap, u, dir);
	float error = (f->error_index == 0) ? e.x : ((f->error_index == 1) ? e.y : e.z);
	v3 color = error_to_color(ctx->k, error);
	return mk3(color.x / ctx->k->exposure_factor, color.y / ctx->k->exposure_factor, color.z / ctx->k->exposure_factor);
}

/* ---- related work: Arvo's projected solid angle sampling
 * (polygon_sampling_related_work.glsl:509-1048) ---------------------------------------------- */

typedef struct {
	float cdf_factor;
	v2 length_coeffs, elevations;
} edge_arvo_t;

typedef struct {
	uint32_t vertex_count;
	float azimuths[O_CAP];
	edge_arvo_t edges[O_CAP];
	edge_arvo_t inner_edge_0;
	float sector_psa[O_CAP];
	float psa;
} psa_arvo_t;

/* prepare_edge_arvo, :559-578 */
static edge_arvo_t prepare_edge_arvo(v3 vertex_0, v3 vertex_1) {
	edge_arvo_t edge;
	v3 normal_a = normalize3(cross3(vertex_0, vertex_1));
	edge.cdf_factor = 0.5f * normal_a.z;
	v3 ccw_vertex = (edge.cdf_factor > 0.0f) ? vertex_0 : vertex_1;
	v2 normal_c = rot90(normalize2(mk2(ccw_vertex.x, ccw_vertex.y)));
	float cos_beta = -dot2(mk2(normal_a.x, normal_a.y), normal_c);
	float sin_beta_sq = fmaf(-cos_beta, cos_beta, 1.0f);
	float csc_beta = rsqrt_f(g_max(0.0f, sin_beta_sq));
	float csc_c = rsqrt_f(g_max(0.0f, fmaf(-ccw_vertex.z, ccw_vertex.z, 1.0f)));
	edge.length_coeffs.x = sin_beta_sq;
	edge.length_coeffs.y = dot2(mk2(normal_a.x, normal_a.y), rot90(normal_c)) * cos_beta;
	edge.length_coeffs = scale2(edge.length_coeffs, csc_beta * csc_c);
	edge.elevations.x = ccw_vertex.z;
	edge.elevations.y = cross3(ccw_vertex, normal_a).z;
	edge.elevations.y = (edge.cdf_factor > 0.0f) ? -edge.elevations.y : edge.elevations.y;
	return edge;
}

/* get_edge_projected_solid_angle_in_sector_arvo, :599-609 */
static float edge_psa_in_sector_arvo(const edge_arvo_t* edge, float relative_azimuth_0, float relative_azimuth_1) {
	float s0, c0, s1, c1;
	o_sincos(relative_azimuth_0, &s0, &c0);
	o_sincos(relative_azimuth_1, &s1, &c1);
	v2 point_0 = mk2(dot2(edge->length_coeffs, mk2(c0, s0)), s0);
	v2 point_1 = mk2(dot2(edge->length_coeffs, mk2(c1, s1)), s1);
	v2 rotated = mk2(point_0.x * point_1.x + point_0.y * point_1.y, point_0.x * point_1.y - point_0.y * point_1.x);
	float length = positive_atan(fabsf(rotated.y) / rotated.x, 0);
	return edge->cdf_factor * length;
}

/* get_edge_projected_solid_angle_in_sector_derivative_arvo, :618-640 */
static v2 edge_psa_in_sector_derivative_arvo(const edge_arvo_t* edge, float relative_azimuth_0, float relative_azimuth_1) {
	float s0, c0, s1, c1;
	o_sincos(relative_azimuth_0, &s0, &c0);
	o_sincos(relative_azimuth_1, &s1, &c1);
	v2 point_0 = mk2(dot2(edge->length_coeffs, mk2(c0, s0)), s0);
	v2 dir_1 = mk2(c1, s1);
	v2 point_1 = mk2(dot2(edge->length_coeffs, dir_1), dir_1.y);
	v2 rotated = mk2(point_0.x * point_1.x + point_0.y * point_1.y, point_0.x * point_1.y - point_0.y * point_1.x);
	float quotient = fabsf(rotated.y) / rotated.x;
	float length = positive_atan(quotient, 0);
	v2 dir_1_deriv = rot90(dir_1);
	v2 point_1_deriv = mk2(dot2(edge->length_coeffs, dir_1_deriv), dir_1_deriv.y);
	v2 rotated_deriv = mk2(point_0.x * point_1_deriv.x + point_0.y * point_1_deriv.y, point_0.x * point_1_deriv.y - point_0.y * point_1_deriv.x);
	float quotient_derivative = (rotated_deriv.y * rotated.x - rotated.y * rotated_deriv.x) / (rotated.x * rotated.x);
	quotient_derivative = (rotated.y < 0.0f) ? (-quotient_derivative) : quotient_derivative;
	float length_deriv = quotient_derivative / fmaf(quotient, quotient, 1.0f);
	return mk2(edge->cdf_factor * length, edge->cdf_factor * length_deriv);
}

/* get_edge_elevation_arvo, :648-653 */
static float edge_elevation_arvo(const edge_arvo_t* edge, float relative_azimuth) {
	float sn, cs;
	o_sincos(relative_azimuth, &sn, &cs);
	v2 point = normalize2(mk2(dot2(edge->length_coeffs, mk2(cs, sn)), sn));
	return dot2(point, edge->elevations);
}

/* compare_and_swap_arvo, :661-669 */
static void compare_and_swap_arvo(psa_arvo_t* p, uint32_t lhs, uint32_t rhs) {
	float lhs_azimuth = p->azimuths[lhs];
	float flip = p->azimuths[lhs] - p->azimuths[rhs];
	p->azimuths[lhs] = (flip > 0.0f) ? p->azimuths[rhs] : lhs_azimuth;
	p->azimuths[rhs] = (flip > 0.0f) ? lhs_azimuth : p->azimuths[rhs];
	edge_arvo_t lhs_edge = p->edges[lhs];
	p->edges[lhs] = (flip > 0.0f) ? p->edges[rhs] : lhs_edge;
	p->edges[rhs] = (flip > 0.0f) ? lhs_edge : p->edges[rhs];
}

/* sort_convex_polygon_vertices_arvo, :674-739: the same networks as sort_convex_polygon_vertices */
static void sort_vertices_arvo(psa_arvo_t* p, uint32_t cap) {
	static const uint8_t networks[9][9][2] = {
		{{0}}, {{0}}, {{0}},
		/* 3 */ {{1, 2}},
		/* 4 */ {{1, 3}},
		/* 5 */ {{2, 4}, {1, 3}, {1, 2}, {0, 3}, {3, 4}},
		/* 6 */ {{3, 5}, {2, 4}, {1, 5}, {0, 4}, {4, 5}, {1, 3}},
		/* 7 */ {{2, 5}, {1, 6}, {5, 6}, {3, 4}, {0, 4}, {4, 6}, {1, 3}, {3, 5}, {4, 5}},
		/* 8 */ {{2, 6}, {3, 7}, {1, 5}, {0, 4}, {4, 6}, {5, 7}, {6, 7}, {4, 5}, {1, 3}}};
	static const uint8_t lengths[9] = {0, 0, 0, 1, 1, 5, 6, 9, 9};
	uint32_t n = p->vertex_count;
	if (n >= 3 && n <= 8 && n <= cap)
		for (uint32_t i = 0; i != lengths[n]; ++i) compare_and_swap_arvo(p, networks[n][i][0], networks[n][i][1]);
	compare_and_swap_arvo(p, 0, 2);
	if (cap >= 4 && n >= 4) compare_and_swap_arvo(p, 2, 3);
	compare_and_swap_arvo(p, 0, 1);
}

/* prepare_projected_solid_angle_polygon_sampling_arvo, :744-812 */
static psa_arvo_t prepare_psa_arvo(uint32_t vertex_count, uint32_t cap, const v3* in_vertices) {
	psa_arvo_t p;
	memset(&p, 0, sizeof(p));
	v3 vertices[O_CAP];
	for (uint32_t i = 0; i != cap; ++i) vertices[i] = normalize3(in_vertices[i]);
	p.vertex_count = vertex_count;
	p.inner_edge_0.cdf_factor = 1.0f;
	p.inner_edge_0.length_coeffs = p.inner_edge_0.elevations = mk2(0.0f, 0.0f);
	p.azimuths[0] = o_atan2(vertices[0].y, vertices[0].x);
	p.edges[0] = prepare_edge_arvo(vertices[0], vertices[1]);
	edge_arvo_t previous_edge = p.edges[0];
	for (uint32_t i = 1; i != cap; ++i) {
		p.azimuths[i] = o_atan2(vertices[i].y, vertices[i].x);
		p.azimuths[i] -= (p.azimuths[i] > p.azimuths[0] + O_PI) ? (2.0f * O_PI) : 0.0f;
		p.azimuths[i] += (p.azimuths[i] < p.azimuths[0] - O_PI) ? (2.0f * O_PI) : 0.0f;
		if (i > 2 && i == p.vertex_count) break;
		edge_arvo_t edge = prepare_edge_arvo(vertices[i], vertices[(i + 1) % cap]);
		p.edges[i] = (edge.cdf_factor >= 0.0f) ? edge : previous_edge;
		p.inner_edge_0 = (previous_edge.cdf_factor < 0.0f && edge.cdf_factor >= 0.0f) ? previous_edge : p.inner_edge_0;
		previous_edge = edge;
	}
	edge_arvo_t edge = p.edges[0];
	p.edges[0] = (edge.cdf_factor >= 0.0f) ? edge : previous_edge;
	p.inner_edge_0 = (previous_edge.cdf_factor < 0.0f && edge.cdf_factor >= 0.0f) ? previous_edge : p.inner_edge_0;
	p.psa = 0.0f;
	if (p.inner_edge_0.cdf_factor > 0.0f) {
		for (uint32_t i = 0; i != cap; ++i) {
			if (i > 2 && i == p.vertex_count) break;
			p.sector_psa[i] = edge_psa_in_sector_arvo(&p.edges[i], 0.0f, p.azimuths[(i + 1) % cap] - p.azimuths[i]);
			p.psa += p.sector_psa[i];
		}
	}
	else {
		sort_vertices_arvo(&p, cap);
		edge_arvo_t inner_edge = p.inner_edge_0;
		float inner_azimuth = p.azimuths[0];
		edge_arvo_t outer_edge;
		memset(&outer_edge, 0, sizeof(outer_edge));
		float outer_azimuth = p.azimuths[0];
		for (uint32_t i = 0; i + 1 != cap; ++i) {
			if (i > 1 && i + 1 == p.vertex_count) break;
			edge_arvo_t vertex_edge = p.edges[i];
			float vertex_azimuth = p.azimuths[i];
			if (i == 0) outer_edge = vertex_edge;
			else {
				int outer = vertex_edge.cdf_factor >= 0.0f;
				inner_edge = outer ? inner_edge : vertex_edge;
				inner_azimuth = outer ? inner_azimuth : vertex_azimuth;
				outer_edge = outer ? vertex_edge : outer_edge;
				outer_azimuth = outer ? vertex_azimuth : outer_azimuth;
			}
			p.sector_psa[i] = edge_psa_in_sector_arvo(&outer_edge, p.azimuths[i] - outer_azimuth, p.azimuths[i + 1] - outer_azimuth);
			p.sector_psa[i] += edge_psa_in_sector_arvo(&inner_edge, p.azimuths[i] - inner_azimuth, p.azimuths[i + 1] - inner_azimuth);
			p.psa += p.sector_psa[i];
		}
	}
	return p;
}

/* evaluate_cubic_interpolation_polynomial, :822-830 */
static float cubic_interpolation(float sample_x, const float x[4], const float y[4]) {
	float y01 = (y[0] - y[1]) / (x[0] - x[1]);
	float y12 = (y[1] - y[2]) / (x[1] - x[2]);
	float y23 = (y[2] - y[3]) / (x[2] - x[3]);
	float y012 = (y01 - y12) / (x[0] - x[2]);
	float y123 = (y12 - y23) / (x[1] - x[3]);
	float y0123 = (y012 - y123) / (x[0] - x[3]);
	return fmaf(sample_x - x[0], fmaf(sample_x - x[1], fmaf(sample_x - x[2], y0123, y012), y01), y[0]);
}

/* sample_sector_within_edge (inner_edge == NULL), :838-866, and sample_sector_between_edges, :890-925 */
static v3 sample_sector_arvo(v2 random_numbers, float target, const edge_arvo_t* inner_edge, float inner_azimuth, const edge_arvo_t* outer_edge, float outer_azimuth, float azimuth_0, float azimuth_1, uint32_t iteration_count) {
	float azimuths[4] = {azimuth_0, mix_fma(azimuth_0, azimuth_1, 1.0f / 3.0f), mix_fma(azimuth_0, azimuth_1, 2.0f / 3.0f), azimuth_1};
	float psas[4];
	for (int i = 0; i != 4; ++i) {
		psas[i] = edge_psa_in_sector_arvo(outer_edge, azimuth_0 - outer_azimuth, azimuths[i] - outer_azimuth);
		if (inner_edge) psas[i] += edge_psa_in_sector_arvo(inner_edge, azimuth_0 - inner_azimuth, azimuths[i] - inner_azimuth);
	}
	float sampled_azimuth = cubic_interpolation(target, psas, azimuths);
	for (uint32_t i = 0; i != iteration_count; ++i) {
		v2 outer_psa = edge_psa_in_sector_derivative_arvo(outer_edge, azimuth_0 - outer_azimuth, sampled_azimuth - outer_azimuth);
		float error, derivative;
		if (inner_edge) {
			v2 inner_psa = edge_psa_in_sector_derivative_arvo(inner_edge, azimuth_0 - inner_azimuth, sampled_azimuth - inner_azimuth);
			error = inner_psa.x + outer_psa.x - target;
			derivative = inner_psa.y + outer_psa.y;
		}
		else {
			error = outer_psa.x - target;
			derivative = outer_psa.y;
		}
		sampled_azimuth -= error / derivative;
		sampled_azimuth = g_clamp(sampled_azimuth, azimuth_0, azimuth_1);
	}
	v3 dir;
	float sn, cs;
	o_sincos(sampled_azimuth, &sn, &cs);
	dir.x = cs;
	dir.y = sn;
	float outer_z = edge_elevation_arvo(outer_edge, sampled_azimuth - outer_azimuth);
	if (inner_edge) {
		float inner_z = edge_elevation_arvo(inner_edge, sampled_azimuth - inner_azimuth);
		dir.z = sqrtf(mix_fma(inner_z * inner_z, outer_z * outer_z, random_numbers.y));
	}
	else
		dir.z = sqrtf(mix_fma(1.0f, outer_z * outer_z, random_numbers.y));
	float scale = sqrtf(fmaf(-dir.z, dir.z, 1.0f));
	dir.x *= scale;
	dir.y *= scale;
	return dir;
}

/* the sector search shared by sampling and error computation in the decentral case, :965-988, :1015-1035 */
static void find_sector_arvo(const psa_arvo_t* p, uint32_t cap, float* target, float* sector_psa, edge_arvo_t* inner_edge, float* inner_azimuth, edge_arvo_t* outer_edge, float* outer_azimuth, float* azimuth_0, float* azimuth_1) {
	*inner_edge = p->inner_edge_0;
	*inner_azimuth = p->azimuths[0];
	for (uint32_t i = 0; i + 1 != cap; ++i) {
		if ((i > 1 && i + 1 == p->vertex_count) || (i > 0 && *target < 0.0f)) break;
		*sector_psa = p->sector_psa[i];
		*target -= *sector_psa;
		edge_arvo_t vertex_edge = p->edges[i];
		float vertex_azimuth = p->azimuths[i];
		if (i == 0) {
			*outer_edge = vertex_edge;
			*outer_azimuth = vertex_azimuth;
		}
		else {
			int outer = vertex_edge.cdf_factor >= 0.0f;
			*inner_edge = outer ? *inner_edge : vertex_edge;
			*inner_azimuth = outer ? *inner_azimuth : vertex_azimuth;
			*outer_edge = outer ? vertex_edge : *outer_edge;
			*outer_azimuth = outer ? vertex_azimuth : *outer_azimuth;
		}
		*azimuth_0 = p->azimuths[i];
		*azimuth_1 = p->azimuths[i + 1];
	}
	*target += *sector_psa;
}

/* sample_projected_solid_angle_polygon_arvo, :934-991 */
static v3 sample_psa_arvo(const psa_arvo_t* p, uint32_t cap, v2 random_numbers, uint32_t iteration_count) {
	float target = random_numbers.x * p->psa;
	float sector_psa = 0.0f;
	edge_arvo_t outer_edge;
	memset(&outer_edge, 0, sizeof(outer_edge));
	float outer_azimuth = 0.0f, azimuth_1 = 0.0f;
	if (p->inner_edge_0.cdf_factor > 0.0f) {
		for (uint32_t i = 0; i != cap; ++i) {
			if ((i > 2 && i == p->vertex_count) || (i > 0 && target < 0.0f)) break;
			sector_psa = p->sector_psa[i];
			target -= sector_psa;
			outer_edge = p->edges[i];
			outer_azimuth = p->azimuths[i];
			azimuth_1 = p->azimuths[(i + 1) % cap];
		}
		azimuth_1 = (azimuth_1 < outer_azimuth) ? (azimuth_1 + 2.0f * O_PI) : azimuth_1;
		target += sector_psa;
		random_numbers.x = g_clamp(target / sector_psa, 0.0f, 1.0f);
		return sample_sector_arvo(random_numbers, target, NULL, 0.0f, &outer_edge, outer_azimuth, outer_azimuth, azimuth_1, iteration_count);
	}
	edge_arvo_t inner_edge;
	float inner_azimuth, azimuth_0 = 0.0f;
	find_sector_arvo(p, cap, &target, &sector_psa, &inner_edge, &inner_azimuth, &outer_edge, &outer_azimuth, &azimuth_0, &azimuth_1);
	random_numbers.x = g_clamp(target / sector_psa, 0.0f, 1.0f);
	return sample_sector_arvo(random_numbers, target, &inner_edge, inner_azimuth, &outer_edge, outer_azimuth, azimuth_0, azimuth_1, iteration_count);
}

/* compute_projected_solid_angle_polygon_sampling_error_arvo, :998-1047: (backward, backward scaled) */
static v2 psa_sampling_error_arvo(const psa_arvo_t* p, uint32_t cap, v2 random_numbers, v3 sampled_dir) {
	float target = random_numbers.x * p->psa;
	if (p->inner_edge_0.cdf_factor > 0.0f) return mk2(0.0f, 0.0f);
	edge_arvo_t inner_edge, outer_edge;
	memset(&outer_edge, 0, sizeof(outer_edge));
	float inner_azimuth, outer_azimuth = 0.0f, sector_psa = 0.0f, azimuth_0 = 0.0f, azimuth_1 = 0.0f;
	find_sector_arvo(p, cap, &target, &sector_psa, &inner_edge, &inner_azimuth, &outer_edge, &outer_azimuth, &azimuth_0, &azimuth_1);
	float sampled_azimuth = o_atan2(sampled_dir.y, sampled_dir.x);
	float outer_psa = edge_psa_in_sector_derivative_arvo(&outer_edge, azimuth_0 - outer_azimuth, sampled_azimuth - outer_azimuth).x;
	float inner_psa = edge_psa_in_sector_derivative_arvo(&inner_edge, azimuth_0 - inner_azimuth, sampled_azimuth - inner_azimuth).x;
	float sampled_psa = outer_psa + inner_psa;
	return mk2((target - sampled_psa) / p->psa, target - sampled_psa);
}

/* ---- related work: Urena's rectangle sampling, Arvo's spherical triangles, Hart's warps ----
 * (polygon_sampling_related_work.glsl:97-386) */

typedef struct {
	v3 o, x, y, z;
	float z0, z0sq, x0, y0, y0sq, x1, y1, y1sq, b0, b1, b0sq, k, solid_angle;
} urena_rectangle_t;

/* prepare_solid_angle_rectangle_sampling_urena, :121-164 */
static urena_rectangle_t prepare_urena(v3 s, float exl, float eyl, const v3 rotation[3], v3 o) {
	urena_rectangle_t q;
	q.o = o;
	q.x = rotation[0]; q.y = rotation[1]; q.z = rotation[2];
	v3 d = sub3(s, o);
	q.z0 = dot3(d, q.z);
	q.z = (q.z0 > 0.0f) ? neg3(q.z) : q.z;
	q.z0 = -fabsf(q.z0);
	q.z0sq = q.z0 * q.z0;
	q.x0 = dot3(d, q.x);
	q.y0 = dot3(d, q.y);
	q.x1 = q.x0 + exl;
	q.y1 = q.y0 + eyl;
	q.y0sq = q.y0 * q.y0;
	q.y1sq = q.y1 * q.y1;
	v3 v00 = mk3(q.x0, q.y0, q.z0), v01 = mk3(q.x0, q.y1, q.z0), v10 = mk3(q.x1, q.y0, q.z0), v11 = mk3(q.x1, q.y1, q.z0);
	v3 n0 = normalize3(cross3(v00, v10)), n1 = normalize3(cross3(v10, v11)), n2 = normalize3(cross3(v11, v01)), n3 = normalize3(cross3(v01, v00));
	float g0 = o_acos(-dot3(n0, n1)), g1 = o_acos(-dot3(n1, n2)), g2 = o_acos(-dot3(n2, n3)), g3 = o_acos(-dot3(n3, n0));
	q.b0 = n0.z;
	q.b1 = n2.z;
	q.b0sq = q.b0 * q.b0;
	q.k = 2.0f * O_PI - g2 - g3;
	q.solid_angle = g0 + g1 - q.k;
	return q;
}

/* sample_solid_angle_rectangle_urena, :171-193 */
static v3 sample_urena(const urena_rectangle_t* q, v2 random_numbers) {
	float u = random_numbers.x, v = random_numbers.y;
	float au = fmaf(u, q->solid_angle, q->k);
	float sin_au, cos_au;
	o_sincos(au, &sin_au, &cos_au);
	float fu = fmaf(cos_au, q->b0, -q->b1) / sin_au;
	float cu = rsqrt_f(fmaf(fu, fu, q->b0sq));
	cu = (fu > 0.0f) ? cu : -cu;
	cu = g_clamp(cu, -1.0f, 1.0f);
	float xu = -(cu * q->z0) * rsqrt_f(fmaf(-cu, cu, 1.0f));
	xu = g_clamp(xu, q->x0, q->x1);
	float d = sqrtf(xu * xu + q->z0sq);
	float h0 = q->y0 * rsqrt_f(fmaf(d, d, q->y0sq));
	float h1 = q->y1 * rsqrt_f(fmaf(d, d, q->y1sq));
	float hv = h0 + v * (h1 - h0);
	float mhv2_1 = fmaf(-hv, hv, 1.0f);
	float yv = (mhv2_1 >= 0.0f) ? ((hv * d) * rsqrt_f(mhv2_1)) : q->y1;
	return normalize3(add3(add3(scale3(q->x, xu), scale3(q->y, yv)), scale3(q->z, q->z0)));
}

typedef struct {
	uint32_t vertex_count;
	v3 dirs[O_CAP];
	float fan[O_CAP];
	v2 opposite[O_CAP];
	float solid_angle;
} arvo_polygon_t;

/* prepare_solid_angle_polygon_sampling_arvo, :219-254 */
static arvo_polygon_t prepare_arvo(uint32_t vertex_count, uint32_t cap, const v3* verts, v3 shading_position) {
	arvo_polygon_t p;
	memset(&p, 0, sizeof(p));
	for (uint32_t i = 0; i != cap; ++i) p.dirs[i] = normalize3(sub3(verts[i], shading_position));
	float solid_angle = 0.0f;
	for (uint32_t i = 0; i + 2 != cap; ++i) {
		if (i >= 1 && i + 2 >= vertex_count) break;
		v3 n0 = normalize3(cross3(sub3(p.dirs[i + 1], p.dirs[0]), p.dirs[0]));
		v3 n1 = normalize3(cross3(sub3(p.dirs[i + 2], p.dirs[i + 1]), p.dirs[i + 1]));
		p.opposite[i].x = -dot3(n0, n1);
		p.opposite[i].y = sqrtf(g_max(0.0f, fmaf(-p.opposite[i].x, p.opposite[i].x, 1.0f)));
		float d01 = dot3(p.dirs[0], p.dirs[i + 1]), d02 = dot3(p.dirs[0], p.dirs[i + 2]), d12 = dot3(p.dirs[i + 1], p.dirs[i + 2]);
		/* determinant(mat3(c0, c1, c2)), expanded along the first column */
		v3 c0 = p.dirs[0], c1 = p.dirs[i + 1], c2 = p.dirs[i + 2];
		float volume = c0.x * (c1.y * c2.z - c2.y * c1.z) - c1.x * (c0.y * c2.z - c2.y * c0.z) + c2.x * (c0.y * c1.z - c1.y * c0.z);
		float tangent = fabsf(volume) / (1.0f + d01 + d02 + d12);
		solid_angle += 2.0f * positive_atan(tangent, 0);
		p.fan[i] = solid_angle;
	}
	p.solid_angle = solid_angle;
	p.vertex_count = vertex_count;
	return p;
}

/* sample_solid_angle_polygon_arvo, :259-294 */
static v3 sample_arvo(const arvo_polygon_t* p, uint32_t cap, v2 random_numbers) {
	float target = p->solid_angle * random_numbers.x;
	float sub = target;
	v2 opposite = p->opposite[0];
	v3 t0 = p->dirs[1], t1 = p->dirs[0], t2 = p->dirs[2];
	for (uint32_t i = 0; i + 3 != cap; ++i) {
		if (i + 3 >= p->vertex_count || p->fan[i] >= target) break;
		sub = target - p->fan[i];
		t0 = p->dirs[i + 2];
		t2 = p->dirs[i + 3];
		opposite = p->opposite[i + 1];
	}
	float sn, cs;
	o_sincos(sub, &sn, &cs);
	float pp = sn * opposite.x - cs * opposite.y;
	float q = sn * opposite.y + cs * opposite.x;
	float u = q - opposite.x;
	float v = pp + opposite.y * dot3(t0, t1);
	float s = ((v * q - u * pp) * opposite.x - v) / ((v * pp + u * q) * opposite.y);
	v3 tangent_2_0 = normalize3(sub3(t2, scale3(t0, dot3(t0, t2))));
	v3 vertex_2 = add3(scale3(t0, s), scale3(tangent_2_0, sqrtf(g_clamp(fmaf(-s, s, 1.0f), 0.0f, 1.0f))));
	float z = 1.0f - random_numbers.y * (1.0f - dot3(vertex_2, t1));
	v3 tangent_2_1 = normalize3(sub3(vertex_2, scale3(t1, dot3(t1, vertex_2))));
	return add3(scale3(t1, z), scale3(tangent_2_1, sqrtf(g_clamp(fmaf(-z, z, 1.0f), 0.0f, 1.0f))));
}

typedef struct {
	sa_polygon_t polygon;
	float density_0;
	v2 density_1;
} hart_bilinear_t;

/* prepare_bilinear_cosine_warp_polygon_sampling_hart, :316-337 */
static hart_bilinear_t prepare_hart_bilinear(uint32_t vertex_count, uint32_t cap, const v3* verts) {
	hart_bilinear_t h;
	h.polygon = prepare_sa(vertex_count, cap, verts, mk3(0.0f, 0.0f, 0.0f));
	h.density_0 = g_max(0.0f, h.polygon.dirs[0].z);
	h.density_1.x = g_max(0.0f, h.polygon.dirs[1].z);
	h.density_1.y = h.polygon.dirs[2].z;
	for (uint32_t i = 3; i != cap; ++i) h.density_1.y = (i < vertex_count) ? h.polygon.dirs[i].z : h.density_1.y;
	h.density_1.y = g_max(0.0f, h.density_1.y);
	float density_sum = 2.0f * h.density_0 + h.density_1.x + h.density_1.y;
	float normalization = 4.0f / (h.polygon.solid_angle * density_sum);
	h.density_0 *= normalization;
	h.density_1 = scale2(h.density_1, normalization);
	float inv_solid_angle = 1.0f / h.polygon.solid_angle;
	if (density_sum <= 0.0f) {
		h.density_0 = inv_solid_angle;
		h.density_1 = mk2(inv_solid_angle, inv_solid_angle);
	}
	return h;
}

/* linear_warp, :349-353 */
static float linear_warp(float random_number, float density_0, float density_1) {
	float lerped_density_sq = mix_fma(density_0 * density_0, density_1 * density_1, random_number);
	float divisor = density_0 + sqrtf(lerped_density_sq);
	return random_number * (density_0 + density_1) / divisor;
}

/* sample_bilinear_cosine_warp_polygon_hart, :373-380 */
static v3 sample_hart_bilinear(float* out_density, const hart_bilinear_t* h, uint32_t cap, v2 u) {
	u.y = linear_warp(u.y, 2.0f * h->density_0, h->density_1.x * 1.0f + h->density_1.y * 1.0f);
	float density_0 = mix_fma(h->density_0, h->density_1.x, u.y);
	float density_1 = mix_fma(h->density_0, h->density_1.y, u.y);
	u.x = linear_warp(u.x, density_0, density_1);
	*out_density = mix_fma(density_0, density_1, u.x);
	return sample_sa(&h->polygon, cap, u);
}

/* solve_cubic, cubic_solver.glsl:29-76: c[0] + c[1] x + c[2] x^2 + c[3] x^3; returns 1 with three
 * roots or 0 with one root in roots[0] */
static int solve_cubic(float roots[3], const float coeffs[4]) {
	float c0 = coeffs[0] / coeffs[3], c1 = coeffs[1] / coeffs[3], c2 = coeffs[2] / coeffs[3];
	c1 = c1 / 3.0f;
	c2 = c2 / 3.0f;
	float d0 = fmaf(-c2, c2, c1), d1 = fmaf(-c1, c2, c0), d2 = c2 * c0 - c1 * c1;
	float discriminant = 4.0f * d0 * d2 - d1 * d1;
	float sqrt_abs_discriminant = sqrtf(fabsf(discriminant));
	float depressed_0 = fmaf(-2.0f * c2, d0, d1), depressed_1 = d0;
	if (discriminant >= 0.0f) {
		float theta = o_atan2(sqrt_abs_discriminant, -depressed_0) * (1.0f / 3.0f);
		float sn, cs;
		o_sincos(theta, &sn, &cs);
		const float sqrt_three_quarters = 0.866025388f; /* sqrt(0.75f) */
		float r0 = cs, r1 = fmaf(-sqrt_three_quarters, sn, -0.5f * cs), r2 = fmaf(sqrt_three_quarters, sn, -0.5f * cs);
		float scale = 2.0f * sqrtf(-depressed_1);
		roots[0] = fmaf(scale, r0, -c2);
		roots[1] = fmaf(scale, r1, -c2);
		roots[2] = fmaf(scale, r2, -c2);
		return 1;
	}
	float signed_sqrt_discriminant = (depressed_0 < 0.0f) ? sqrt_abs_discriminant : -sqrt_abs_discriminant;
	float quadratic_root = 0.5f * (signed_sqrt_discriminant - depressed_0);
	float cube_root_0 = o_pow_third(fabsf(quadratic_root));
	cube_root_0 = (quadratic_root < 0.0f) ? -cube_root_0 : cube_root_0;
	float cube_root_1 = -depressed_1 / cube_root_0;
	roots[0] = (cube_root_0 + cube_root_1) - c2;
	roots[1] = roots[2] = 0.0f;
	return 0;
}

typedef struct {
	sa_polygon_t polygon;
	float density_0;
	v3 density_1, density_2;
} hart_biquadratic_t;

/* prepare_biquadratic_cosine_warp_polygon_sampling_hart, :405-446 */
static hart_biquadratic_t prepare_hart_biquadratic(uint32_t vertex_count, uint32_t cap, const v3* verts) {
	hart_biquadratic_t h;
	h.polygon = prepare_sa(vertex_count, cap, verts, mk3(0.0f, 0.0f, 0.0f));
	v3 last_vertex = h.polygon.dirs[2];
	for (uint32_t i = 3; i != cap; ++i) last_vertex = (i < vertex_count) ? h.polygon.dirs[i] : last_vertex;
	v3 vertex_0 = h.polygon.dirs[0];
	h.density_0 = g_max(0.0f, vertex_0.z);
	h.density_2.x = g_max(0.0f, h.polygon.dirs[1].z);
	h.density_2.z = g_max(0.0f, last_vertex.z);
	v3 sample_2_1 = sample_sa(&h.polygon, cap, mk2(0.5f, 1.0f));
	h.density_2.y = g_max(0.0f, sample_2_1.z);
	v3 far_vertices[3] = {vertex_0, sample_2_1, last_vertex};
	float density_1[3];
	for (int i = 0; i != 3; ++i) {
		float s2 = dot3(vertex_0, far_vertices[i]);
		float s = fmaf(0.5f, s2, 0.5f);
		float t = sqrtf(g_max(0.0f, fmaf(-s, s, 1.0f)));
		float t_axis_z = fmaf(-s2, vertex_0.z, far_vertices[i].z);
		float normalization_t_axis = rsqrt_f(2.0f * fmaf(-s2, s2, 1.0f));
		float sample_1_i_z = s * vertex_0.z + (t * normalization_t_axis) * t_axis_z;
		density_1[i] = g_max(0.0f, sample_1_i_z);
	}
	h.density_1 = mk3(density_1[0], density_1[1], density_1[2]);
	float density_sum = 3.0f * h.density_0 + ((h.density_1.x + h.density_1.y) + h.density_1.z) + ((h.density_2.x + h.density_2.y) + h.density_2.z);
	float normalization = 9.0f / (h.polygon.solid_angle * density_sum);
	h.density_0 *= normalization;
	h.density_1 = scale3(h.density_1, normalization);
	h.density_2 = scale3(h.density_2, normalization);
	float inv_solid_angle = 1.0f / h.polygon.solid_angle;
	if (density_sum <= 0.0f) {
		h.density_0 = inv_solid_angle;
		h.density_1 = h.density_2 = mk3(inv_solid_angle, inv_solid_angle, inv_solid_angle);
	}
	return h;
}

/* quadratic_warp, :457-474 */
static float quadratic_warp(float random_number, float density_0, float density_1, float density_2) {
	float q0 = density_0, q1 = 2.0f * (density_1 - density_0), q2 = density_0 - 2.0f * density_1 + density_2;
	float cubic[4] = {0.0f, q0, 0.5f * q1, (1.0f / 3.0f) * q2};
	random_number *= (cubic[1] + cubic[2]) + cubic[3];
	cubic[0] = -random_number;
	float roots[3];
	if (solve_cubic(roots, cubic)) {
		float result = roots[0];
		result = (roots[1] >= 0.0f && roots[1] <= 1.0f) ? roots[1] : result;
		result = (roots[2] >= 0.0f && roots[2] <= 1.0f) ? roots[2] : result;
		return result;
	}
	return roots[0];
}

/* quadratic_bezier, :484-488 */
static float quadratic_bezier(float b00, float b01, float b02, float location) {
	return mix_fma(mix_fma(b00, b01, location), mix_fma(b01, b02, location), location);
}

/* sample_biquadratic_cosine_warp_polygon_hart, :493-503 */
static v3 sample_hart_biquadratic(float* out_density, const hart_biquadratic_t* h, uint32_t cap, v2 u) {
	u.y = quadratic_warp(u.y, 3.0f * h->density_0, (h->density_1.x + h->density_1.y) + h->density_1.z, (h->density_2.x + h->density_2.y) + h->density_2.z);
	float density_0 = quadratic_bezier(h->density_0, h->density_1.x, h->density_2.x, u.y);
	float density_1 = quadratic_bezier(h->density_0, h->density_1.y, h->density_2.y, u.y);
	float density_2 = quadratic_bezier(h->density_0, h->density_1.z, h->density_2.z, u.y);
	u.x = quadratic_warp(u.x, density_0, density_1, density_2);
	*out_density = quadratic_bezier(density_0, density_1, density_2, u.x);
	return sample_sa(&h->polygon, cap, u);
}

/* sample_area_polygon_turk, polygon_sampling_related_work.glsl:38-64 (cap = MAX_POLYGON_VERTEX_COUNT) */
static v3 sample_area_turk(uint32_t vertex_count, uint32_t cap, const v3* vertices, const v2* fan_areas, v2 u) {
	float target_area = fan_areas[cap - 3].y * u.x;
	float subtriangle_area = target_area;
	float triangle_area = fan_areas[0].x;
	v3 t0 = vertices[1], t1 = vertices[0], t2 = vertices[2];
	for (uint32_t i = 0; i + 3 != cap; ++i) {
		if (i + 3 >= vertex_count || fan_areas[i].y >= target_area) break;
		subtriangle_area = target_area - fan_areas[i].y;
		triangle_area = fan_areas[i + 1].x;
		t0 = vertices[i + 2];
		t2 = vertices[i + 3];
	}
	u.x = subtriangle_area / triangle_area;
	float sqrt_u = sqrtf(u.x);
	float b0 = 1.0f - sqrt_u, b1 = sqrt_u * u.y, b2 = fmaf(-sqrt_u, u.y, sqrt_u);
	return add3(add3(scale3(t0, b0), scale3(t1, b1)), scale3(t2, b2));
}

/* get_area_sample_density, polygon_sampling_related_work.glsl:78-85 */
static float area_sample_density(v3* out_dir, v3 light_sample, v3 shading_position, v3 light_normal, float light_area) {
	v3 d = sub3(light_sample, shading_position);
	float distance_squared = dot3(d, d);
	float normalization = rsqrt_f(distance_squared);
	*out_dir = scale3(d, normalization);
	float projected_area = fabsf(dot3(light_normal, *out_dir)) * light_area;
	return distance_squared / projected_area;
}

/* evaluate_polygonal_light_shading, shading_pass.frag.glsl:329-711 */
static v3 evaluate_light(pixel_ctx_t* ctx, const shading_data_t* sd, const ltc_t* ltc_in, const light_view_t* light, noise_accessor_t* noise) {
	const oracle_frame_t* f = ctx->f;
	const frame_constants_t* k = ctx->k;
	uint32_t vmax = f->max_light_vertex_count;
	uint32_t S = f->sample_count;
	int technique = f->polygon_technique;
	int strategy = f->sampling_strategies;
	int biased = technique == O_TECHNIQUE_PROJECTED_SOLID_ANGLE_BIASED;
	int is_psa = technique == O_TECHNIQUE_PROJECTED_SOLID_ANGLE || biased;
	v3 result = mk3(0.0f, 0.0f, 0.0f);
	v3 zero = mk3(0.0f, 0.0f, 0.0f);
	ltc_t ltc = *ltc_in;
	float density_factor = 0.0f;

	if (technique == O_TECHNIQUE_BASELINE) {
		/* :332-342: "broken" on purpose, the run time baseline of the paper */
		v3 corner_offset = sub3(light->translation, sd->position);
		for (uint32_t s = 0; s != S; ++s) {
			v2 u = next_noise_2(f, k, noise);
			v3 dir = normalize3(add3(add3(corner_offset, scale3(light->rotation_columns[0], u.x)), scale3(light->rotation_columns[1], u.y)));
			result = add3(result, light_mis_estimate(ctx, dir, 1.0f, sd, light));
		}
	}
	else if (technique == O_TECHNIQUE_AREA_TURK) {
		/* :344-350 */
		for (uint32_t s = 0; s != S; ++s) {
			v3 light_sample = sample_area_turk(light->vertex_count, vmax, light->vertices_world, light->fan_areas, next_noise_2(f, k, noise));
			v3 dir;
			float density = area_sample_density(&dir, light_sample, sd->position, mk3(light->plane.x, light->plane.y, light->plane.z), light->area);
			result = add3(result, light_mis_estimate(ctx, dir, density, sd, light));
		}
	}
	else if (technique == O_TECHNIQUE_RECTANGLE_SOLID_ANGLE_URENA) {
		/* :352-362: the light is taken to be the unit square of its plane */
		urena_rectangle_t pd = prepare_urena(light->translation, light->scaling_x, light->scaling_y, light->rotation_columns, sd->position);
		for (uint32_t s = 0; s != S; ++s) {
			v3 dir = sample_urena(&pd, next_noise_2(f, k, noise));
			result = add3(result, light_mis_estimate(ctx, dir, 1.0f / pd.solid_angle, sd, light));
		}
		density_factor = 1.0f / pd.solid_angle;
	}
	else if (technique == O_TECHNIQUE_SOLID_ANGLE_ARVO) {
		/* :364-373 */
		arvo_polygon_t pd = prepare_arvo(light->vertex_count, vmax, light->vertices_world, sd->position);
		for (uint32_t s = 0; s != S; ++s) {
			v3 dir = sample_arvo(&pd, vmax, next_noise_2(f, k, noise));
			result = add3(result, light_mis_estimate(ctx, dir, 1.0f / pd.solid_angle, sd, light));
		}
		density_factor = 1.0f / pd.solid_angle;
	}
	else if (technique == O_TECHNIQUE_BILINEAR_COSINE_WARP_HART || technique == O_TECHNIQUE_BILINEAR_COSINE_WARP_CLIPPING_HART) {
		/* :386-427 */
		int clipping = technique == O_TECHNIQUE_BILINEAR_COSINE_WARP_CLIPPING_HART;
		uint32_t cap = vmax + (clipping ? 1 : 0);
		v3 vs[O_CAP];
		memset(vs, 0, sizeof(vs));
		for (uint32_t i = 0; i != vmax; ++i) vs[i] = m43_mul(&ltc.world_to_shading, light->vertices_world[i], 1.0f);
		uint32_t clipped = light->vertex_count;
		if (clipping) {
			clipped = clip_polygon(light->vertex_count, 3, cap, vs);
			if (clipped == 0) return zero;
		}
		hart_bilinear_t pd = prepare_hart_bilinear(clipped, cap, vs);
		for (uint32_t s = 0; s != S; ++s) {
			float density;
			v3 dir = sample_hart_bilinear(&density, &pd, cap, next_noise_2(f, k, noise));
			dir = m43_mul_transposed(&ltc.world_to_shading, dir);
			result = add3(result, light_mis_estimate(ctx, dir, density, sd, light));
		}
	}
	else if (technique == O_TECHNIQUE_BIQUADRATIC_COSINE_WARP_HART || technique == O_TECHNIQUE_BIQUADRATIC_COSINE_WARP_CLIPPING_HART) {
		/* :386-401, :428-438 */
		int clipping = technique == O_TECHNIQUE_BIQUADRATIC_COSINE_WARP_CLIPPING_HART;
		uint32_t cap = vmax + (clipping ? 1 : 0);
		v3 vs[O_CAP];
		memset(vs, 0, sizeof(vs));
		for (uint32_t i = 0; i != vmax; ++i) vs[i] = m43_mul(&ltc.world_to_shading, light->vertices_world[i], 1.0f);
		uint32_t clipped = light->vertex_count;
		if (clipping) {
			clipped = clip_polygon(light->vertex_count, 3, cap, vs);
			if (clipped == 0) return zero;
		}
		hart_biquadratic_t pd = prepare_hart_biquadratic(clipped, cap, vs);
		for (uint32_t s = 0; s != S; ++s) {
			float density;
			v3 dir = sample_hart_biquadratic(&density, &pd, cap, next_noise_2(f, k, noise));
			dir = m43_mul_transposed(&ltc.world_to_shading, dir);
			result = add3(result, light_mis_estimate(ctx, dir, density, sd, light));
		}
	}
	else if (technique == O_TECHNIQUE_SOLID_ANGLE) {
		/* :375-384 */
		sa_polygon_t pd = prepare_sa(light->vertex_count, vmax, light->vertices_world, sd->position);
		for (uint32_t s = 0; s != S; ++s) {
			v3 dir = sample_sa(&pd, vmax, next_noise_2(f, k, noise));
			result = add3(result, light_mis_estimate(ctx, dir, 1.0f / pd.solid_angle, sd, light));
		}
		density_factor = 1.0f / pd.solid_angle;
	}
	else if (technique == O_TECHNIQUE_CLIPPED_SOLID_ANGLE) {
		/* :386-413 */
		uint32_t cap = vmax + 1;
		v3 vs[O_CAP];
		memset(vs, 0, sizeof(vs));
		for (uint32_t i = 0; i != vmax; ++i) vs[i] = m43_mul(&ltc.world_to_shading, light->vertices_world[i], 1.0f);
		uint32_t clipped = clip_polygon(light->vertex_count, 3, cap, vs);
		if (clipped == 0) return zero;
		sa_polygon_t pd = prepare_sa(clipped, cap, vs, zero);
		for (uint32_t s = 0; s != S; ++s) {
			v3 dir = sample_sa(&pd, cap, next_noise_2(f, k, noise));
			dir = m43_mul_transposed(&ltc.world_to_shading, dir);
			result = add3(result, light_mis_estimate(ctx, dir, 1.0f / pd.solid_angle, sd, light));
		}
		density_factor = 1.0f / pd.solid_angle;
	}
	else if (is_psa || technique == O_TECHNIQUE_PROJECTED_SOLID_ANGLE_ARVO) {
		uint32_t cap = vmax + 1;
		/* Arvo's sampler only exists in the diffuse-only / GGX-MIS branch (:462-481); the combined
		 * branch uses the paper's own sampler whatever the technique says (:506-547) */
		int is_arvo = technique == O_TECHNIQUE_PROJECTED_SOLID_ANGLE_ARVO
			&& (strategy == O_STRATEGY_DIFFUSE_ONLY || strategy == O_STRATEGY_DIFFUSE_GGX_MIS);
		/* :444-449 flip the frame when the shading point is behind the light */
		float side = dot4_point(sd->position, light->plane);
		if (side < 0.0f)
			for (int i = 0; i != 4; ++i) {
				ltc.world_to_shading.c[i].y = -ltc.world_to_shading.c[i].y;
				ltc.world_to_cosine.c[i].y = -ltc.world_to_cosine.c[i].y;
			}
		if (strategy == O_STRATEGY_DIFFUSE_ONLY || strategy == O_STRATEGY_DIFFUSE_GGX_MIS) {
			/* :451-502 */
			v3 vs[O_CAP];
			memset(vs, 0, sizeof(vs));
			for (uint32_t i = 0; i != vmax; ++i) vs[i] = m43_mul(&ltc.world_to_shading, light->vertices_world[i], 1.0f);
			uint32_t clipped = clip_polygon(light->vertex_count, 3, cap, vs);
			if (clipped == 0) return zero;
			if (is_arvo) {
				psa_arvo_t pa = prepare_psa_arvo(clipped, cap, vs);
				if (pa.psa <= 0.0f) return zero;
				if (f->error_display == 1) {
					v2 u = next_noise_2(f, k, noise);
					v3 dir = sample_psa_arvo(&pa, cap, u, 3);
					v2 e = psa_sampling_error_arvo(&pa, cap, u, dir);
					v3 color = error_to_color(k, (f->error_index == 0) ? e.x : e.y);
					return mk3(color.x / k->exposure_factor, color.y / k->exposure_factor, color.z / k->exposure_factor);
				}
				for (uint32_t s = 0; s != S; ++s) {
					v3 dir = sample_psa_arvo(&pa, cap, next_noise_2(f, k, noise), 3);
					float density = dir.z / pa.psa;
					dir = m43_mul_transposed(&ltc.world_to_shading, dir);
					result = add3(result, light_mis_estimate(ctx, dir, density, sd, light));
				}
				density_factor = 1.0f / pa.psa;
				goto ggx_tail;
			}
			psa_polygon_t pd = prepare_psa(clipped, cap, vs, biased);
			if (pd.psa <= 0.0f) return zero;
			if (f->error_display == 1) return display_sampling_error(ctx, &pd, cap, noise, biased);
			for (uint32_t s = 0; s != S; ++s) {
				v3 dir = sample_psa(&pd, cap, next_noise_2(f, k, noise), biased);
				float density = dir.z / pd.psa;
				dir = m43_mul_transposed(&ltc.world_to_shading, dir);
				result = add3(result, light_mis_estimate(ctx, dir, density, sd, light));
			}
			density_factor = 1.0f / pd.psa;
		}
		else {
			/* :506-547 prepare both techniques */
			psa_polygon_t pd, ps;
			memset(&pd, 0, sizeof(pd));
			memset(&ps, 0, sizeof(ps));
			for (int t = 0; t != 2; ++t) {
				const m43* to_local = (t == 0) ? &ltc.world_to_shading : &ltc.world_to_cosine;
				if (t > 0) pd = ps;
				v3 vl[O_CAP];
				memset(vl, 0, sizeof(vl));
				for (uint32_t j = 0; j != vmax; ++j) vl[j] = m43_mul(to_local, light->vertices_world[j], 1.0f);
				uint32_t clipped = clip_polygon(light->vertex_count, 3, cap, vl);
				if (clipped == 0 && t == 0) return zero;
				else if (clipped == 0) { ps.psa = 0.0f; break; }
				ps = prepare_psa(clipped, cap, vl, biased);
			}
			if (pd.psa == 0.0f) return zero;
			float specular_albedo = ltc.albedo;
			float specular_weight = specular_albedo * ps.psa;
			if (f->error_display == 1) return display_sampling_error(ctx, &pd, cap, noise, biased);
			if (f->error_display == 2) return (ps.psa > 0.0f) ? display_sampling_error(ctx, &ps, cap, noise, biased) : zero;
			if (strategy == O_STRATEGY_DIFFUSE_SPECULAR_SEPARATELY) {
				/* :565-586 */
				for (uint32_t s = 0; s != S; ++s) {
					v3 dd = sample_psa(&pd, cap, next_noise_2(f, k, noise), biased);
					dd = m43_mul_transposed(&ltc.world_to_shading, dd);
					v3 rb = radiance_visibility_brdf(ctx, NULL, NULL, dd, sd, light, 1, 0);
					result = add3(result, scale3(rb, pd.psa));
					if (ps.psa > 0.0f) {
						v3 dc = sample_psa(&ps, cap, next_noise_2(f, k, noise), biased);
						v3 ds = normalize3(m3_mul(&ltc.cosine_to_shading, dc));
						float ltc_density = evaluate_ltc_density(&ltc, ds, 1.0f);
						v3 rb2 = radiance_visibility_brdf(ctx, NULL, NULL, m43_mul_transposed(&ltc.world_to_shading, ds), sd, light, 0, 1);
						if (!(ds.z <= 0.0f || dc.z <= 0.0f))
							{
							/* vec3 * float * float / float, evaluated left to right (:584) */
							v3 t = scale3(scale3(rb2, ds.z), ps.psa);
							result = add3(result, mk3(t.x / ltc_density, t.y / ltc_density, t.z / ltc_density));
						}
					}
				}
			}
			else if (strategy == O_STRATEGY_DIFFUSE_SPECULAR_MIS) {
				/* :588-637 */
				int heuristic = f->mis_heuristic;
				v3 albedo = mk3(g_max(sd->diffuse_albedo.x, 0.01f), g_max(sd->diffuse_albedo.y, 0.01f), g_max(sd->diffuse_albedo.z, 0.01f));
				v3 diffuse_weight = scale3(albedo, pd.psa);
				uint32_t technique_count = (ps.psa > 0.0f) ? 2 : 1;
				float rcp_d = 1.0f / pd.psa;
				float rcp_s = 1.0f / ps.psa;
				v3 specular_weight_rgb = mk3(specular_weight, specular_weight, specular_weight);
				if (heuristic == O_MIS_OPTIMAL) {
					v3 radiance_over_pi = scale3(light->surface_radiance, O_INV_PI);
					diffuse_weight = mul3(diffuse_weight, radiance_over_pi);
					specular_weight_rgb = mul3(specular_weight_rgb, radiance_over_pi);
				}
				for (uint32_t s = 0; s != S; ++s) {
					v3 dir_d = sample_psa(&pd, cap, next_noise_2(f, k, noise), biased);
					v3 dir_s = zero;
					if (ps.psa > 0.0f) {
						dir_s = sample_psa(&ps, cap, next_noise_2(f, k, noise), biased);
						dir_s = normalize3(m3_mul(&ltc.cosine_to_shading, dir_s));
					}
					for (uint32_t j = 0; j != technique_count; ++j) {
						v3 ds = (j == 0) ? dir_d : dir_s;
						if (ds.z <= 0.0f) continue;
						float dens_d = ds.z * rcp_d;
						float dens_s = evaluate_ltc_density(&ltc, ds, rcp_s);
						int visibility;
						v3 rb = radiance_visibility_brdf(ctx, NULL, &visibility, m43_mul_transposed(&ltc.world_to_shading, ds), sd, light, 1, 1);
						v3 integrand = scale3(rb, ds.z);
						if (j == 0 && ps.psa <= 0.0f)
							result = add3(result, visibility ? scale3(integrand, 1.0f / dens_d) : zero);
						else if (j == 0)
							result = add3(result, mis_estimate(heuristic, integrand, diffuse_weight, dens_d, specular_weight_rgb, dens_s, k->mis_visibility_estimate));
						else
							result = add3(result, mis_estimate(heuristic, integrand, specular_weight_rgb, dens_s, diffuse_weight, dens_d, k->mis_visibility_estimate));
					}
				}
			}
			else if (strategy == O_STRATEGY_DIFFUSE_SPECULAR_RANDOM) {
				/* :639-670 */
				float lum = (sd->diffuse_albedo.x * 0.21263901f + sd->diffuse_albedo.y * 0.71516868f) + sd->diffuse_albedo.z * 0.07219232f;
				float diffuse_albedo = g_max(lum, 0.01f);
				float diffuse_weight = diffuse_albedo * pd.psa;
				float diffuse_ratio = diffuse_weight / (diffuse_weight + specular_weight);
				for (uint32_t s = 0; s != S; ++s) {
					v2 u = next_noise_2(f, k, noise);
					int specular_selected = u.x >= diffuse_ratio;
					float offset = specular_selected ? 1.0f : 0.0f;
					u.x = (u.x - offset) / (diffuse_ratio - offset);
					v3 ds = sample_psa(specular_selected ? &ps : &pd, cap, u, biased);
					if (specular_selected) ds = normalize3(m3_mul(&ltc.cosine_to_shading, ds));
					float lambert = ds.z;
					float dens_d = lambert * diffuse_albedo;
					float dens_s = evaluate_ltc_density(&ltc, ds, specular_albedo);
					float density = (dens_d + dens_s) / (diffuse_weight + specular_weight);
					v3 rb = radiance_visibility_brdf(ctx, &lambert, NULL, m43_mul_transposed(&ltc.world_to_shading, ds), sd, light, 1, 1);
					if (!(ds.z <= 0.0f)) {
						v3 t = scale3(rb, ds.z);
						result = add3(result, mk3(t.x / density, t.y / density, t.z / density));
					}
				}
			}
		}
	}

ggx_tail:
	if (strategy == O_STRATEGY_DIFFUSE_GGX_MIS) {
		/* :676-709 */
		v3 out_shading = m43_mul(&ltc.world_to_shading, sd->outgoing, 0.0f);
		out_shading.y = 0.0f;
		for (uint32_t s = 0; s != S; ++s) {
			float ggx_density;
			v3 dg = sample_ggx_reflected(&ggx_density, out_shading, sd->roughness, next_noise_2(f, k, noise));
			v3 dw = m43_mul_transposed(&ltc.world_to_shading, dg);
			if (dg.z > 0.0f && light_ray_intersection(light, vmax, sd->position, dw, 0.0f)) {
				float lambert;
				v3 rb = radiance_visibility_brdf(ctx, &lambert, NULL, dw, sd, light, 1, 1);
				float polygon_density = is_psa ? (lambert * density_factor) : density_factor;
				result = add3(result, scale3(scale3(rb, lambert), mis_weight_over_density(f->mis_heuristic, ggx_density, polygon_density)));
			}
		}
	}
	return scale3(result, 1.0f / (float) S);
}

/* main, shading_pass.frag.glsl:824-866 (up to the linear output) */
static void shade_pixel(pixel_ctx_t* ctx, uint32_t px, uint32_t py, float out[4]) {
	const oracle_frame_t* f = ctx->f;
	const frame_constants_t* k = ctx->k;
	uint32_t primitive = f->visibility[(size_t) py * f->width + px];
	v3 color = mk3(0.0f, 0.0f, 0.0f);
	float fx = (float) (int32_t) px, fy = (float) (int32_t) py;
	v3 ray = mk3(
		(k->pixel_to_ray[0][0] * fx + k->pixel_to_ray[0][1] * fy) + k->pixel_to_ray[0][2] * 1.0f,
		(k->pixel_to_ray[1][0] * fx + k->pixel_to_ray[1][1] * fy) + k->pixel_to_ray[1][2] * 1.0f,
		(k->pixel_to_ray[2][0] * fx + k->pixel_to_ray[2][1] * fy) + k->pixel_to_ray[2][2] * 1.0f);
	shading_data_t sd;
	memset(&sd, 0, sizeof(sd));
	v3 end_xyz = ray;
	float end_w = 0.0f;
	if (primitive != 0xFFFFFFFFu) {
		sd = get_shading_data(f, k, primitive, ray);
		end_xyz = sd.position;
		end_w = 1.0f;
	}
	if (f->show_polygonal_lights) {
		/* :841-850 */
		v3 view_dir = normalize3(ray);
		for (uint32_t i = 0; i != f->light_count; ++i) {
			light_view_t light = read_light(f->constants, i, f->max_light_vertex_count);
			if (light_ray_intersection(&light, f->max_light_vertex_count, k->camera_position, end_xyz, end_w))
				color = add3(color, polygon_radiance(f, view_dir, k->camera_position, &light));
		}
	}
	if (primitive != 0xFFFFFFFFu) {
		float fresnel_luminance = (sd.fresnel_0.x * 0.2126f + sd.fresnel_0.y * 0.7152f) + sd.fresnel_0.z * 0.0722f;
		ltc_t ltc = get_ltc_coefficients(f, k, fresnel_luminance, sd.roughness, sd.position, sd.normal, sd.outgoing);
		noise_accessor_t noise;
		memset(&noise, 0, sizeof(noise));
		noise.pixel[0] = px; noise.pixel[1] = py;
		for (uint32_t i = 0; i != f->light_count; ++i) {
			light_view_t light = read_light(f->constants, i, f->max_light_vertex_count);
			color = add3(color, evaluate_light(ctx, &sd, &ltc, &light, &noise));
		}
	}
	if (isnan(color.x) || isnan(color.y) || isnan(color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z))
		color = mk3(1.0f / k->exposure_factor, 0.0f / k->exposure_factor, 0.8f / k->exposure_factor);
	out[0] = color.x * k->exposure_factor;
	out[1] = color.y * k->exposure_factor;
	out[2] = color.z * k->exposure_factor;
	out[3] = 1.0f;
}

void oracle_shade_rows(const oracle_frame_t* frame, float* out_rgba, uint32_t y0, uint32_t y1, int thread_count) {
	frame_constants_t k = read_frame_constants(frame->constants);
	uint64_t rays = 0;
#ifdef _OPENMP
	if (thread_count > 0) omp_set_num_threads(thread_count);
#endif
	(void) thread_count;
	/* chunks of 64 consecutive pixels so that even a few rows keep every core busy */
	int64_t first = (int64_t) y0 * frame->width, last = (int64_t) y1 * frame->width;
	int64_t chunks = (last - first + 63) / 64;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays)
	for (int64_t c = 0; c < chunks; ++c) {
		pixel_ctx_t ctx = {frame, &k, 0};
		int64_t end = first + (c + 1) * 64 < last ? first + (c + 1) * 64 : last;
		for (int64_t i = first + c * 64; i < end; ++i)
			shade_pixel(&ctx, (uint32_t) (i % frame->width), (uint32_t) (i / frame->width), out_rgba + 4 * (size_t) i);
		rays += ctx.rays;
	}
	g_ray_count = rays;
}

/* one pixel (debugging aid: oracle/tools/nan_origin.py runs it with floating point traps) */
void oracle_shade_pixel(const oracle_frame_t* frame, uint32_t x, uint32_t y, float out_rgba[4]) {
	frame_constants_t k = read_frame_constants(frame->constants);
	pixel_ctx_t ctx = {frame, &k, 0};
	shade_pixel(&ctx, x, y, out_rgba);
}

void oracle_primary_visibility(const uint8_t* constants, const void* bvh, uint32_t width, uint32_t height, float near, float far, uint32_t* out) {
	frame_constants_t k = read_frame_constants(constants);
#pragma omp parallel for schedule(dynamic, 4)
	for (int64_t y = 0; y < (int64_t) height; ++y)
		for (uint32_t x = 0; x != width; ++x) {
			float fx = (float) x, fy = (float) y;
			float d[3] = {
				(k.pixel_to_ray[0][0] * fx + k.pixel_to_ray[0][1] * fy) + k.pixel_to_ray[0][2],
				(k.pixel_to_ray[1][0] * fx + k.pixel_to_ray[1][1] * fy) + k.pixel_to_ray[1][2],
				(k.pixel_to_ray[2][0] * fx + k.pixel_to_ray[2][1] * fy) + k.pixel_to_ray[2][2]};
			float o[3] = {k.camera_position.x, k.camera_position.y, k.camera_position.z};
			out[(size_t) y * width + x] = oracle_bvh_closest_front_hit(bvh, o, d, near, far);
		}
}

/* ---- output encodings, shading_pass.frag.glsl:871-892, srgb_utility.glsl --- */

static float linear_to_srgb(float c) {
	c = g_clamp(c, 0.0f, 1.0f);
	return (c <= 0.0031308f) ? (12.92f * c) : (1.055f * l_powf(c, 1.0f / 2.4f) - 0.055f);
}
static float srgb_to_linear(float c) {
	c = g_clamp(c, 0.0f, 1.0f);
	return (c <= 0.04045f) ? ((1.0f / 12.92f) * c) : l_powf(fmaf(c, 1.0f / 1.055f, 0.055f / 1.055f), 2.4f);
}
/* UNORM8 store of the render target: round to nearest */
static uint8_t to_unorm8(float c) {
	c = g_clamp(c, 0.0f, 1.0f);
	return (uint8_t) (c * 255.0f + 0.5f);
}

void oracle_encode_srgb8(const float* rgba, uint8_t* out, uint64_t pixel_count) {
	for (uint64_t i = 0; i != pixel_count; ++i) {
		for (int c = 0; c != 3; ++c) out[4 * i + c] = to_unorm8(linear_to_srgb(rgba[4 * i + c]));
		out[4 * i + 3] = to_unorm8(rgba[4 * i + 3]);
	}
}

/* packHalf2x16 semantics: round to nearest even, overflow to infinity */
static uint16_t float_to_half(float f) {
	uint32_t x = f2u(f);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t mant = x & 0x007FFFFFu;
	int32_t exp = (int32_t) ((x >> 23) & 0xFF);
	if (exp == 255) return (uint16_t) (sign | 0x7C00u | (mant ? 0x200u : 0));
	int32_t e = exp - 127 + 15;
	if (e >= 31) return (uint16_t) (sign | 0x7C00u);
	if (e <= 0) {
		if (e < -10) return (uint16_t) sign;
		mant |= 0x00800000u;
		uint32_t shift = (uint32_t) (14 - e);
		uint32_t half_mant = mant >> shift;
		uint32_t rem = mant & ((1u << shift) - 1), halfway = 1u << (shift - 1);
		if (rem > halfway || (rem == halfway && (half_mant & 1))) ++half_mant;
		return (uint16_t) (sign | half_mant);
	}
	uint32_t half = sign | ((uint32_t) e << 10) | (mant >> 13);
	uint32_t rem = mant & 0x1FFFu;
	if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;
	return (uint16_t) half;
}

void oracle_encode_half_bits(const float* rgba, uint8_t* out, uint64_t pixel_count, uint32_t frame_bits, int output_linear_rgb) {
	uint32_t mask = (frame_bits == 1) ? 0xFF : 0xFF00, shift = (frame_bits == 1) ? 0 : 8;
	for (uint64_t i = 0; i != pixel_count; ++i) {
		uint32_t h0 = (uint32_t) float_to_half(rgba[4 * i + 0]) | ((uint32_t) float_to_half(rgba[4 * i + 1]) << 16);
		uint32_t h1 = (uint32_t) float_to_half(rgba[4 * i + 2]) | ((uint32_t) float_to_half(rgba[4 * i + 3]) << 16);
		float c[3] = {
			(float) ((h0 & mask) >> shift) * (1.0f / 255.0f),
			(float) ((((h0 & 0xFFFF0000u) >> 16) & mask) >> shift) * (1.0f / 255.0f),
			(float) ((h1 & mask) >> shift) * (1.0f / 255.0f)};
		for (int j = 0; j != 3; ++j) {
			/* with an *_SRGB target the shader pre-applies the inverse transfer and the
			   hardware re-applies the forward one on store; the stored byte is the same */
			float v = output_linear_rgb ? linear_to_srgb(srgb_to_linear(c[j])) : c[j];
			out[4 * i + j] = to_unorm8(v);
		}
		out[4 * i + 3] = 255;
	}
}

/* ------------------------------------------------------------------------ */
/* test entry points                                                        */

uint32_t oracle_clip_polygon(uint32_t vertex_count, uint32_t min_count, uint32_t max_count, float* vertices) {
	v3 v[O_CAP];
	memset(v, 0, sizeof(v));
	for (uint32_t i = 0; i != max_count; ++i) v[i] = mk3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	uint32_t r = clip_polygon(vertex_count, min_count, max_count, v);
	for (uint32_t i = 0; i != max_count; ++i) { vertices[3 * i] = v[i].x; vertices[3 * i + 1] = v[i].y; vertices[3 * i + 2] = v[i].z; }
	return r;
}

static void psa_to_state(const psa_polygon_t* p, float* s) {
	s[0] = (float) p->vertex_count;
	for (int i = 0; i != O_CAP; ++i) {
		s[1 + 2 * i] = p->vertices[i].x; s[2 + 2 * i] = p->vertices[i].y;
		s[19 + 2 * i] = p->ellipses[i].x; s[20 + 2 * i] = p->ellipses[i].y;
		s[39 + i] = p->sector_psa[i];
	}
	s[37] = p->inner_ellipse_0.x; s[38] = p->inner_ellipse_0.y;
	s[48] = p->psa;
}
static psa_polygon_t psa_from_state(const float* s) {
	psa_polygon_t p;
	p.vertex_count = (uint32_t) s[0];
	for (int i = 0; i != O_CAP; ++i) {
		p.vertices[i] = mk2(s[1 + 2 * i], s[2 + 2 * i]);
		p.ellipses[i] = mk2(s[19 + 2 * i], s[20 + 2 * i]);
		p.sector_psa[i] = s[39 + i];
	}
	p.inner_ellipse_0 = mk2(s[37], s[38]);
	p.psa = s[48];
	return p;
}

void oracle_psa_prepare(uint32_t vertex_count, uint32_t max_count, const float* vertices, float* state) {
	v3 v[O_CAP];
	memset(v, 0, sizeof(v));
	for (uint32_t i = 0; i != max_count; ++i) v[i] = mk3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	psa_polygon_t p = prepare_psa(vertex_count, max_count, v, 0);
	psa_to_state(&p, state);
}
void oracle_psa_sample(const float* state, uint32_t max_count, float u0, float u1, int biased, float out_dir[3]) {
	psa_polygon_t p = psa_from_state(state);
	v3 d = sample_psa(&p, max_count, mk2(u0, u1), biased);
	out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
}
void oracle_psa_error(const float* state, uint32_t max_count, float u0, float u1, const float dir[3], float out_error[3]) {
	psa_polygon_t p = psa_from_state(state);
	v3 e = psa_sampling_error(&p, max_count, mk2(u0, u1), mk3(dir[0], dir[1], dir[2]));
	out_error[0] = e.x; out_error[1] = e.y; out_error[2] = e.z;
}
float oracle_solid_angle_sample(uint32_t vertex_count, uint32_t max_count, const float* vertices, const float sp[3], float u0, float u1, float out_dir[3]) {
	v3 v[O_CAP];
	memset(v, 0, sizeof(v));
	for (uint32_t i = 0; i != max_count; ++i) v[i] = mk3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	sa_polygon_t p = prepare_sa(vertex_count, max_count, v, mk3(sp[0], sp[1], sp[2]));
	v3 d = sample_sa(&p, max_count, mk2(u0, u1));
	out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
	return p.solid_angle;
}
float oracle_atan(float x) { return o_atan(x); }
float oracle_acos_unit(float x) { return o_acos_unit(x); }
float oracle_rsqrt(float x) { return rsqrt_f(x); }
float oracle_log2(float x) { return o_log2(x); }
float oracle_atan2(float y, float x) { return o_atan2(y, x); }
float oracle_pow_third(float x) { return o_pow_third(x); }
void oracle_sincos(float x, float* s, float* c) { o_sincos(x, s, c); }
float oracle_fast_positive_atan(float x) { return fast_positive_atan(x); }
float oracle_kahan(float a, float b, float c, float d) { return kahan(a, b, c, d); }
void oracle_decode_position(uint32_t q0, uint32_t q1, const float factor[3], const float summand[3], float out[3]) {
	v3 p = decode_position(q0, q1, mk3(factor[0], factor[1], factor[2]), mk3(summand[0], summand[1], summand[2]));
	out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
void oracle_decode_normal(uint16_t x, uint16_t y, float out[3]) {
	v3 n = decode_normal((float) x / 65535.0f, (float) y / 65535.0f);
	out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
void oracle_ltc_coefficients(const oracle_frame_t* frame, float fresnel_0, float roughness, const float p[3], const float n[3], const float o[3], float out[44]) {
	frame_constants_t k = read_frame_constants(frame->constants);
	ltc_t l = get_ltc_coefficients(frame, &k, fresnel_0, roughness, mk3(p[0], p[1], p[2]), mk3(n[0], n[1], n[2]), mk3(o[0], o[1], o[2]));
	memcpy(out, &l.world_to_shading, 48);
	memcpy(out + 12, &l.shading_to_cosine, 36);
	memcpy(out + 21, &l.world_to_cosine, 48);
	memcpy(out + 33, &l.cosine_to_shading, 36);
	out[42] = l.albedo;
	out[43] = l.determinant;
}
static v3 pixel_ray(const frame_constants_t* k, uint32_t px, uint32_t py) {
	float fx = (float) (int32_t) px, fy = (float) (int32_t) py;
	return mk3(
		(k->pixel_to_ray[0][0] * fx + k->pixel_to_ray[0][1] * fy) + k->pixel_to_ray[0][2] * 1.0f,
		(k->pixel_to_ray[1][0] * fx + k->pixel_to_ray[1][1] * fy) + k->pixel_to_ray[1][2] * 1.0f,
		(k->pixel_to_ray[2][0] * fx + k->pixel_to_ray[2][1] * fy) + k->pixel_to_ray[2][2] * 1.0f);
}
void oracle_shading_data(const oracle_frame_t* frame, uint32_t x, uint32_t y, float out[17]) {
	frame_constants_t k = read_frame_constants(frame->constants);
	uint32_t primitive = frame->visibility[(size_t) y * frame->width + x];
	memset(out, 0, 17 * sizeof(float));
	if (primitive == 0xFFFFFFFFu) return;
	shading_data_t sd = get_shading_data(frame, &k, primitive, pixel_ray(&k, x, y));
	memcpy(out, &sd, 17 * sizeof(float));
}
void oracle_evaluate_brdf(const float shading_data[17], const float incoming[3], int diffuse, int specular, float out_rgb[3]) {
	shading_data_t sd;
	memcpy(&sd, shading_data, 17 * sizeof(float));
	v3 b = evaluate_brdf(&sd, mk3(incoming[0], incoming[1], incoming[2]), diffuse, specular);
	out_rgb[0] = b.x; out_rgb[1] = b.y; out_rgb[2] = b.z;
}
void oracle_noise_stream(const oracle_frame_t* frame, uint32_t x, uint32_t y, uint32_t count, float* out_pairs) {
	frame_constants_t k = read_frame_constants(frame->constants);
	noise_accessor_t a;
	memset(&a, 0, sizeof(a));
	a.pixel[0] = x; a.pixel[1] = y;
	for (uint32_t i = 0; i != count; ++i) {
		v2 u = next_noise_2(frame, &k, &a);
		out_pairs[2 * i] = u.x; out_pairs[2 * i + 1] = u.y;
	}
}
