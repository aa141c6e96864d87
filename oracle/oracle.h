/* TEST INFRASTRUCTURE — not part of the product.
 *
 * CPU oracle for the shading pass of MomentsInGraphics/vulkan_renderer: a plain
 * C99 restatement of the reference's GLSL (src/shaders/shading_pass.frag.glsl and
 * its includes).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (vulkan_renderer_amd/)
 * never does.
 *
 * Pinning status: the reference ships no tests or golden data and its shaders
 * cannot run in a container without Vulkan.  The oracle is pinned instead
 * against the reference's GLSL compiled as C++ through a compatibility header
 * (oracle/_ref, built by oracle/Makefile from the sources where they lie under
 * /root/reference) and against the golden vectors generated from it
 * (tests/golden/, script tests/golden/make_golden.py). */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* values follow the reference enums: src/main.h:45-89, src/polygonal_light.h:30-69 */
enum { O_STRATEGY_DIFFUSE_ONLY = 0, O_STRATEGY_DIFFUSE_GGX_MIS = 1, O_STRATEGY_DIFFUSE_SPECULAR_SEPARATELY = 2,
	O_STRATEGY_DIFFUSE_SPECULAR_MIS = 3, O_STRATEGY_DIFFUSE_SPECULAR_RANDOM = 4 };
enum { O_MIS_BALANCE = 0, O_MIS_POWER = 1, O_MIS_WEIGHTED = 2, O_MIS_OPTIMAL_CLAMPED = 3, O_MIS_OPTIMAL = 4 };
enum { O_TECHNIQUE_BASELINE = 0, O_TECHNIQUE_AREA_TURK = 1, O_TECHNIQUE_RECTANGLE_SOLID_ANGLE_URENA = 2, O_TECHNIQUE_SOLID_ANGLE_ARVO = 3,
	O_TECHNIQUE_BILINEAR_COSINE_WARP_HART = 6, O_TECHNIQUE_BILINEAR_COSINE_WARP_CLIPPING_HART = 7,
	O_TECHNIQUE_BIQUADRATIC_COSINE_WARP_HART = 8, O_TECHNIQUE_BIQUADRATIC_COSINE_WARP_CLIPPING_HART = 9,
	O_TECHNIQUE_PROJECTED_SOLID_ANGLE_ARVO = 10, O_TECHNIQUE_SOLID_ANGLE = 4, O_TECHNIQUE_CLIPPED_SOLID_ANGLE = 5,
	O_TECHNIQUE_PROJECTED_SOLID_ANGLE = 11, O_TECHNIQUE_PROJECTED_SOLID_ANGLE_BIASED = 12 };

/* One material texture (reference: a VkImage with its mip chain, scene.c:486-559): RGBA8 texels,
 * mip 0 first, every level tightly packed, rows top to bottom.  The filter that stands in for the
 * driver's sampler (unpinned by the reference) is oracle_sample_texture below. */
typedef struct oracle_texture_s {
	const uint8_t* texels;
	uint32_t width, height, mip_count;
	/* 1: the colour channels are sRGB encoded (VK_FORMAT_*_SRGB) */
	uint32_t srgb;
} oracle_texture_t;

/* One light texture (reference: g_light_textures[], shading_pass.frag.glsl:61, loaded by
 * create_and_assign_light_textures main.c:371-417): level 0 only, because the shader reads it with
 * textureLod(..., 0.0f) (:182); RGBA fp32 texels, rows top to bottom.  width 0 = white. */
typedef struct oracle_light_texture_s {
	const float* texels;
	uint32_t width, height;
} oracle_light_texture_t;

/* Everything the per-pixel program reads.  Buffers are byte-identical to what
 * the product uploads to the GPU. */
typedef struct oracle_frame_s {
	/* per_frame_constants_t followed by the packed light array
	 * (reference layout: src/main.h:488-505, src/main.c:2159-2187) */
	const uint8_t* constants;
	uint32_t light_count;
	uint32_t max_light_vertex_count;
	/* mesh buffers of the .vks file (src/scene.h:47-76) */
	const uint32_t* quantized_positions;
	const uint16_t* normals_and_tex_coords;
	const uint8_t* material_indices;
	uint64_t triangle_count;
	/* constant stand-ins for the three material textures: per material
	 * base_color.rgb, specular.rgb (occlusion, linear roughness, metalicity),
	 * normal.xy; 8 floats each */
	const float* material_constants;
	uint32_t material_count;
	/* R32_UINT primitive index per pixel, 0xFFFFFFFF = background */
	const uint32_t* visibility;
	uint32_t width, height;
	/* LTC tables as quantised by load_ltc_table (src/ltc_table.c:82-116) */
	const uint16_t* ltc_rgba;
	const uint16_t* ltc_rg;
	uint32_t ltc_resolution, ltc_fresnel_count;
	/* noise table RGBA16_UNORM, layer major (src/noise_table.c:46-106) */
	const uint16_t* noise;
	uint32_t noise_width, noise_height, noise_depth;
	/* compile-time switches of the reference shader (src/main.c:752-792) */
	int32_t sampling_strategies, mis_heuristic, polygon_technique;
	uint32_t sample_count;
	int32_t trace_shadow_rays, show_polygonal_lights;
	/* opaque BVH handle from oracle_bvh_build (NULL: brute force over all triangles) */
	const void* bvh;
	int32_t brute_force_rays;
	/* ERROR_DISPLAY_DIFFUSE / _SPECULAR / ERROR_INDEX of the reference (main.c:728-750):
	 * error_display 0 none, 1 diffuse, 2 specular; error_index 0 backward, 1 backward
	 * times projected solid angle, 2 forward */
	int32_t error_display, error_index;
	/* 3 textures per material (base colour, specular, normal) or NULL: then material_constants
	 * stand for constant textures and the texture coordinate derivatives are not computed */
	const oracle_texture_t* material_textures;
	/* textures that polygonal_light_t.texture_index selects, or NULL (then every one is white) */
	const oracle_light_texture_t* light_textures;
	uint32_t light_texture_count;
} oracle_frame_t;

/* textureGrad() of this build: anisotropic filtering as the Vulkan specification sketches it, repeat addressing.
 * N = min(ceil(P_max / P_min), 16, ceil(P_max)) trilinear taps along the longer axis of the footprint (P_max, P_min: the
 * lengths of the two screen-space derivative vectors in texels), level of detail = log2(P_max / N) clamped to the mip
 * chain, averaged in order; N = 1 is plain trilinear.  Bilinear weights in exact fp32, x first; sRGB texels are decoded
 * before filtering.  (Rounds 1 - 4: isotropic trilinear.) */
void oracle_sample_texture(const oracle_texture_t* texture, const float uv[2], const float duv_dx[2], const float duv_dy[2], float out_rgba[4]);
/* textureLod(g_light_textures[i], uv, 0) of this build (sampler of main.c:611-621: linear filter,
 * u repeats, v clamps to the edge): bilinear in exact fp32, x first.  u is wrapped to [0,1) before
 * scaling; non-finite coordinates read texel column / row 0. */
void oracle_sample_light_texture(const oracle_light_texture_t* texture, const float uv[2], float out_rgba[4]);
/* the sRGB -> linear table that the sampler uses (256 floats); the product uploads the same table */
const float* oracle_srgb_table(void);

/* Shades rows [y0, y1) of the frame into out_rgba (width*height*4 floats,
 * vec4(final_color * exposure, 1), reference shading_pass.frag.glsl:866).
 * thread_count <= 0 uses all OpenMP threads. */
void oracle_shade_rows(const oracle_frame_t* frame, float* out_rgba, uint32_t y0, uint32_t y1, int thread_count);
/* One pixel of the frame (debugging aid) */
void oracle_shade_pixel(const oracle_frame_t* frame, uint32_t x, uint32_t y, float out_rgba[4]);
/* Number of shadow rays traced by the last oracle_shade_rows call */
uint64_t oracle_last_ray_count(void);
/* 0 = libm transcendental functions, 1 = the polynomial forms shared with the GPU */
void oracle_set_math_mode(int mode);
/* where the transcendentals of math mode 0 come from: 0 (default) the restatement of glibc 2.35 in
 * vulkan_renderer_amd/csrc/glibc_math.h (the same on every machine), 1 the C library of this machine */
void oracle_set_libm_source(int use_system_library);
int oracle_get_libm_source(void);
/* diagnostics: print every shadow ray (origin, direction, t_max, the light's plane, blocked) to stdout */
void oracle_set_ray_log(int on);

/* Output encodings of the reference (shading_pass.frag.glsl:871-892) */
void oracle_encode_srgb8(const float* rgba, uint8_t* out_rgba8, uint64_t pixel_count);
void oracle_encode_half_bits(const float* rgba, uint8_t* out_rgba8, uint64_t pixel_count, uint32_t frame_bits, int output_linear_rgb);

/* BVH over the de-quantised triangle soup (contract: reference scene.c:176-187
 * for de-quantisation, shading_pass.frag.glsl:120-138 for the ray query) */
void* oracle_bvh_build(const uint32_t* quantized_positions, uint64_t triangle_count, const float dequantization_factor[3], const float dequantization_summand[3]);
void oracle_bvh_destroy(void* bvh);
uint32_t oracle_bvh_closest_front_hit(const void* bvh, const float origin[3], const float dir[3], float t_min, float t_max);
/* visibility buffer through pixel centres: closest front-facing hit with view depth in [near, far] */
void oracle_primary_visibility(const uint8_t* constants, const void* bvh, uint32_t width, uint32_t height, float near, float far, uint32_t* out_primitives);
int oracle_bvh_any_hit(const void* bvh, const float origin[3], const float dir[3], float t_min, float t_max, int brute_force);

/* ---- entry points for unit / property tests ---------------------------- */
/* vertices: 3 floats each, capacity max_count; returns the clipped count */
uint32_t oracle_clip_polygon(uint32_t vertex_count, uint32_t min_count, uint32_t max_count, float* vertices);
/* state layout: [0]=vertex_count, [1..18]=vertices (9x2), [19..36]=ellipses (9x2),
 * [37..38]=inner_ellipse_0, [39..47]=sector areas, [48]=projected solid angle */
#define ORACLE_PSA_STATE_FLOATS 49
void oracle_psa_prepare(uint32_t vertex_count, uint32_t max_count, const float* vertices, float* state);
void oracle_psa_sample(const float* state, uint32_t max_count, float u0, float u1, int biased, float out_dir[3]);
void oracle_psa_error(const float* state, uint32_t max_count, float u0, float u1, const float dir[3], float out_error[3]);
/* solid angle sampler: returns the solid angle, writes a direction */
float oracle_solid_angle_sample(uint32_t vertex_count, uint32_t max_count, const float* vertices, const float shading_position[3], float u0, float u1, float out_dir[3]);
/* math */
float oracle_atan(float x);
float oracle_acos_unit(float x);
float oracle_rsqrt(float x);
float oracle_log2(float x);
float oracle_atan2(float y, float x);
float oracle_pow_third(float x);
void oracle_sincos(float x, float* s, float* c);
float oracle_fast_positive_atan(float x);
float oracle_kahan(float a, float b, float c, float d);
void oracle_decode_position(uint32_t q0, uint32_t q1, const float factor[3], const float summand[3], float out[3]);
void oracle_decode_normal(uint16_t x, uint16_t y, float out[3]);
/* LTC lookup + frame construction; out = 12 (world_to_shading) + 9 (shading_to_cosine)
 * + 12 (world_to_cosine) + 9 (cosine_to_shading) + albedo + determinant = 44 floats */
void oracle_ltc_coefficients(const oracle_frame_t* frame, float fresnel_0, float roughness, const float position[3], const float normal[3], const float outgoing[3], float out[44]);
/* shading data for one pixel: position(3) normal(3) outgoing(3) lambert_outgoing diffuse_albedo(3) fresnel_0(3) roughness = 17 floats */
void oracle_shading_data(const oracle_frame_t* frame, uint32_t x, uint32_t y, float out[17]);
void oracle_evaluate_brdf(const float shading_data[17], const float incoming[3], int diffuse, int specular, float out_rgb[3]);
/* first 'count' pairs of the noise stream of one pixel */
void oracle_noise_stream(const oracle_frame_t* frame, uint32_t x, uint32_t y, uint32_t count, float* out_pairs);

#ifdef __cplusplus
}
#endif
#endif
