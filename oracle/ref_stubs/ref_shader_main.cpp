// TEST INFRASTRUCTURE - not part of the product.
//
// Harness around the reference's fragment shader compiled as C++ (see
// glsl_compat.hpp and build_ref_shaders.py).  REF_SHADER_SOURCE is the path of the
// pre-processed copy of src/shaders/shading_pass.frag.glsl in a temporary
// directory; the variant (sampling strategy, heuristic, counts ...) is fixed by the
// same -D defines the reference passes to glslangValidator (src/main.c:752-792).
// Exports ref_shade_rows() with the oracle's frame description, so one Python
// binding drives both.
#include "glsl_compat.hpp"
#include "../oracle.h"

namespace glsl {
thread_local int g_current_pixel_x = 0, g_current_pixel_y = 0;
unsigned long long g_shadow_ray_count = 0;
#include REF_SHADER_SOURCE
}  // namespace glsl

using namespace glsl;

static float rd_f(const uint8_t* p, size_t o) { float f; std::memcpy(&f, p + o, 4); return f; }
static uint32_t rd_u(const uint8_t* p, size_t o) { uint32_t u; std::memcpy(&u, p + o, 4); return u; }
static vec3 rd_v3(const uint8_t* p, size_t o) { return vec3(rd_f(p, o), rd_f(p, o + 4), rd_f(p, o + 8)); }

// Fills the uniform block from the byte image of write_constants().  The block is
// std140 / row_major: C rows become GLSL rows, i.e. m[column][row] = bytes[row][column].
static void set_uniforms(const oracle_frame_t* f) {
	const uint8_t* c = f->constants;
	g_mesh_dequantization_factor = rd_v3(c, 0);
	g_mesh_dequantization_summand = rd_v3(c, 16);
	g_error_factor = rd_f(c, 28);
	for (int row = 0; row != 3; ++row)
		for (int col = 0; col != 3; ++col)
			g_pixel_to_ray_direction_world_space[col][row] = rd_f(c, 96 + 16 * row + 4 * col);
	g_camera_position_world_space = rd_v3(c, 144);
	g_mis_visibility_estimate = rd_f(c, 156);
	g_viewport_size = uvec2(rd_u(c, 160), rd_u(c, 164));
	g_exposure_factor = rd_f(c, 176);
	g_roughness_factor = rd_f(c, 180);
	g_noise_resolution_mask = uvec2(rd_u(c, 184), rd_u(c, 188));
	g_noise_texture_index_mask = rd_u(c, 192);
	g_frame_bits = rd_u(c, 196);
	g_noise_random_numbers = uvec4(rd_u(c, 208), rd_u(c, 212), rd_u(c, 216), rd_u(c, 220));
	g_ltc_constants.fresnel_index_factor = rd_f(c, 224);
	g_ltc_constants.fresnel_index_summand = rd_f(c, 228);
	g_ltc_constants.roughness_factor = rd_f(c, 232);
	g_ltc_constants.roughness_summand = rd_f(c, 236);
	g_ltc_constants.inclination_factor = rd_f(c, 240);
	g_ltc_constants.inclination_summand = rd_f(c, 244);
	const size_t vmax = MAX_POLYGONAL_LIGHT_VERTEX_COUNT;
	const size_t stride = 160 + 16 * vmax * 2 + 16 * (vmax - 2);
	for (uint32_t i = 0; i != f->light_count && i != POLYGONAL_LIGHT_ARRAY_SIZE; ++i) {
		const uint8_t* p = c + 256 + stride * i;
		polygonal_light_t& l = g_polygonal_lights[i];
		l.rotation_angles = rd_v3(p, 0); l.scaling_x = rd_f(p, 12);
		l.translation = rd_v3(p, 16); l.scaling_y = rd_f(p, 28);
		l.radiant_flux = rd_v3(p, 32); l.inv_scaling_x = rd_f(p, 44);
		l.surface_radiance = rd_v3(p, 48); l.inv_scaling_y = rd_f(p, 60);
		l.plane = vec4(rd_f(p, 64), rd_f(p, 68), rd_f(p, 72), rd_f(p, 76));
		l.vertex_count = rd_u(p, 80); l.texturing_technique = rd_u(p, 84); l.texture_index = rd_u(p, 88);
		for (int row = 0; row != 3; ++row)
			for (int col = 0; col != 3; ++col)
				l.rotation[col][row] = rd_f(p, 96 + 16 * row + 4 * col);
		l.area = rd_f(p, 144); l.rcp_area = rd_f(p, 148);
		for (size_t v = 0; v != vmax; ++v) {
			l.vertices_plane_space[v] = vec2(rd_f(p, 160 + 16 * v), rd_f(p, 164 + 16 * v));
			l.vertices_world_space[v] = rd_v3(p, 160 + 16 * vmax + 16 * v);
		}
		for (size_t v = 0; v + 2 != vmax; ++v)
			l.fan_areas[v] = vec2(rd_f(p, 160 + 32 * vmax + 16 * v), rd_f(p, 164 + 32 * vmax + 16 * v));
	}
}

extern "C" int ref_variant_matches(const oracle_frame_t* f) {
	// the variant is compiled in; refuse frames that were set up for another one
	int strategy = SAMPLING_STRATEGIES_DIFFUSE_ONLY ? 0 : SAMPLING_STRATEGIES_DIFFUSE_GGX_MIS ? 1 : SAMPLING_STRATEGIES_DIFFUSE_SPECULAR_SEPARATELY ? 2
		: SAMPLING_STRATEGIES_DIFFUSE_SPECULAR_MIS ? 3 : 4;
	int heuristic = MIS_HEURISTIC_BALANCE ? 0 : MIS_HEURISTIC_POWER ? 1 : MIS_HEURISTIC_WEIGHTED ? 2 : MIS_HEURISTIC_OPTIMAL_CLAMPED ? 3 : 4;
	return f->sampling_strategies == strategy && f->mis_heuristic == heuristic && f->sample_count == SAMPLE_COUNT
		&& f->light_count == POLYGONAL_LIGHT_COUNT && f->max_light_vertex_count == MAX_POLYGONAL_LIGHT_VERTEX_COUNT
		&& (f->trace_shadow_rays != 0) == (TRACE_SHADOW_RAYS != 0) && (f->show_polygonal_lights != 0) == (SHOW_POLYGONAL_LIGHTS != 0)
		&& f->material_count == MATERIAL_COUNT;
}

// Runs main() of the reference shader for rows [y0, y1).  With frame_bits == 0 and
// OUTPUT_LINEAR_RGB == 1 g_out_color is the linear value of
// shading_pass.frag.glsl:866; otherwise it is the encoded colour.
extern "C" void ref_shade_rows(const oracle_frame_t* f, float* out_rgba, uint32_t y0, uint32_t y1) {
	set_uniforms(f);
	g_quantized_vertex_positions.data = f->quantized_positions; g_quantized_vertex_positions.kind = 0;
	g_packed_normals_and_tex_coords.data = f->normals_and_tex_coords;
	g_material_indices.data = f->material_indices; g_material_indices.kind = 1;
	g_visibility_buffer.data = f->visibility; g_visibility_buffer.width = (int) f->width;
	for (uint32_t m = 0; m != f->material_count && m != MATERIAL_COUNT; ++m) {
		const float* k = f->material_constants + 8 * (size_t) m;
		g_material_textures[3 * m + 0].constant = vec4(k[0], k[1], k[2], 1.0f);
		g_material_textures[3 * m + 1].constant = vec4(k[3], k[4], k[5], 1.0f);
		g_material_textures[3 * m + 2].constant = vec4(k[6], k[7], 1.0f, 1.0f);
		for (int t = 0; t != 3; ++t) g_material_textures[3 * m + t].texture = f->material_textures ? (const void*) &f->material_textures[3 * m + t] : nullptr;
	}
	for (int t = 0; t != LIGHT_TEXTURE_COUNT; ++t) {
		g_light_textures[t].constant = vec4(1.0f, 1.0f, 1.0f, 1.0f);
		g_light_textures[t].light_texture = (f->light_textures && (uint32_t) t < f->light_texture_count) ? (const void*) &f->light_textures[t] : nullptr;
	}
	g_noise_table.data = f->noise; g_noise_table.width = (int) f->noise_width; g_noise_table.height = (int) f->noise_height; g_noise_table.depth = (int) f->noise_depth;
	g_ltc_tables[0].data = f->ltc_rgba; g_ltc_tables[0].channels = 4;
	g_ltc_tables[1].data = f->ltc_rg; g_ltc_tables[1].channels = 2;
	for (int i = 0; i != 2; ++i) { g_ltc_tables[i].resolution = (int) f->ltc_resolution; g_ltc_tables[i].layers = (int) f->ltc_fresnel_count; }
#if TRACE_SHADOW_RAYS
	g_top_level_acceleration_structure.bvh = f->bvh;
	g_top_level_acceleration_structure.brute_force = f->brute_force_rays;
#endif
	g_shadow_ray_count = 0;
	for (uint32_t y = y0; y != y1; ++y)
		for (uint32_t x = 0; x != f->width; ++x) {
			g_current_pixel_x = (int) x; g_current_pixel_y = (int) y;
			gl_FragCoord = vec4((float) x + 0.5f, (float) y + 0.5f, 0.0f, 1.0f);
			shader_main();
			float* o = out_rgba + 4 * ((size_t) y * f->width + x);
			o[0] = g_out_color.x; o[1] = g_out_color.y; o[2] = g_out_color.z; o[3] = g_out_color.w;
		}
}

extern "C" unsigned long long ref_last_ray_count(void) { return g_shadow_ray_count; }

// ---- sub-function entry points (layouts as in oracle.h) ---------------------------------
extern "C" uint32_t ref_clip_polygon(uint32_t vertex_count, float* vertices) {
	vec3 v[MAX_POLYGON_VERTEX_COUNT];
	for (int i = 0; i != MAX_POLYGON_VERTEX_COUNT; ++i) v[i] = vec3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	uint32_t r = clip_polygon(vertex_count, v);
	for (int i = 0; i != MAX_POLYGON_VERTEX_COUNT; ++i) { vertices[3 * i] = v[i].x; vertices[3 * i + 1] = v[i].y; vertices[3 * i + 2] = v[i].z; }
	return r;
}
static void to_state(const projected_solid_angle_polygon_t& p, float* s) {
	for (int i = 0; i != ORACLE_PSA_STATE_FLOATS; ++i) s[i] = 0.0f;
	s[0] = (float) p.vertex_count;
	for (int i = 0; i != MAX_POLYGON_VERTEX_COUNT; ++i) {
		s[1 + 2 * i] = p.vertices[i].x; s[2 + 2 * i] = p.vertices[i].y;
		s[19 + 2 * i] = p.ellipses[i].x; s[20 + 2 * i] = p.ellipses[i].y;
		s[39 + i] = p.sector_projected_solid_angles[i];
	}
	s[37] = p.inner_ellipse_0.x; s[38] = p.inner_ellipse_0.y;
	s[48] = p.projected_solid_angle;
}
static projected_solid_angle_polygon_t from_state(const float* s) {
	projected_solid_angle_polygon_t p;
	p.vertex_count = (uint) s[0];
	for (int i = 0; i != MAX_POLYGON_VERTEX_COUNT; ++i) {
		p.vertices[i] = vec2(s[1 + 2 * i], s[2 + 2 * i]);
		p.ellipses[i] = vec2(s[19 + 2 * i], s[20 + 2 * i]);
		p.sector_projected_solid_angles[i] = s[39 + i];
	}
	p.inner_ellipse_0 = vec2(s[37], s[38]);
	p.projected_solid_angle = s[48];
	return p;
}
extern "C" void ref_psa_prepare(uint32_t vertex_count, const float* vertices, float* state) {
	vec3 v[MAX_POLYGON_VERTEX_COUNT];
	for (int i = 0; i != MAX_POLYGON_VERTEX_COUNT; ++i) v[i] = vec3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	to_state(prepare_projected_solid_angle_polygon_sampling(vertex_count, v), state);
}
extern "C" void ref_psa_sample(const float* state, float u0, float u1, float out_dir[3]) {
	vec3 d = sample_projected_solid_angle_polygon(from_state(state), vec2(u0, u1));
	out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
}
extern "C" void ref_psa_error(const float* state, float u0, float u1, const float dir[3], float out_error[3]) {
	vec3 e = compute_projected_solid_angle_polygon_sampling_error(from_state(state), vec2(u0, u1), vec3(dir[0], dir[1], dir[2]));
	out_error[0] = e.x; out_error[1] = e.y; out_error[2] = e.z;
}
extern "C" float ref_solid_angle_sample(uint32_t vertex_count, const float* vertices, const float sp[3], float u0, float u1, float out_dir[3]) {
	vec3 v[MAX_POLYGON_VERTEX_COUNT];
	for (int i = 0; i != MAX_POLYGON_VERTEX_COUNT; ++i) v[i] = vec3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	solid_angle_polygon_t p = prepare_solid_angle_polygon_sampling(vertex_count, v, vec3(sp[0], sp[1], sp[2]));
	vec3 d = sample_solid_angle_polygon(p, vec2(u0, u1));
	out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
	return p.solid_angle;
}
extern "C" float ref_fast_positive_atan(float x) { return fast_positive_atan(x); }
extern "C" float ref_kahan(float a, float b, float c, float d) { return kahan(a, b, c, d); }
extern "C" void ref_decode_position(uint32_t q0, uint32_t q1, const float f[3], const float s[3], float out[3]) {
	vec3 p = decode_position_64_bit(uvec2(q0, q1), vec3(f[0], f[1], f[2]), vec3(s[0], s[1], s[2]));
	out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
extern "C" void ref_decode_normal(uint16_t x, uint16_t y, float out[3]) {
	vec3 n = decode_normal_32_bit(vec2((float) x / 65535.0f, (float) y / 65535.0f));
	out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
extern "C" void ref_evaluate_brdf(const float sd[17], const float incoming[3], int diffuse, int specular, float out_rgb[3]) {
	shading_data_t d;
	d.position = vec3(sd[0], sd[1], sd[2]); d.normal = vec3(sd[3], sd[4], sd[5]); d.outgoing = vec3(sd[6], sd[7], sd[8]);
	d.lambert_outgoing = sd[9]; d.diffuse_albedo = vec3(sd[10], sd[11], sd[12]); d.fresnel_0 = vec3(sd[13], sd[14], sd[15]); d.roughness = sd[16];
	vec3 b = evaluate_brdf(d, vec3(incoming[0], incoming[1], incoming[2]), diffuse != 0, specular != 0);
	out_rgb[0] = b.x; out_rgb[1] = b.y; out_rgb[2] = b.z;
}
extern "C" void ref_srgb(float linear, float* to_srgb, float* back) {
	*to_srgb = convert_linear_to_srgb(linear);
	*back = convert_srgb_to_linear(*to_srgb);
}
