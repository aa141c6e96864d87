// Samplers of the reference's related-work comparison set that the experiment table uses
// next to the paper's own: Urena's spherical rectangles, Arvo's spherical triangles and Hart's
// bilinear cosine warp (reference src/shaders/polygon_sampling_related_work.glsl:97-386).
// Same operations in the same order as oracle/oracle_shading.c; loops are written without
// early exits for the same reason as in polygon_sampling.h (no scratch memory).
#pragma once
#include "polygon_sampling.h"

namespace vkr {

// ---- Urena, Fajardo, King 2013: area-preserving parametrisation of spherical rectangles ----

struct urena_rectangle {
	f3 x, y, z;
	float z0, z0sq, x0, y0, y0sq, x1, y1, y1sq, b0, b1, b0sq, k, solid_angle;
};

// prepare_solid_angle_rectangle_sampling_urena, :121-164 (s: a corner, exl / eyl: edge lengths,
// r0..r2: orthonormal columns along the edges and the normal, o: shading point)
VKR_DEV urena_rectangle prepare_urena(f3 s, float exl, float eyl, f3 r0, f3 r1, f3 r2, f3 o) {
	urena_rectangle q;
	q.x = r0; q.y = r1; q.z = r2;
	f3 d = s - o;
	q.z0 = dot(d, q.z);
	q.z = (q.z0 > 0.0f) ? mk3(-q.z.x, -q.z.y, -q.z.z) : q.z;
	q.z0 = -fabsf(q.z0);
	q.z0sq = q.z0 * q.z0;
	q.x0 = dot(d, q.x);
	q.y0 = dot(d, q.y);
	q.x1 = q.x0 + exl;
	q.y1 = q.y0 + eyl;
	q.y0sq = q.y0 * q.y0;
	q.y1sq = q.y1 * q.y1;
	f3 v00 = mk3(q.x0, q.y0, q.z0), v01 = mk3(q.x0, q.y1, q.z0), v10 = mk3(q.x1, q.y0, q.z0), v11 = mk3(q.x1, q.y1, q.z0);
	f3 n0 = normalize(cross(v00, v10)), n1 = normalize(cross(v10, v11)), n2 = normalize(cross(v11, v01)), n3 = normalize(cross(v01, v00));
	float g0 = arccos(-dot(n0, n1)), g1 = arccos(-dot(n1, n2)), g2 = arccos(-dot(n2, n3)), g3 = arccos(-dot(n3, n0));
	q.b0 = n0.z;
	q.b1 = n2.z;
	q.b0sq = q.b0 * q.b0;
	q.k = 2.0f * kPi - g2 - g3;
	q.solid_angle = g0 + g1 - q.k;
	return q;
}

// sample_solid_angle_rectangle_urena, :171-193
VKR_DEV f3 sample_urena(const urena_rectangle& q, f2 random_numbers) {
	float u = random_numbers.x, v = random_numbers.y;
	float au = fmaf(u, q.solid_angle, q.k);
	float sin_au, cos_au;
	sincos_poly(au, sin_au, cos_au);
	float fu = divide(fmaf(cos_au, q.b0, -q.b1), sin_au);
	float cu = rsqrt(fmaf(fu, fu, q.b0sq));
	cu = (fu > 0.0f) ? cu : -cu;
	cu = gclamp(cu, -1.0f, 1.0f);
	float xu = -(cu * q.z0) * rsqrt(fmaf(-cu, cu, 1.0f));
	xu = gclamp(xu, q.x0, q.x1);
	float d = square_root(xu * xu + q.z0sq);
	float h0 = q.y0 * rsqrt(fmaf(d, d, q.y0sq));
	float h1 = q.y1 * rsqrt(fmaf(d, d, q.y1sq));
	float hv = h0 + v * (h1 - h0);
	float mhv2_1 = fmaf(-hv, hv, 1.0f);
	float yv = (mhv2_1 >= 0.0f) ? ((hv * d) * rsqrt(mhv2_1)) : q.y1;
	return normalize((q.x * xu + q.y * yv) + q.z * q.z0);
}

// ---- Arvo 1995: stratified sampling of spherical triangles, over a triangle fan ------------

template <int V>
struct arvo_polygon {
	uint32_t vertex_count;
	f3 dirs[V];
	float fan[V > 2 ? V - 2 : 1];
	f2 opposite[V > 2 ? V - 2 : 1];
	float solid_angle;
};

// prepare_solid_angle_polygon_sampling_arvo, :219-254
template <int V>
VKR_DEV void prepare_arvo(arvo_polygon<V>& p, uint32_t vertex_count, const f3 (&verts)[V], f3 shading_position) {
	p.vertex_count = vertex_count;
#pragma unroll
	for (int i = 0; i < V; ++i) p.dirs[i] = normalize(verts[i] - shading_position);
	float solid_angle = 0.0f;
#pragma unroll
	for (int i = 0; i < V - 2; ++i) {
		p.fan[i] = 0.0f;
		p.opposite[i] = mk2(0.0f, 0.0f);
		if (i >= 1 && (uint32_t) (i + 2) >= vertex_count) continue;
		f3 n0 = normalize(cross(p.dirs[i + 1] - p.dirs[0], p.dirs[0]));
		f3 n1 = normalize(cross(p.dirs[i + 2] - p.dirs[i + 1], p.dirs[i + 1]));
		float ox = -dot(n0, n1);
		p.opposite[i] = mk2(ox, square_root(gmax(0.0f, fmaf(-ox, ox, 1.0f))));
		float d01 = dot(p.dirs[0], p.dirs[i + 1]), d02 = dot(p.dirs[0], p.dirs[i + 2]), d12 = dot(p.dirs[i + 1], p.dirs[i + 2]);
		f3 c0 = p.dirs[0], c1 = p.dirs[i + 1], c2 = p.dirs[i + 2];
		float volume = c0.x * (c1.y * c2.z - c2.y * c1.z) - c1.x * (c0.y * c2.z - c2.y * c0.z) + c2.x * (c0.y * c1.z - c1.y * c0.z);
		float tangent = divide(fabsf(volume), ((1.0f + d01) + d02) + d12);
		solid_angle += 2.0f * positive_atan<false>(tangent);
		p.fan[i] = solid_angle;
	}
	p.solid_angle = solid_angle;
}

// sample_solid_angle_polygon_arvo, :259-294
template <int V>
VKR_DEV f3 sample_arvo(const arvo_polygon<V>& p, f2 random_numbers) {
	float target = p.solid_angle * random_numbers.x;
	float sub = target;
	f2 opposite = p.opposite[0];
	f3 t0 = p.dirs[1], t1 = p.dirs[0], t2 = p.dirs[2];
	bool done = false;
#pragma unroll
	for (int i = 0; i < V - 3; ++i) {
		done = done || (uint32_t) (i + 3) >= p.vertex_count || p.fan[i] >= target;
		if (!done) {
			sub = target - p.fan[i];
			t0 = p.dirs[i + 2];
			t2 = p.dirs[i + 3];
			opposite = p.opposite[i + 1];
		}
	}
	float sn, cs;
	sincos_poly(sub, sn, cs);
	float pp = sn * opposite.x - cs * opposite.y;
	float q = sn * opposite.y + cs * opposite.x;
	float u = q - opposite.x;
	float v = pp + opposite.y * dot(t0, t1);
	float s = divide((v * q - u * pp) * opposite.x - v, (v * pp + u * q) * opposite.y);
	f3 tangent_2_0 = normalize(t2 - t0 * dot(t0, t2));
	f3 vertex_2 = t0 * s + tangent_2_0 * square_root(gclamp(fmaf(-s, s, 1.0f), 0.0f, 1.0f));
	float z = 1.0f - random_numbers.y * (1.0f - dot(vertex_2, t1));
	f3 tangent_2_1 = normalize(vertex_2 - t1 * dot(t1, vertex_2));
	return t1 * z + tangent_2_1 * square_root(gclamp(fmaf(-z, z, 1.0f), 0.0f, 1.0f));
}

// ---- Hart et al. 2020: bilinear warp of the primary sample space towards the cosine ----------

template <int V>
struct hart_bilinear {
	sa_polygon<V> polygon;
	float density_0;
	f2 density_1;
};

// prepare_bilinear_cosine_warp_polygon_sampling_hart, :316-337
template <int V>
VKR_DEV void prepare_hart_bilinear(hart_bilinear<V>& h, uint32_t vertex_count, const f3 (&verts)[V]) {
	prepare_sa<V>(h.polygon, vertex_count, verts, mk3(0.0f, 0.0f, 0.0f));
	h.density_0 = gmax(0.0f, h.polygon.dirs[0].z);
	float d1x = gmax(0.0f, h.polygon.dirs[1].z);
	float d1y = h.polygon.dirs[2].z;
#pragma unroll
	for (int i = 3; i < V; ++i) d1y = ((uint32_t) i < vertex_count) ? h.polygon.dirs[i].z : d1y;
	d1y = gmax(0.0f, d1y);
	float density_sum = (2.0f * h.density_0 + d1x) + d1y;
	float normalization = divide(4.0f, h.polygon.solid_angle * density_sum);
	h.density_0 *= normalization;
	h.density_1 = mk2(d1x * normalization, d1y * normalization);
	float inv_solid_angle = rcp(h.polygon.solid_angle);
	if (density_sum <= 0.0f) {
		h.density_0 = inv_solid_angle;
		h.density_1 = mk2(inv_solid_angle, inv_solid_angle);
	}
}

// linear_warp, :349-353
VKR_DEV float linear_warp(float random_number, float density_0, float density_1) {
	float lerped_density_sq = mix_fma(density_0 * density_0, density_1 * density_1, random_number);
	float divisor = density_0 + square_root(lerped_density_sq);
	return divide(random_number * (density_0 + density_1), divisor);
}

// sample_bilinear_cosine_warp_polygon_hart, :373-380
template <int V>
VKR_DEV f3 sample_hart_bilinear(float& out_density, const hart_bilinear<V>& h, f2 u) {
	u.y = linear_warp(u.y, 2.0f * h.density_0, h.density_1.x * 1.0f + h.density_1.y * 1.0f);
	float density_0 = mix_fma(h.density_0, h.density_1.x, u.y);
	float density_1 = mix_fma(h.density_0, h.density_1.y, u.y);
	u.x = linear_warp(u.x, density_0, density_1);
	out_density = mix_fma(density_0, density_1, u.x);
	return sample_sa<V>(h.polygon, u);
}

// ---- Hart et al. 2020: biquadratic warp (needs the roots of a cubic) -------------------------

// solve_cubic, cubic_solver.glsl:29-76: c0 + c1 x + c2 x^2 + c3 x^3; true with three roots, false
// with one root in r0
VKR_DEV bool solve_cubic(float& r0, float& r1, float& r2, float k0, float k1, float k2, float k3) {
	float c0 = divide(k0, k3), c1 = divide(k1, k3), c2 = divide(k2, k3);
	c1 = divide(c1, 3.0f);
	c2 = divide(c2, 3.0f);
	float d0 = fmaf(-c2, c2, c1), d1 = fmaf(-c1, c2, c0), d2 = c2 * c0 - c1 * c1;
	float discriminant = 4.0f * d0 * d2 - d1 * d1;
	float sqrt_abs_discriminant = square_root(fabsf(discriminant));
	float depressed_0 = fmaf(-2.0f * c2, d0, d1), depressed_1 = d0;
	if (discriminant >= 0.0f) {
		float theta = arctan2(sqrt_abs_discriminant, -depressed_0) * (1.0f / 3.0f);
		float sn, cs;
		sincos_poly(theta, sn, cs);
		const float sqrt_three_quarters = 0.866025388f;
		float q0 = cs, q1 = fmaf(-sqrt_three_quarters, sn, -0.5f * cs), q2 = fmaf(sqrt_three_quarters, sn, -0.5f * cs);
		float scale = 2.0f * square_root(-depressed_1);
		r0 = fmaf(scale, q0, -c2);
		r1 = fmaf(scale, q1, -c2);
		r2 = fmaf(scale, q2, -c2);
		return true;
	}
	float signed_sqrt_discriminant = (depressed_0 < 0.0f) ? sqrt_abs_discriminant : -sqrt_abs_discriminant;
	float quadratic_root = 0.5f * (signed_sqrt_discriminant - depressed_0);
	float cube_root_0 = cube_root_positive(fabsf(quadratic_root));
	cube_root_0 = (quadratic_root < 0.0f) ? -cube_root_0 : cube_root_0;
	float cube_root_1 = divide(-depressed_1, cube_root_0);
	r0 = (cube_root_0 + cube_root_1) - c2;
	r1 = r2 = 0.0f;
	return false;
}

template <int V>
struct hart_biquadratic {
	sa_polygon<V> polygon;
	float density_0;
	f3 density_1, density_2;
};

// prepare_biquadratic_cosine_warp_polygon_sampling_hart, :405-446
template <int V>
VKR_DEV void prepare_hart_biquadratic(hart_biquadratic<V>& h, uint32_t vertex_count, const f3 (&verts)[V]) {
	prepare_sa<V>(h.polygon, vertex_count, verts, mk3(0.0f, 0.0f, 0.0f));
	f3 last_vertex = h.polygon.dirs[2];
#pragma unroll
	for (int i = 3; i < V; ++i) last_vertex = ((uint32_t) i < vertex_count) ? h.polygon.dirs[i] : last_vertex;
	f3 vertex_0 = h.polygon.dirs[0];
	h.density_0 = gmax(0.0f, vertex_0.z);
	f3 sample_2_1 = sample_sa<V>(h.polygon, mk2(0.5f, 1.0f));
	h.density_2 = mk3(gmax(0.0f, h.polygon.dirs[1].z), gmax(0.0f, sample_2_1.z), gmax(0.0f, last_vertex.z));
	f3 far_vertices[3] = {vertex_0, sample_2_1, last_vertex};
	float density_1[3];
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		float s2 = dot(vertex_0, far_vertices[i]);
		float s = fmaf(0.5f, s2, 0.5f);
		float t = square_root(gmax(0.0f, fmaf(-s, s, 1.0f)));
		float t_axis_z = fmaf(-s2, vertex_0.z, far_vertices[i].z);
		float normalization_t_axis = rsqrt(2.0f * fmaf(-s2, s2, 1.0f));
		float sample_1_i_z = s * vertex_0.z + (t * normalization_t_axis) * t_axis_z;
		density_1[i] = gmax(0.0f, sample_1_i_z);
	}
	h.density_1 = mk3(density_1[0], density_1[1], density_1[2]);
	float density_sum = (3.0f * h.density_0 + ((h.density_1.x + h.density_1.y) + h.density_1.z)) + ((h.density_2.x + h.density_2.y) + h.density_2.z);
	float normalization = divide(9.0f, h.polygon.solid_angle * density_sum);
	h.density_0 *= normalization;
	h.density_1 = h.density_1 * normalization;
	h.density_2 = h.density_2 * normalization;
	float inv_solid_angle = rcp(h.polygon.solid_angle);
	if (density_sum <= 0.0f) {
		h.density_0 = inv_solid_angle;
		h.density_1 = h.density_2 = mk3(inv_solid_angle, inv_solid_angle, inv_solid_angle);
	}
}

// quadratic_warp, :457-474
VKR_DEV float quadratic_warp(float random_number, float density_0, float density_1, float density_2) {
	float q0 = density_0, q1 = 2.0f * (density_1 - density_0), q2 = density_0 - 2.0f * density_1 + density_2;
	float k1 = q0, k2 = 0.5f * q1, k3 = (1.0f / 3.0f) * q2;
	random_number *= (k1 + k2) + k3;
	float r0, r1, r2;
	if (solve_cubic(r0, r1, r2, -random_number, k1, k2, k3)) {
		float result = r0;
		result = (r1 >= 0.0f && r1 <= 1.0f) ? r1 : result;
		result = (r2 >= 0.0f && r2 <= 1.0f) ? r2 : result;
		return result;
	}
	return r0;
}

// quadratic_bezier, :484-488
VKR_DEV float quadratic_bezier(float b00, float b01, float b02, float location) {
	return mix_fma(mix_fma(b00, b01, location), mix_fma(b01, b02, location), location);
}

// sample_biquadratic_cosine_warp_polygon_hart, :493-503
template <int V>
VKR_DEV f3 sample_hart_biquadratic(float& out_density, const hart_biquadratic<V>& h, f2 u) {
	u.y = quadratic_warp(u.y, 3.0f * h.density_0, (h.density_1.x + h.density_1.y) + h.density_1.z, (h.density_2.x + h.density_2.y) + h.density_2.z);
	float density_0 = quadratic_bezier(h.density_0, h.density_1.x, h.density_2.x, u.y);
	float density_1 = quadratic_bezier(h.density_0, h.density_1.y, h.density_2.y, u.y);
	float density_2 = quadratic_bezier(h.density_0, h.density_1.z, h.density_2.z, u.y);
	u.x = quadratic_warp(u.x, density_0, density_1, density_2);
	out_density = quadratic_bezier(density_0, density_1, density_2, u.x);
	return sample_sa<V>(h.polygon, u);
}

}  // namespace vkr
