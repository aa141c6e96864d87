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

// ---- Arvo 2001: projected solid angle sampling (the baseline the paper improves on) ----------
// polygon_sampling_related_work.glsl:509-1048.  Azimuthal sectors as in the paper's own sampler,
// but the CDF is inverted by cubic interpolation plus Newton iterations in the azimuth.

struct edge_arvo {
	float cdf_factor;
	f2 length_coeffs, elevations;
};

template <int V>
struct psa_arvo {
	uint32_t vertex_count;
	float azimuths[V];
	float edge_cdf[V];       // edges[i] split into register arrays
	float edge_lx[V], edge_ly[V], edge_ex[V], edge_ey[V];
	edge_arvo inner_edge_0;
	float sector[V];
	float total;
};

template <int V>
VKR_DEV edge_arvo get_edge(const psa_arvo<V>& p, int i) {
	edge_arvo e;
	e.cdf_factor = opaque(p.edge_cdf[i]);
	e.length_coeffs = mk2(opaque(p.edge_lx[i]), p.edge_ly[i]);
	e.elevations = mk2(p.edge_ex[i], p.edge_ey[i]);
	return e;
}
template <int V>
VKR_DEV void set_edge(psa_arvo<V>& p, int i, const edge_arvo& e) {
	p.edge_cdf[i] = e.cdf_factor;
	p.edge_lx[i] = e.length_coeffs.x; p.edge_ly[i] = e.length_coeffs.y;
	p.edge_ex[i] = e.elevations.x; p.edge_ey[i] = e.elevations.y;
}
VKR_DEV edge_arvo select_edge(bool take_a, const edge_arvo& a, const edge_arvo& b) {
	edge_arvo e;
	e.cdf_factor = take_a ? a.cdf_factor : b.cdf_factor;
	e.length_coeffs = mk2(take_a ? a.length_coeffs.x : b.length_coeffs.x, take_a ? a.length_coeffs.y : b.length_coeffs.y);
	e.elevations = mk2(take_a ? a.elevations.x : b.elevations.x, take_a ? a.elevations.y : b.elevations.y);
	return e;
}

// prepare_edge_arvo, :559-578
VKR_DEV edge_arvo prepare_edge_arvo(f3 vertex_0, f3 vertex_1) {
	edge_arvo edge;
	f3 normal_a = normalize(cross(vertex_0, vertex_1));
	edge.cdf_factor = 0.5f * normal_a.z;
	f3 ccw_vertex = (edge.cdf_factor > 0.0f) ? vertex_0 : vertex_1;
	f2 normal_c = rot90(normalize(mk2(ccw_vertex.x, ccw_vertex.y)));
	float cos_beta = -dot(mk2(normal_a.x, normal_a.y), normal_c);
	float sin_beta_sq = fmaf(-cos_beta, cos_beta, 1.0f);
	float csc_beta = rsqrt(gmax(0.0f, sin_beta_sq));
	float csc_c = rsqrt(gmax(0.0f, fmaf(-ccw_vertex.z, ccw_vertex.z, 1.0f)));
	float scale = csc_beta * csc_c;
	edge.length_coeffs = mk2(sin_beta_sq * scale, (dot(mk2(normal_a.x, normal_a.y), rot90(normal_c)) * cos_beta) * scale);
	float elevation_1 = cross(ccw_vertex, normal_a).z;
	edge.elevations = mk2(ccw_vertex.z, (edge.cdf_factor > 0.0f) ? -elevation_1 : elevation_1);
	return edge;
}

// get_edge_projected_solid_angle_in_sector_arvo, :599-609
VKR_DEV float edge_psa_in_sector_arvo(const edge_arvo& edge, float relative_azimuth_0, float relative_azimuth_1) {
	float s0, c0, s1, c1;
	sincos_poly(relative_azimuth_0, s0, c0);
	sincos_poly(relative_azimuth_1, s1, c1);
	f2 point_0 = mk2(dot(edge.length_coeffs, mk2(c0, s0)), s0);
	f2 point_1 = mk2(dot(edge.length_coeffs, mk2(c1, s1)), s1);
	f2 rotated = mk2(point_0.x * point_1.x + point_0.y * point_1.y, point_0.x * point_1.y - point_0.y * point_1.x);
	float length = positive_atan<false>(divide(fabsf(rotated.y), rotated.x));
	return edge.cdf_factor * length;
}

// get_edge_projected_solid_angle_in_sector_derivative_arvo, :618-640: (value, d / d azimuth_1)
VKR_DEV f2 edge_psa_in_sector_derivative_arvo(const edge_arvo& edge, float relative_azimuth_0, float relative_azimuth_1) {
	float s0, c0, s1, c1;
	sincos_poly(relative_azimuth_0, s0, c0);
	sincos_poly(relative_azimuth_1, s1, c1);
	f2 point_0 = mk2(dot(edge.length_coeffs, mk2(c0, s0)), s0);
	f2 dir_1 = mk2(c1, s1);
	f2 point_1 = mk2(dot(edge.length_coeffs, dir_1), dir_1.y);
	f2 rotated = mk2(point_0.x * point_1.x + point_0.y * point_1.y, point_0.x * point_1.y - point_0.y * point_1.x);
	float quotient = divide(fabsf(rotated.y), rotated.x);
	float length = positive_atan<false>(quotient);
	f2 dir_1_deriv = rot90(dir_1);
	f2 point_1_deriv = mk2(dot(edge.length_coeffs, dir_1_deriv), dir_1_deriv.y);
	f2 rotated_deriv = mk2(point_0.x * point_1_deriv.x + point_0.y * point_1_deriv.y, point_0.x * point_1_deriv.y - point_0.y * point_1_deriv.x);
	float quotient_derivative = divide(rotated_deriv.y * rotated.x - rotated.y * rotated_deriv.x, rotated.x * rotated.x);
	quotient_derivative = (rotated.y < 0.0f) ? (-quotient_derivative) : quotient_derivative;
	float length_deriv = divide(quotient_derivative, fmaf(quotient, quotient, 1.0f));
	return mk2(edge.cdf_factor * length, edge.cdf_factor * length_deriv);
}

// get_edge_elevation_arvo, :648-653
VKR_DEV float edge_elevation_arvo(const edge_arvo& edge, float relative_azimuth) {
	float sn, cs;
	sincos_poly(relative_azimuth, sn, cs);
	f2 point = normalize(mk2(dot(edge.length_coeffs, mk2(cs, sn)), sn));
	return dot(point, edge.elevations);
}

// compare_and_swap_arvo, :661-669, with compile-time indices
template <int V, int L, int R>
VKR_DEV void compare_and_swap_arvo(psa_arvo<V>& p) {
	if constexpr (L < V && R < V) {
		float la = p.azimuths[L], ra = p.azimuths[R];
		bool swap = (la - ra) > 0.0f;
		p.azimuths[L] = swap ? ra : la;
		p.azimuths[R] = swap ? la : ra;
		edge_arvo le = get_edge<V>(p, L), re = get_edge<V>(p, R);
		set_edge<V>(p, L, select_edge(swap, re, le));
		set_edge<V>(p, R, select_edge(swap, le, re));
	}
}

// sort_convex_polygon_vertices_arvo, :674-739
template <int V>
VKR_DEV void sort_vertices_arvo(psa_arvo<V>& p) {
	switch (p.vertex_count) {
	case 3: compare_and_swap_arvo<V, 1, 2>(p); break;
	case 4: compare_and_swap_arvo<V, 1, 3>(p); break;
	case 5:
		compare_and_swap_arvo<V, 2, 4>(p); compare_and_swap_arvo<V, 1, 3>(p); compare_and_swap_arvo<V, 1, 2>(p);
		compare_and_swap_arvo<V, 0, 3>(p); compare_and_swap_arvo<V, 3, 4>(p);
		break;
	case 6:
		compare_and_swap_arvo<V, 3, 5>(p); compare_and_swap_arvo<V, 2, 4>(p); compare_and_swap_arvo<V, 1, 5>(p);
		compare_and_swap_arvo<V, 0, 4>(p); compare_and_swap_arvo<V, 4, 5>(p); compare_and_swap_arvo<V, 1, 3>(p);
		break;
	case 7:
		compare_and_swap_arvo<V, 2, 5>(p); compare_and_swap_arvo<V, 1, 6>(p); compare_and_swap_arvo<V, 5, 6>(p);
		compare_and_swap_arvo<V, 3, 4>(p); compare_and_swap_arvo<V, 0, 4>(p); compare_and_swap_arvo<V, 4, 6>(p);
		compare_and_swap_arvo<V, 1, 3>(p); compare_and_swap_arvo<V, 3, 5>(p); compare_and_swap_arvo<V, 4, 5>(p);
		break;
	case 8:
		compare_and_swap_arvo<V, 2, 6>(p); compare_and_swap_arvo<V, 3, 7>(p); compare_and_swap_arvo<V, 1, 5>(p);
		compare_and_swap_arvo<V, 0, 4>(p); compare_and_swap_arvo<V, 4, 6>(p); compare_and_swap_arvo<V, 5, 7>(p);
		compare_and_swap_arvo<V, 6, 7>(p); compare_and_swap_arvo<V, 4, 5>(p); compare_and_swap_arvo<V, 1, 3>(p);
		break;
	default: break;
	}
	compare_and_swap_arvo<V, 0, 2>(p);
	if (p.vertex_count >= 4) compare_and_swap_arvo<V, 2, 3>(p);
	compare_and_swap_arvo<V, 0, 1>(p);
}

// prepare_projected_solid_angle_polygon_sampling_arvo, :744-812
template <int V>
VKR_DEV void prepare_psa_arvo(psa_arvo<V>& p, uint32_t vertex_count, const f3 (&in_vertices)[V]) {
	f3 vertices[V];
#pragma unroll
	for (int i = 0; i < V; ++i) vertices[i] = normalize(in_vertices[i]);
	p.vertex_count = vertex_count;
	p.inner_edge_0.cdf_factor = 1.0f;
	p.inner_edge_0.length_coeffs = p.inner_edge_0.elevations = mk2(0.0f, 0.0f);
	p.azimuths[0] = arctan2(vertices[0].y, vertices[0].x);
	edge_arvo first_edge = prepare_edge_arvo(vertices[0], vertices[1]);
	set_edge<V>(p, 0, first_edge);
	edge_arvo previous_edge = first_edge;
	bool done = false;
#pragma unroll
	for (int i = 1; i < V; ++i) {
		float azimuth = arctan2(vertices[i].y, vertices[i].x);
		azimuth -= (azimuth > p.azimuths[0] + kPi) ? (2.0f * kPi) : 0.0f;
		azimuth += (azimuth < p.azimuths[0] - kPi) ? (2.0f * kPi) : 0.0f;
		// (the reference also writes the azimuth of the slot at which it stops; later slots
		// stay untouched there and are never read)
		p.azimuths[i] = done ? 0.0f : azimuth;
		set_edge<V>(p, i, first_edge);
		p.sector[i] = 0.0f;
		done = done || (i > 2 && (uint32_t) i == vertex_count);
		if (!done) {
			edge_arvo edge = prepare_edge_arvo(vertices[i], vertices[(i + 1) % V]);
			set_edge<V>(p, i, select_edge(edge.cdf_factor >= 0.0f, edge, previous_edge));
			p.inner_edge_0 = select_edge(previous_edge.cdf_factor < 0.0f && edge.cdf_factor >= 0.0f, previous_edge, p.inner_edge_0);
			previous_edge = edge;
		}
	}
	p.sector[0] = 0.0f;
	set_edge<V>(p, 0, select_edge(first_edge.cdf_factor >= 0.0f, first_edge, previous_edge));
	p.inner_edge_0 = select_edge(previous_edge.cdf_factor < 0.0f && first_edge.cdf_factor >= 0.0f, previous_edge, p.inner_edge_0);
	p.total = 0.0f;
	if (p.inner_edge_0.cdf_factor > 0.0f) {
		done = false;
#pragma unroll
		for (int i = 0; i < V; ++i) {
			done = done || (i > 2 && (uint32_t) i == vertex_count);
			if (!done) {
				p.sector[i] = edge_psa_in_sector_arvo(get_edge<V>(p, i), 0.0f, p.azimuths[(i + 1) % V] - p.azimuths[i]);
				p.total += p.sector[i];
			}
		}
	}
	else {
		sort_vertices_arvo<V>(p);
		edge_arvo inner_edge = p.inner_edge_0;
		float inner_azimuth = p.azimuths[0];
		edge_arvo outer_edge = get_edge<V>(p, 0);
		float outer_azimuth = p.azimuths[0];
		done = false;
#pragma unroll
		for (int i = 0; i < V - 1; ++i) {
			done = done || (i > 1 && (uint32_t) (i + 1) == vertex_count);
			if (!done) {
				edge_arvo vertex_edge = get_edge<V>(p, i);
				float vertex_azimuth = p.azimuths[i];
				if (i > 0) {
					bool outer = vertex_edge.cdf_factor >= 0.0f;
					inner_edge = select_edge(outer, inner_edge, vertex_edge);
					inner_azimuth = outer ? inner_azimuth : vertex_azimuth;
					outer_edge = select_edge(outer, vertex_edge, outer_edge);
					outer_azimuth = outer ? vertex_azimuth : outer_azimuth;
				}
				float sector = edge_psa_in_sector_arvo(outer_edge, p.azimuths[i] - outer_azimuth, p.azimuths[i + 1] - outer_azimuth);
				sector += edge_psa_in_sector_arvo(inner_edge, p.azimuths[i] - inner_azimuth, p.azimuths[i + 1] - inner_azimuth);
				p.sector[i] = sector;
				p.total += sector;
			}
		}
	}
}

// evaluate_cubic_interpolation_polynomial, :822-830
VKR_DEV float cubic_interpolation(float sample_x, const float (&x)[4], const float (&y)[4]) {
	float y01 = divide(y[0] - y[1], x[0] - x[1]);
	float y12 = divide(y[1] - y[2], x[1] - x[2]);
	float y23 = divide(y[2] - y[3], x[2] - x[3]);
	float y012 = divide(y01 - y12, x[0] - x[2]);
	float y123 = divide(y12 - y23, x[1] - x[3]);
	float y0123 = divide(y012 - y123, x[0] - x[3]);
	return fmaf(sample_x - x[0], fmaf(sample_x - x[1], fmaf(sample_x - x[2], y0123, y012), y01), y[0]);
}

// sample_sector_within_edge (WITH_INNER false), :838-866, and sample_sector_between_edges, :890-925
template <bool WITH_INNER>
VKR_DEV f3 sample_sector_arvo(f2 random_numbers, float target, const edge_arvo& inner_edge, float inner_azimuth, const edge_arvo& outer_edge, float outer_azimuth, float azimuth_0, float azimuth_1, uint32_t iteration_count) {
	float azimuths[4] = {azimuth_0, mix_fma(azimuth_0, azimuth_1, 1.0f / 3.0f), mix_fma(azimuth_0, azimuth_1, 2.0f / 3.0f), azimuth_1};
	float psas[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		psas[i] = edge_psa_in_sector_arvo(outer_edge, azimuth_0 - outer_azimuth, azimuths[i] - outer_azimuth);
		if constexpr (WITH_INNER) psas[i] += edge_psa_in_sector_arvo(inner_edge, azimuth_0 - inner_azimuth, azimuths[i] - inner_azimuth);
	}
	float sampled_azimuth = cubic_interpolation(target, psas, azimuths);
	for (uint32_t i = 0; i != iteration_count; ++i) {
		f2 outer_psa = edge_psa_in_sector_derivative_arvo(outer_edge, azimuth_0 - outer_azimuth, sampled_azimuth - outer_azimuth);
		float error, derivative;
		if constexpr (WITH_INNER) {
			f2 inner_psa = edge_psa_in_sector_derivative_arvo(inner_edge, azimuth_0 - inner_azimuth, sampled_azimuth - inner_azimuth);
			error = inner_psa.x + outer_psa.x - target;
			derivative = inner_psa.y + outer_psa.y;
		}
		else {
			error = outer_psa.x - target;
			derivative = outer_psa.y;
		}
		sampled_azimuth -= divide(error, derivative);
		sampled_azimuth = gclamp(sampled_azimuth, azimuth_0, azimuth_1);
	}
	float sn, cs;
	sincos_poly(sampled_azimuth, sn, cs);
	float outer_z = edge_elevation_arvo(outer_edge, sampled_azimuth - outer_azimuth);
	float z;
	if constexpr (WITH_INNER) {
		float inner_z = edge_elevation_arvo(inner_edge, sampled_azimuth - inner_azimuth);
		z = square_root(mix_fma(inner_z * inner_z, outer_z * outer_z, random_numbers.y));
	}
	else
		z = square_root(mix_fma(1.0f, outer_z * outer_z, random_numbers.y));
	float scale = square_root(fmaf(-z, z, 1.0f));
	return mk3(cs * scale, sn * scale, z);
}

// the sector search of the decentral case, shared by sampling (:965-988) and error (:1015-1035)
template <int V>
VKR_DEV void find_sector_arvo(const psa_arvo<V>& p, float& target, float& sector_psa, edge_arvo& inner_edge, float& inner_azimuth, edge_arvo& outer_edge, float& outer_azimuth, float& azimuth_0, float& azimuth_1) {
	inner_edge = p.inner_edge_0;
	inner_azimuth = p.azimuths[0];
	bool done = false;
#pragma unroll
	for (int i = 0; i < V - 1; ++i) {
		done = done || (i > 1 && (uint32_t) (i + 1) == p.vertex_count) || (i > 0 && target < 0.0f);
		if (!done) {
			sector_psa = p.sector[i];
			target -= sector_psa;
			edge_arvo vertex_edge = get_edge<V>(p, i);
			float vertex_azimuth = opaque(p.azimuths[i]);
			if (i == 0) {
				outer_edge = vertex_edge;
				outer_azimuth = vertex_azimuth;
			}
			else {
				bool outer = vertex_edge.cdf_factor >= 0.0f;
				inner_edge = select_edge(outer, inner_edge, vertex_edge);
				inner_azimuth = outer ? inner_azimuth : vertex_azimuth;
				outer_edge = select_edge(outer, vertex_edge, outer_edge);
				outer_azimuth = outer ? vertex_azimuth : outer_azimuth;
			}
			azimuth_0 = vertex_azimuth;
			azimuth_1 = opaque(p.azimuths[i + 1]);
		}
	}
	target += sector_psa;
}

// sample_projected_solid_angle_polygon_arvo, :934-991
template <int V>
VKR_DEV f3 sample_psa_arvo(const psa_arvo<V>& p, f2 random_numbers, uint32_t iteration_count) {
	float target = random_numbers.x * p.total;
	float sector_psa = 0.0f;
	edge_arvo outer_edge = p.inner_edge_0, inner_edge = p.inner_edge_0;
	float outer_azimuth = 0.0f, inner_azimuth = 0.0f, azimuth_0 = 0.0f, azimuth_1 = 0.0f;
	if (p.inner_edge_0.cdf_factor > 0.0f) {
		bool done = false;
#pragma unroll
		for (int i = 0; i < V; ++i) {
			done = done || (i > 2 && (uint32_t) i == p.vertex_count) || (i > 0 && target < 0.0f);
			if (!done) {
				sector_psa = p.sector[i];
				target -= sector_psa;
				outer_edge = get_edge<V>(p, i);
				outer_azimuth = opaque(p.azimuths[i]);
				azimuth_1 = opaque(p.azimuths[(i + 1) % V]);
			}
		}
		azimuth_1 = (azimuth_1 < outer_azimuth) ? (azimuth_1 + 2.0f * kPi) : azimuth_1;
		target += sector_psa;
		return sample_sector_arvo<false>(random_numbers, target, inner_edge, 0.0f, outer_edge, outer_azimuth, outer_azimuth, azimuth_1, iteration_count);
	}
	find_sector_arvo<V>(p, target, sector_psa, inner_edge, inner_azimuth, outer_edge, outer_azimuth, azimuth_0, azimuth_1);
	return sample_sector_arvo<true>(random_numbers, target, inner_edge, inner_azimuth, outer_edge, outer_azimuth, azimuth_0, azimuth_1, iteration_count);
}

// compute_projected_solid_angle_polygon_sampling_error_arvo, :998-1047: (backward, backward scaled)
template <int V>
VKR_DEV f2 psa_sampling_error_arvo(const psa_arvo<V>& p, f2 random_numbers, f3 sampled_dir) {
	float target = random_numbers.x * p.total;
	if (p.inner_edge_0.cdf_factor > 0.0f) return mk2(0.0f, 0.0f);
	edge_arvo inner_edge = p.inner_edge_0, outer_edge = p.inner_edge_0;
	float inner_azimuth = 0.0f, outer_azimuth = 0.0f, sector_psa = 0.0f, azimuth_0 = 0.0f, azimuth_1 = 0.0f;
	find_sector_arvo<V>(p, target, sector_psa, inner_edge, inner_azimuth, outer_edge, outer_azimuth, azimuth_0, azimuth_1);
	float sampled_azimuth = arctan2(sampled_dir.y, sampled_dir.x);
	float outer_psa = edge_psa_in_sector_derivative_arvo(outer_edge, azimuth_0 - outer_azimuth, sampled_azimuth - outer_azimuth).x;
	float inner_psa = edge_psa_in_sector_derivative_arvo(inner_edge, azimuth_0 - inner_azimuth, sampled_azimuth - inner_azimuth).x;
	float sampled_psa = outer_psa + inner_psa;
	return mk2(divide(target - sampled_psa, p.total), target - sampled_psa);
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
