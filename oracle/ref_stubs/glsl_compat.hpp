// TEST INFRASTRUCTURE - not part of the product.
//
// A small GLSL 4.60 look-alike for C++17 so that the reference's shader sources
// (/root/reference/src/shaders/*.glsl) compile as C++ and can be executed on the
// CPU.  build_ref_shaders.py includes the lightly pre-processed shader text inside
// namespace glsl below; no shader code is stored in this repository.
//
// Semantics follow the GLSL specification: column-major matrices, m[i] is column i,
// constructors fill column by column, component-wise vector arithmetic.  Where the
// specification leaves precision or order open (dot products, matrix products,
// normalize, the transcendental functions, texture filtering) the choices are the
// ones documented in oracle/oracle_math.h: left-to-right accumulation, libm
// functions, exact fp32 bilinear weights.  Must be compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#undef M_PI

namespace glsl {

typedef uint32_t uint;

// ---- vectors -------------------------------------------------------------------
// Swizzles are proxy members of a union that alias the component array.

template <class V, class T, int A, int B> struct swz2 {
	T d[4];
	operator V() const { return V(d[A], d[B]); }
	swz2& operator=(const V& v) { T a = v.x, b = v.y; d[A] = a; d[B] = b; return *this; }
	swz2& operator=(const swz2& o) { return *this = V(o); }
	swz2& operator*=(T s) { d[A] *= s; d[B] *= s; return *this; }
	swz2& operator/=(T s) { d[A] /= s; d[B] /= s; return *this; }
	swz2& operator*=(const V& v) { d[A] *= v.x; d[B] *= v.y; return *this; }
	swz2& operator+=(const V& v) { d[A] += v.x; d[B] += v.y; return *this; }
};
template <class V, class T, int A, int B, int C> struct swz3 {
	T d[4];
	operator V() const { return V(d[A], d[B], d[C]); }
	swz3& operator=(const V& v) { T a = v.x, b = v.y, c = v.z; d[A] = a; d[B] = b; d[C] = c; return *this; }
	swz3& operator=(const swz3& o) { return *this = V(o); }
	swz3& operator*=(T s) { d[A] *= s; d[B] *= s; d[C] *= s; return *this; }
	swz3& operator/=(T s) { d[A] /= s; d[B] /= s; d[C] /= s; return *this; }
};
template <class V, class T, int A, int B, int C, int D> struct swz4 {
	T d[4];
	operator V() const { return V(d[A], d[B], d[C], d[D]); }
};

#define GLSL_VEC_TYPES(P, T) \
struct P##vec2; struct P##vec3; struct P##vec4; \
struct P##vec2 { \
	union { struct { T x, y; }; struct { T r, g; }; T d[2]; \
		swz2<P##vec2, T, 0, 1> xy; swz2<P##vec2, T, 1, 0> yx; swz2<P##vec2, T, 0, 1> rg; }; \
	P##vec2() : x(0), y(0) {} \
	explicit P##vec2(T s) : x(s), y(s) {} \
	P##vec2(T x_, T y_) : x(x_), y(y_) {} \
	P##vec2(const P##vec2& o) : x(o.x), y(o.y) {} \
	P##vec2& operator=(const P##vec2& o) { x = o.x; y = o.y; return *this; } \
	T& operator[](int i) { return d[i]; } \
	const T& operator[](int i) const { return d[i]; } \
}; \
struct P##vec3 { \
	union { struct { T x, y, z; }; struct { T r, g, b; }; T d[3]; \
		swz2<P##vec2, T, 0, 1> xy; swz2<P##vec2, T, 1, 2> yz; swz2<P##vec2, T, 1, 0> yx; swz2<P##vec2, T, 0, 1> rg; \
		swz3<P##vec3, T, 0, 1, 2> xyz; swz3<P##vec3, T, 0, 1, 2> rgb; }; \
	P##vec3() : x(0), y(0), z(0) {} \
	explicit P##vec3(T s) : x(s), y(s), z(s) {} \
	P##vec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {} \
	P##vec3(const P##vec3& o) : x(o.x), y(o.y), z(o.z) {} \
	P##vec3& operator=(const P##vec3& o) { x = o.x; y = o.y; z = o.z; return *this; } \
	P##vec3(const P##vec2& v, T z_) : x(v.x), y(v.y), z(z_) {} \
	T& operator[](int i) { return d[i]; } \
	const T& operator[](int i) const { return d[i]; } \
}; \
struct P##vec4 { \
	union { struct { T x, y, z, w; }; struct { T r, g, b, a; }; T d[4]; \
		swz2<P##vec2, T, 0, 1> xy; swz2<P##vec2, T, 1, 2> yz; swz2<P##vec2, T, 2, 3> zw; swz2<P##vec2, T, 0, 1> rg; swz2<P##vec2, T, 2, 3> ba; \
		swz3<P##vec3, T, 0, 1, 2> xyz; swz3<P##vec3, T, 1, 2, 3> yzw; swz3<P##vec3, T, 0, 1, 2> rgb; \
		swz4<P##vec4, T, 2, 3, 0, 1> zwxy; }; \
	P##vec4() : x(0), y(0), z(0), w(0) {} \
	explicit P##vec4(T s) : x(s), y(s), z(s), w(s) {} \
	P##vec4(T x_, T y_, T z_, T w_) : x(x_), y(y_), z(z_), w(w_) {} \
	P##vec4(const P##vec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {} \
	P##vec4& operator=(const P##vec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; } \
	P##vec4(const P##vec3& v, T w_) : x(v.x), y(v.y), z(v.z), w(w_) {} \
	T& operator[](int i) { return d[i]; } \
	const T& operator[](int i) const { return d[i]; } \
};

GLSL_VEC_TYPES(, float)
GLSL_VEC_TYPES(u, uint)
GLSL_VEC_TYPES(i, int)

// conversions between element types that the shaders use
inline ivec2 to_ivec2(const vec2& v) { return ivec2((int) v.x, (int) v.y); }
inline vec3 make_vec3(const ivec2& p, float z) { return vec3((float) p.x, (float) p.y, z); }
inline ivec3 make_ivec3(const uvec2& p, uint z) { return ivec3((int) p.x, (int) p.y, (int) z); }

#define GLSL_VEC_OPS(V, N) \
inline V operator+(const V& a, const V& b) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; } \
inline V operator-(const V& a, const V& b) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; } \
inline V operator*(const V& a, const V& b) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] * b.d[i]; return r; } \
inline V operator/(const V& a, const V& b) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] / b.d[i]; return r; } \
inline V operator*(const V& a, float s) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] * s; return r; } \
inline V operator*(float s, const V& a) { V r; for (int i = 0; i != N; ++i) r.d[i] = s * a.d[i]; return r; } \
inline V operator/(const V& a, float s) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] / s; return r; } \
inline V operator-(float s, const V& a) { V r; for (int i = 0; i != N; ++i) r.d[i] = s - a.d[i]; return r; } \
inline V operator+(const V& a, float s) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] + s; return r; } \
inline V operator-(const V& a) { V r; for (int i = 0; i != N; ++i) r.d[i] = -a.d[i]; return r; } \
inline V& operator+=(V& a, const V& b) { for (int i = 0; i != N; ++i) a.d[i] += b.d[i]; return a; } \
inline V& operator-=(V& a, const V& b) { for (int i = 0; i != N; ++i) a.d[i] -= b.d[i]; return a; } \
inline V& operator*=(V& a, const V& b) { for (int i = 0; i != N; ++i) a.d[i] *= b.d[i]; return a; } \
inline V& operator*=(V& a, float s) { for (int i = 0; i != N; ++i) a.d[i] *= s; return a; } \
inline V fma(const V& a, const V& b, const V& c) { V r; for (int i = 0; i != N; ++i) r.d[i] = std::fmaf(a.d[i], b.d[i], c.d[i]); return r; } \
inline V abs(const V& a) { V r; for (int i = 0; i != N; ++i) r.d[i] = std::fabs(a.d[i]); return r; } \
inline V max(const V& a, const V& b) { V r; for (int i = 0; i != N; ++i) r.d[i] = (a.d[i] < b.d[i]) ? b.d[i] : a.d[i]; return r; } \
inline V mix(const V& a, const V& b, float t) { V r; for (int i = 0; i != N; ++i) r.d[i] = a.d[i] * (1.0f - t) + b.d[i] * t; return r; }

GLSL_VEC_OPS(vec2, 2)
GLSL_VEC_OPS(vec3, 3)
GLSL_VEC_OPS(vec4, 4)

inline uvec2 operator+(const uvec2& a, const uvec2& b) { return uvec2(a.x + b.x, a.y + b.y); }
inline uvec2 operator&(const uvec2& a, const uvec2& b) { return uvec2(a.x & b.x, a.y & b.y); }
inline uvec2 operator>>(const uvec2& a, uint s) { return uvec2(a.x >> s, a.y >> s); }

// ---- scalar built-ins ------------------------------------------------------------
inline float abs(float x) { return std::fabs(x); }
inline float sqrt(float x) { return std::sqrt(x); }
inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
inline float fma(float a, float b, float c) { return std::fmaf(a, b, c); }
inline float max(float x, float y) { return (x < y) ? y : x; }
inline float min(float x, float y) { return (y < x) ? y : x; }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float atan(float y) { return std::atan(y); }
inline float atan(float y, float x) { return std::atan2(y, x); }
inline float acos(float x) { return std::acos(x); }
inline float sin(float x) { return std::sin(x); }
inline float cos(float x) { return std::cos(x); }
inline float pow(float x, float y) { return std::pow(x, y); }
inline float log2(float x) { return std::log2(x); }
inline bool isnan(float x) { return std::isnan(x); }
inline bool isinf(float x) { return std::isinf(x); }
inline uint floatBitsToUint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
inline float uintBitsToFloat(uint u) { float f; std::memcpy(&f, &u, 4); return f; }

inline float dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(const vec4& a, const vec4& b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
inline vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline vec2 normalize(const vec2& a) { return a * inversesqrt(dot(a, a)); }
inline vec3 normalize(const vec3& a) { return a * inversesqrt(dot(a, a)); }
inline float length(const vec3& a) { return sqrt(dot(a, a)); }

// ---- matrices (column major) ------------------------------------------------------
struct mat2 {
	vec2 c[2];
	mat2() {}
	mat2(const vec2& a, const vec2& b) { c[0] = a; c[1] = b; }
	vec2& operator[](int i) { return c[i]; }
	const vec2& operator[](int i) const { return c[i]; }
};
inline mat2 operator-(const mat2& a, const mat2& b) { return mat2(a[0] - b[0], a[1] - b[1]); }
inline mat2& operator-=(mat2& a, const mat2& b) { a[0] -= b[0]; a[1] -= b[1]; return a; }
inline mat2 outerProduct(const vec2& col, const vec2& row) { return mat2(col * row.x, col * row.y); }
inline float determinant(const mat2& m) { return m[0][0] * m[1][1] - m[0][1] * m[1][0]; }
inline mat2 transpose(const mat2& m) { return mat2(vec2(m[0][0], m[1][0]), vec2(m[0][1], m[1][1])); }

struct mat3 {
	vec3 c[3];
	mat3() {}
	mat3(const vec3& a, const vec3& b, const vec3& d) { c[0] = a; c[1] = b; c[2] = d; }
	mat3(float a0, float a1, float a2, float b0, float b1, float b2, float d0, float d1, float d2) { c[0] = vec3(a0, a1, a2); c[1] = vec3(b0, b1, b2); c[2] = vec3(d0, d1, d2); }
	vec3& operator[](int i) { return c[i]; }
	const vec3& operator[](int i) const { return c[i]; }
};
inline vec3 operator*(const mat3& m, const vec3& v) {
	return vec3((m[0].x * v.x + m[1].x * v.y) + m[2].x * v.z, (m[0].y * v.x + m[1].y * v.y) + m[2].y * v.z, (m[0].z * v.x + m[1].z * v.y) + m[2].z * v.z);
}
inline mat3 operator-(const mat3& m) { return mat3(-m[0], -m[1], -m[2]); }
inline mat3 transpose(const mat3& m) { return mat3(vec3(m[0].x, m[1].x, m[2].x), vec3(m[0].y, m[1].y, m[2].y), vec3(m[0].z, m[1].z, m[2].z)); }
inline float determinant(const mat3& m) {
	// expansion along the first column
	return m[0].x * (m[1].y * m[2].z - m[2].y * m[1].z) - m[1].x * (m[0].y * m[2].z - m[2].y * m[0].z) + m[2].x * (m[0].y * m[1].z - m[1].y * m[0].z);
}

struct mat3x4;
struct mat4x3 {
	vec3 c[4];
	mat4x3() {}
	mat4x3(const vec3& a, const vec3& b, const vec3& d, const vec3& e) { c[0] = a; c[1] = b; c[2] = d; c[3] = e; }
	vec3& operator[](int i) { return c[i]; }
	const vec3& operator[](int i) const { return c[i]; }
};
inline vec3 operator*(const mat4x3& m, const vec4& v) {
	return vec3(((m[0].x * v.x + m[1].x * v.y) + m[2].x * v.z) + m[3].x * v.w,
		((m[0].y * v.x + m[1].y * v.y) + m[2].y * v.z) + m[3].y * v.w,
		((m[0].z * v.x + m[1].z * v.y) + m[2].z * v.z) + m[3].z * v.w);
}
inline mat4x3 operator*(const mat3& a, const mat4x3& b) { return mat4x3(a * b[0], a * b[1], a * b[2], a * b[3]); }
// transpose(mat4x3) is a 3-column, 4-row matrix; only its product with a vec3 is used
struct mat3x4 { vec3 rows_of_result[4]; };
inline mat3x4 transpose(const mat4x3& m) { mat3x4 r; for (int i = 0; i != 4; ++i) r.rows_of_result[i] = m[i]; return r; }
inline vec4 operator*(const mat3x4& m, const vec3& v) { return vec4(dot(m.rows_of_result[0], v), dot(m.rows_of_result[1], v), dot(m.rows_of_result[2], v), dot(m.rows_of_result[3], v)); }
struct mat4 { vec4 c[4]; vec4& operator[](int i) { return c[i]; } };

// ---- packing -----------------------------------------------------------------------
inline uint16_t float_to_half_bits(float f) {
	uint x = floatBitsToUint(f);
	uint sign = (x >> 16) & 0x8000u, mant = x & 0x007FFFFFu;
	int exp = (int) ((x >> 23) & 0xFF);
	if (exp == 255) return (uint16_t) (sign | 0x7C00u | (mant ? 0x200u : 0));
	int e = exp - 127 + 15;
	if (e >= 31) return (uint16_t) (sign | 0x7C00u);
	if (e <= 0) {
		if (e < -10) return (uint16_t) sign;
		mant |= 0x00800000u;
		uint shift = (uint) (14 - e);
		uint half_mant = mant >> shift, rem = mant & ((1u << shift) - 1), halfway = 1u << (shift - 1);
		if (rem > halfway || (rem == halfway && (half_mant & 1))) ++half_mant;
		return (uint16_t) (sign | half_mant);
	}
	uint half = sign | ((uint) e << 10) | (mant >> 13), rem = mant & 0x1FFFu;
	if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;
	return (uint16_t) half;
}
inline uint packHalf2x16(const vec2& v) { return (uint) float_to_half_bits(v.x) | ((uint) float_to_half_bits(v.y) << 16); }

// ---- resources ----------------------------------------------------------------------
// Opaque GLSL resource types become small views onto host memory.

struct utextureBuffer { const void* data = nullptr; int kind = 0; };  // 0: RG32_UINT, 1: R8_UINT
struct textureBuffer { const uint16_t* data = nullptr; };             // RGBA16_UNORM
struct texture2DArray { const uint16_t* data = nullptr; int width = 0, height = 0, depth = 0; };  // RGBA16_UNORM
struct sampler2DArray { const uint16_t* data = nullptr; int channels = 0, resolution = 0, layers = 0; };
// a constant texel, or a view onto an oracle_texture_t that is filtered by the oracle's sampler
// (texture filtering is the driver's in the reference; oracle.h documents the stand-in)
struct sampler2D { vec4 constant; const void* texture = nullptr; const void* light_texture = nullptr; };
extern "C" void oracle_sample_texture(const void* texture, const float uv[2], const float duv_dx[2], const float duv_dy[2], float out_rgba[4]);
struct usubpassInput { const uint32_t* data = nullptr; int width = 0; };
struct accelerationStructureEXT { const void* bvh = nullptr; int brute_force = 0; };
struct rayQueryEXT { bool hit = false; };

extern thread_local int g_current_pixel_x, g_current_pixel_y;
extern unsigned long long g_shadow_ray_count;

inline uvec4 texelFetch(const utextureBuffer& t, int i) {
	if (t.kind == 0) { const uint32_t* p = (const uint32_t*) t.data + 2 * (size_t) i; return uvec4(p[0], p[1], 0, 1); }
	return uvec4(((const uint8_t*) t.data)[i], 0, 0, 1);
}
inline vec4 texelFetch(const textureBuffer& t, int i) {
	const uint16_t* p = t.data + 4 * (size_t) i;
	return vec4((float) p[0] / 65535.0f, (float) p[1] / 65535.0f, (float) p[2] / 65535.0f, (float) p[3] / 65535.0f);
}
inline vec4 texelFetch(const texture2DArray& t, const ivec3& c, int) {
	const uint16_t* p = t.data + 4 * (((size_t) c.z * t.height + c.y) * t.width + c.x);
	return vec4((float) p[0] / 65535.0f, (float) p[1] / 65535.0f, (float) p[2] / 65535.0f, (float) p[3] / 65535.0f);
}
inline uvec4 subpassLoad(const usubpassInput& s) { return uvec4(s.data[(size_t) g_current_pixel_y * s.width + g_current_pixel_x], 0, 0, 1); }
inline vec4 textureGrad(const sampler2D& s, const vec2& uv, const vec2& dx, const vec2& dy) {
	if (!s.texture) return s.constant;
	const float uv_[2] = {uv.x, uv.y}, dx_[2] = {dx.x, dx.y}, dy_[2] = {dy.x, dy.y};
	float out[4];
	oracle_sample_texture(s.texture, uv_, dx_, dy_, out);
	return vec4(out[0], out[1], out[2], out[3]);
}
extern "C" void oracle_sample_light_texture(const void* texture, const float uv[2], float out_rgba[4]);
// only the light textures are read with textureLod (level 0)
inline vec4 textureLod(const sampler2D& s, const vec2& uv, float) {
	if (!s.light_texture) return s.constant;
	const float uv_[2] = {uv.x, uv.y};
	float out[4];
	oracle_sample_light_texture(s.light_texture, uv_, out);
	return vec4(out[0], out[1], out[2], out[3]);
}
// VK_FILTER_LINEAR, clamp to edge, nearest array layer (round to nearest even);
// weights in exact fp32, x filtered first (the oracle's documented choice)
inline vec4 textureLod(const sampler2DArray& s, const vec3& coord, float) {
	int res = s.resolution;
	float fx = coord.x * (float) res - 0.5f, fy = coord.y * (float) res - 0.5f;
	float flx = std::floor(fx), fly = std::floor(fy);
	float wx = fx - flx, wy = fy - fly;
	auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
	int x0 = clampi((int) flx, res - 1), x1 = clampi((int) flx + 1, res - 1);
	int y0 = clampi((int) fly, res - 1), y1 = clampi((int) fly + 1, res - 1);
	int layer = clampi((int) std::rint(coord.z), s.layers - 1);
	size_t base = (size_t) layer * res * res;
	vec4 out(0.0f, 0.0f, 0.0f, 1.0f);
	for (int ch = 0; ch != s.channels; ++ch) {
		auto at = [&](int x, int y) { return (float) s.data[(base + (size_t) y * res + x) * s.channels + ch] / 65535.0f; };
		float top = at(x0, y0) * (1.0f - wx) + at(x1, y0) * wx;
		float bottom = at(x0, y1) * (1.0f - wx) + at(x1, y1) * wx;
		out[ch] = top * (1.0f - wy) + bottom * wy;
	}
	return out;
}

// ray queries answered by the oracle's BVH (the reference delegates them to the driver)
extern "C" int oracle_bvh_any_hit(const void* bvh, const float origin[3], const float dir[3], float t_min, float t_max, int brute_force);
const uint gl_RayFlagsTerminateOnFirstHitEXT = 4u, gl_RayFlagsOpaqueEXT = 1u, gl_RayFlagsSkipClosestHitShaderEXT = 8u;
const uint gl_RayQueryCommittedIntersectionNoneEXT = 0u;
inline void rayQueryInitializeEXT(rayQueryEXT& q, const accelerationStructureEXT& as, uint, uint, const vec3& origin, float t_min, const vec3& dir, float t_max) {
	float o[3] = {origin.x, origin.y, origin.z}, d[3] = {dir.x, dir.y, dir.z};
	++g_shadow_ray_count;
	q.hit = oracle_bvh_any_hit(as.bvh, o, d, t_min, t_max, as.brute_force) != 0;
}
inline bool rayQueryProceedEXT(rayQueryEXT&) { return false; }
inline uint rayQueryGetIntersectionTypeEXT(const rayQueryEXT& q, bool) { return q.hit ? 1u : 0u; }

#define nonuniformEXT(x) (x)

}  // namespace glsl
