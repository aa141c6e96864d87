// The error-display variants of the shading kernel (ERROR_DISPLAY_DIFFUSE /
// ERROR_DISPLAY_SPECULAR of the reference, src/main.c:728-750, 788-790).  With an error
// display the per-light program returns before any sample is shaded, so only two code
// paths exist: the diffuse-only preparation (strategies diffuse_only, diffuse_ggx_mis) and
// the combined diffuse + specular preparation (the other three strategies); no rays.
// Built once per arithmetic mode.
#include "shading_kernel.h"

#if VKR_FAST_MATH
#define VKR_ERROR_LAUNCH_NAME vkr_launch_error_display_fast
#define VKR_RESOLVE_LAUNCH_NAME vkr_launch_resolve_materials_fast
#elif VKR_MATH_MODE == 2
#define VKR_ERROR_LAUNCH_NAME vkr_launch_error_display_libm
#define VKR_RESOLVE_LAUNCH_NAME vkr_launch_resolve_materials_libm
#else
#define VKR_ERROR_LAUNCH_NAME vkr_launch_error_display_exact
#define VKR_RESOLVE_LAUNCH_NAME vkr_launch_resolve_materials_exact
#endif

using namespace vkr;

template <int STRATEGY, int TECHNIQUE, int ERROR>
static int launch_capacity(int capacity, const shade_params& p, dim3 grid, hipStream_t stream) {
	switch (capacity) {
	case 4: shade_pixels<STRATEGY, TECHNIQUE, 4, kRaysNone, ERROR><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(STRATEGY, TECHNIQUE, 4, ERROR), stream>>>(p); break;
	case 5: shade_pixels<STRATEGY, TECHNIQUE, 5, kRaysNone, ERROR><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(STRATEGY, TECHNIQUE, 5, ERROR), stream>>>(p); break;
	case 6: shade_pixels<STRATEGY, TECHNIQUE, 6, kRaysNone, ERROR><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(STRATEGY, TECHNIQUE, 6, ERROR), stream>>>(p); break;
	case 7: shade_pixels<STRATEGY, TECHNIQUE, 7, kRaysNone, ERROR><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(STRATEGY, TECHNIQUE, 7, ERROR), stream>>>(p); break;
	case 8: shade_pixels<STRATEGY, TECHNIQUE, 8, kRaysNone, ERROR><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(STRATEGY, TECHNIQUE, 8, ERROR), stream>>>(p); break;
	default: return -1;
	}
	return hipGetLastError() != hipSuccess;
}

template <int STRATEGY, int ERROR>
static int launch_technique(int technique, int capacity, const shade_params& p, dim3 grid, hipStream_t stream) {
	if (technique == kTechniquePsa) return launch_capacity<STRATEGY, kTechniquePsa, ERROR>(capacity, p, grid, stream);
	if (technique == kTechniquePsaBiased) return launch_capacity<STRATEGY, kTechniquePsaBiased, ERROR>(capacity, p, grid, stream);
	// Arvo's sampler has its own error function, in the diffuse-only preparation only
	if constexpr (STRATEGY == kStrategyDiffuseOnly && ERROR == kErrorDiffuse)
		if (technique == kTechniquePsaArvo) return launch_capacity<STRATEGY, kTechniquePsaArvo, ERROR>(capacity, p, grid, stream);
	return -1;
}

// combined_path: 0 for the strategies that prepare only the diffuse technique, 1 otherwise.
// Returns 0 on success, 1 on a launch error, -1 if the combination does not exist.
extern "C" int VKR_ERROR_LAUNCH_NAME(int combined_path, int technique, int capacity, int error_mode, const shade_params* p, unsigned int grid_x, void* stream) {
	dim3 grid(grid_x, 1, 1);
	hipStream_t s = (hipStream_t) stream;
	if (!combined_path && error_mode == kErrorDiffuse) return launch_technique<kStrategyDiffuseOnly, kErrorDiffuse>(technique, capacity, *p, grid, s);
	if (combined_path && error_mode == kErrorDiffuse) return launch_technique<kStrategyMis, kErrorDiffuse>(technique, capacity, *p, grid, s);
	if (combined_path && error_mode == kErrorSpecular) return launch_technique<kStrategyMis, kErrorSpecular>(technique, capacity, *p, grid, s);
	return -1;
}

// ---- material resolve (textured scenes), in this unit's arithmetic ------------------------------

// One lane per pixel of the frame: the three texture reads of get_shading_data
// (shading_pass.frag.glsl:754-785) with their screen-space derivatives, written as the eight
// numbers of a constant material so that the shading kernels stay as they are.
inline namespace VKR_MODE_NAMESPACE {
__global__ void __launch_bounds__(256) k_resolve_materials(const shade_params p, float* pixel_materials) {
	uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint32_t px = blockIdx.x * 16 + ((wave & 1) << 3) + (lane & 7);
	uint32_t py = blockIdx.y * 16 + ((wave >> 1) << 3) + (lane >> 3);
	if (px >= p.width || py >= p.height) return;
	uint32_t primitive = p.visibility[(size_t) py * p.width + px];
	if (primitive == 0xFFFFFFFFu) return;
	const uint8_t* c = p.constants;
	float fx = (float) (int32_t) px, fy = (float) (int32_t) py;
	f3 ray = mk3(
		(load_f(c, 96) * fx + load_f(c, 100) * fy) + load_f(c, 104) * 1.0f,
		(load_f(c, 112) * fx + load_f(c, 116) * fy) + load_f(c, 120) * 1.0f,
		(load_f(c, 128) * fx + load_f(c, 132) * fy) + load_f(c, 136) * 1.0f);
	float values[8];
	resolve_material(p, primitive, ray, values);
	float4* out = (float4*) (pixel_materials + 8 * ((size_t) py * p.width + px));
	out[0] = make_float4(values[0], values[1], values[2], values[3]);
	out[1] = make_float4(values[4], values[5], values[6], values[7]);
}
}

extern "C" int VKR_RESOLVE_LAUNCH_NAME(const shade_params* p, float* pixel_materials, void* stream) {
	dim3 grid((p->width + 15) / 16, (p->height + 15) / 16);
	k_resolve_materials<<<grid, 256, 0, (hipStream_t) stream>>>(*p, pixel_materials);
	return hipGetLastError() != hipSuccess;
}
