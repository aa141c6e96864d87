// The error-display variants of the shading kernel (ERROR_DISPLAY_DIFFUSE /
// ERROR_DISPLAY_SPECULAR of the reference, src/main.c:728-750, 788-790).  With an error
// display the per-light program returns before any sample is shaded, so only two code
// paths exist: the diffuse-only preparation (strategies diffuse_only, diffuse_ggx_mis) and
// the combined diffuse + specular preparation (the other three strategies); no rays.
// Built once per arithmetic mode.
#include "shading_kernel.h"

#if VKR_FAST_MATH
#define VKR_ERROR_LAUNCH_NAME vkr_launch_error_display_fast
#else
#define VKR_ERROR_LAUNCH_NAME vkr_launch_error_display_exact
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
