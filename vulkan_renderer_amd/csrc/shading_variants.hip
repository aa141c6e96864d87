// Instantiates the shade_pixels variants of one sampling strategy and one
// arithmetic mode.  Built once per (VKR_STRATEGY, VKR_MATH_MODE) pair so that the
// translation units compile in parallel; the libm- and exact-mode units are compiled with
// -ffp-contract=off, the fast-mode units with -ffp-contract=fast.
#include "shading_kernel.h"

#ifndef VKR_STRATEGY
#error "define VKR_STRATEGY (0..4)"
#endif

// VKR_LIGHT_TEXTURES=1 builds the same matrix once more with light textures compiled in
// (get_polygon_radiance of the reference with a texturing technique, shading_pass.frag.glsl:151-185)
#ifndef VKR_LIGHT_TEXTURES
#define VKR_LIGHT_TEXTURES 0
#endif

#define VKR_CAT2(a, b, c, d) a##b##c##d
#define VKR_CAT(a, b, c, d) VKR_CAT2(a, b, c, d)
#if VKR_LIGHT_TEXTURES
#define VKR_MODE kLightTextures
#if VKR_FAST_MATH
#define VKR_LAUNCH_NAME VKR_CAT(vkr_launch_shade_textured_fast_, VKR_STRATEGY, , )
#elif VKR_MATH_MODE == 2
#define VKR_LAUNCH_NAME VKR_CAT(vkr_launch_shade_textured_libm_, VKR_STRATEGY, , )
#else
#define VKR_LAUNCH_NAME VKR_CAT(vkr_launch_shade_textured_exact_, VKR_STRATEGY, , )
#endif
#else
#define VKR_MODE kErrorNone
#if VKR_FAST_MATH
#define VKR_LAUNCH_NAME VKR_CAT(vkr_launch_shade_fast_, VKR_STRATEGY, , )
#elif VKR_MATH_MODE == 2
#define VKR_LAUNCH_NAME VKR_CAT(vkr_launch_shade_libm_, VKR_STRATEGY, , )
#else
#define VKR_LAUNCH_NAME VKR_CAT(vkr_launch_shade_exact_, VKR_STRATEGY, , )
#endif
#endif

using namespace vkr;

// VKR_EXPERIMENT_EXTRA_LDS=<bytes> (a measurement aid; frames are unchanged): every shading workgroup asks for that much more
// LDS than it uses, i.e. fewer of them are resident per CU - how the kernel's speed depends on its resident waves, measured
// downwards from the nine that fit at V = 7 (profiles/r10f)
static uint32_t extra_lds() {
	static const uint32_t bytes = [] { const char* text = getenv("VKR_EXPERIMENT_EXTRA_LDS"); return text ? (uint32_t) strtoul(text, NULL, 10) : 0u; }();
	return bytes;
}
#define shade_lds_bytes(...) (shade_lds_bytes(__VA_ARGS__) + extra_lds())

template <int TECHNIQUE, int V>
static int launch_rays(int rays, const shade_params& p, dim3 grid, hipStream_t stream) {
	if (rays == kRaysDeferred) shade_pixels<VKR_STRATEGY, TECHNIQUE, V, kRaysDeferred, VKR_MODE><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(VKR_STRATEGY, TECHNIQUE, V, VKR_MODE), stream>>>(p);
#if VKR_STRATEGY >= 2
	// block-wise reservation of queue slots: built for the strategies that sample two techniques
	// per light and sample, where a lane queues many rays (the host picks it from 8 rays per lane)
	else if (rays == kRaysDeferredBlocks) shade_pixels<VKR_STRATEGY, TECHNIQUE, V, kRaysDeferredBlocks, VKR_MODE><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(VKR_STRATEGY, TECHNIQUE, V, VKR_MODE), stream>>>(p);
#else
	else if (rays == kRaysDeferredBlocks) return -1;
#endif
	else if (rays == kRaysInline) shade_pixels<VKR_STRATEGY, TECHNIQUE, V, kRaysInline, VKR_MODE><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(VKR_STRATEGY, TECHNIQUE, V, VKR_MODE), stream>>>(p);
	else shade_pixels<VKR_STRATEGY, TECHNIQUE, V, kRaysNone, VKR_MODE><<<dim3(shade_grid_size(grid.x)), kShadeThreads, shade_lds_bytes(VKR_STRATEGY, TECHNIQUE, V, VKR_MODE), stream>>>(p);
	return hipGetLastError() != hipSuccess;
}

template <int TECHNIQUE>
static int launch_capacity(int capacity, int rays, const shade_params& p, dim3 grid, hipStream_t stream) {
	// techniques that clip the polygon at the horizon need one more vertex slot (reference main.c:194-216)
	constexpr bool kClips = TECHNIQUE == kTechniquePsa || TECHNIQUE == kTechniquePsaBiased || TECHNIQUE == kTechniqueClippedSolidAngle || TECHNIQUE == kTechniqueHartBilinearClipping
		|| TECHNIQUE == kTechniqueHartBiquadraticClipping || TECHNIQUE == kTechniquePsaArvo;
	switch (capacity) {
	case 3: if constexpr (!kClips) return launch_rays<TECHNIQUE, 3>(rays, p, grid, stream); else return -1;
	case 4: return launch_rays<TECHNIQUE, 4>(rays, p, grid, stream);
	case 5: return launch_rays<TECHNIQUE, 5>(rays, p, grid, stream);
	case 6: return launch_rays<TECHNIQUE, 6>(rays, p, grid, stream);
	case 7: return launch_rays<TECHNIQUE, 7>(rays, p, grid, stream);
	case 8: if constexpr (kClips) return launch_rays<TECHNIQUE, 8>(rays, p, grid, stream); else return -1;
	default: return -1;
	}
}

// Returns 0 on success, 1 on a launch error, -1 if this combination is not built
extern "C" int VKR_LAUNCH_NAME(int technique, int capacity, int rays, const shade_params* p, unsigned int grid_x, void* stream) {
	dim3 grid(grid_x, 1, 1);
	hipStream_t s = (hipStream_t) stream;
	switch (technique) {
	case kTechniquePsa: return launch_capacity<kTechniquePsa>(capacity, rays, *p, grid, s);
	case kTechniquePsaBiased: return launch_capacity<kTechniquePsaBiased>(capacity, rays, *p, grid, s);
#if VKR_STRATEGY == 0 || VKR_STRATEGY == 1
	// the solid-angle samplers only pair with these two strategies
	// (reference shading_pass.frag.glsl:305-323 returns black otherwise)
	case kTechniqueSolidAngle: return launch_capacity<kTechniqueSolidAngle>(capacity, rays, *p, grid, s);
	case kTechniqueClippedSolidAngle: return launch_capacity<kTechniqueClippedSolidAngle>(capacity, rays, *p, grid, s);
	// related work whose shader branches define a solid angle for the GGX MIS tail
	case kTechniqueUrena: return launch_capacity<kTechniqueUrena>(capacity, rays, *p, grid, s);
	case kTechniqueArvoSolidAngle: return launch_capacity<kTechniqueArvoSolidAngle>(capacity, rays, *p, grid, s);
	case kTechniquePsaArvo: return launch_capacity<kTechniquePsaArvo>(capacity, rays, *p, grid, s);
#endif
#if VKR_STRATEGY == 0
	// the two simplest related-work techniques of the reference's comparison set; their shader
	// branches only exist for the diffuse-only strategy (shading_pass.frag.glsl:332-350, :676-686)
	case kTechniqueBaseline: return launch_capacity<kTechniqueBaseline>(capacity, rays, *p, grid, s);
	case kTechniqueAreaTurk: return launch_capacity<kTechniqueAreaTurk>(capacity, rays, *p, grid, s);
	case kTechniqueHartBilinear: return launch_capacity<kTechniqueHartBilinear>(capacity, rays, *p, grid, s);
	case kTechniqueHartBilinearClipping: return launch_capacity<kTechniqueHartBilinearClipping>(capacity, rays, *p, grid, s);
	case kTechniqueHartBiquadratic: return launch_capacity<kTechniqueHartBiquadratic>(capacity, rays, *p, grid, s);
	case kTechniqueHartBiquadraticClipping: return launch_capacity<kTechniqueHartBiquadraticClipping>(capacity, rays, *p, grid, s);
#endif
	default: return -1;
	}
}
