"""Shared helpers of the test-suite (not collected by pytest)."""
import numpy as np

import oracle
from vulkan_renderer_amd import renderer, synthetic


def oracle_render(host_scene, visibility=None, math_mode=1, bvh=None, width=None, height=None):
    """Shades the host scene with the CPU oracle.  Returns (image, inputs, bvh)."""
    inputs = host_scene.host_inputs()
    e = host_scene.app.swapchain.extent
    if bvh is None:
        bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    if visibility is None:
        cam = host_scene.app.scene_specification.camera
        visibility = oracle.primary_visibility(inputs["constants"], bvh, e.width, e.height, cam.near, cam.far)
    inputs["visibility"] = visibility
    frame = oracle.make_frame(inputs, host_scene.oracle_settings(), bvh)
    oracle.set_math_mode(math_mode)
    try:
        image = oracle.shade(frame)
    finally:
        oracle.set_math_mode(0)
    return image, inputs, bvh


def compare(gpu, cpu):
    """Error statistics over RGB of exposure-scaled radiance."""
    a, b = gpu[..., :3].astype(np.float64), cpu[..., :3].astype(np.float64)
    diff = np.abs(a - b)
    per_pixel = diff.max(axis=-1)
    return {
        "rmse": float(np.sqrt(((a - b) ** 2).mean())),
        "max_abs": float(diff.max()),
        "mismatched_pixels": int((per_pixel > 0).sum()),
        "pixels_over_1e-3": int((per_pixel > 1e-3).sum()),
        "nan": int(np.isnan(gpu).sum()),
        "bit_exact": bool(np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))),
    }


class DeviceBuffer:
    """Plain HIP device memory through ctypes on the HIP runtime that
    libvkr_shading.so already loaded (keeps torch out of the GPU tests)."""
    import ctypes as _C
    _hip = None

    def __init__(self, nbytes):
        C = self._C
        if DeviceBuffer._hip is None:
            DeviceBuffer._hip = C.CDLL("libamdhip64.so")
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        assert DeviceBuffer._hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0
        assert DeviceBuffer._hip.hipMemset(self.ptr, 0, C.c_size_t(nbytes)) == 0

    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        assert DeviceBuffer._hip.hipMemcpy(self.ptr, self._C.c_void_p(a.ctypes.data), self._C.c_size_t(a.nbytes), 1) == 0

    def download(self, shape, dtype):
        out = np.zeros(shape, dtype)
        assert out.nbytes <= self.nbytes
        assert DeviceBuffer._hip.hipDeviceSynchronize() == 0
        assert DeviceBuffer._hip.hipMemcpy(self._C.c_void_p(out.ctypes.data), self.ptr, self._C.c_size_t(out.nbytes), 2) == 0
        return out

    def zero(self):
        assert DeviceBuffer._hip.hipMemset(self.ptr, 0, self._C.c_size_t(self.nbytes)) == 0
        assert DeviceBuffer._hip.hipDeviceSynchronize() == 0

    def free(self):
        if self.ptr:
            DeviceBuffer._hip.hipFree(self.ptr)
            self.ptr = None


# ---- pixels on a discontinuity of the shader ---------------------------------------------------
#
# Two arithmetics that agree to a few ulp per operation (libm vs polynomial transcendentals, IEEE vs
# approximate reciprocals) produce frames that agree to ~1e-5 everywhere except at pixels where a
# last-bit difference is pushed through a discontinuity of the shader itself:
#   guard        the NaN guard of the reference (src/shaders/shading_pass.frag.glsl:861-864): a sliver
#                sector makes normalize_approx_and_flip(0) = NaN ("undefined if rhs is zero",
#                polygon_sampling.glsl:597-611) and the pixel is painted (1, 0, 0.8); whether a sample
#                lands in such a sector is decided by the last bits of the sector areas
#   silhouette   a shadow ray whose direction differs in the last bits passes a triangle edge on the
#                other side: one whole estimator term appears or disappears (ray query, :120-138).
#                Recognised mechanically: the two frames agree at that pixel when rendered WITHOUT
#                shadow rays
#   other        anything else - must not exist
GUARD_COLOR = np.array([1.0, 0.0, 0.8])


def is_guard_pixel(image):
    return np.abs(image[..., :3].astype(np.float64) - GUARD_COLOR).max(axis=-1) < 1.0e-5


def classify_outliers(a, b, a_without_rays=None, b_without_rays=None, threshold=1.0e-2):
    """Statistics of frame `a` against frame `b` (same scene and settings, two arithmetics) with
    every pixel that differs by more than `threshold` put into one of the classes above.
    The frames without shadow rays are only needed if such pixels exist outside the guard class."""
    da = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    per_pixel = np.abs(np.nan_to_num(da, nan=1.0e3)).max(axis=-1)
    outlier = per_pixel > threshold
    guard = outlier & (is_guard_pixel(a) | is_guard_pixel(b))
    rest = outlier & ~guard
    silhouette = np.zeros_like(rest)
    if rest.any() and a_without_rays is not None and b_without_rays is not None:
        dn = np.abs(a_without_rays[..., :3].astype(np.float64) - b_without_rays[..., :3].astype(np.float64)).max(axis=-1)
        silhouette = rest & (dn <= threshold)
    other = rest & ~silhouette
    inlier = ~outlier
    return {
        "rmse": float(np.sqrt((np.nan_to_num(da, nan=1.0e3) ** 2).mean())),
        "rmse_without_outliers": float(np.sqrt((da[inlier] ** 2).sum() / da.size)),
        "pixels_over_threshold": int(outlier.sum()), "threshold": threshold,
        "guard_pixels": int(guard.sum()), "silhouette_pixels": int(silhouette.sum()), "other_pixels": int(other.sum()),
        "guard_pixels_in_a": int(is_guard_pixel(a).sum()), "guard_pixels_in_b": int(is_guard_pixel(b).sum()),
        "other_coordinates": [tuple(int(v) for v in yx) for yx in np.argwhere(other)[:8]],
    }
