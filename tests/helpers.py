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
