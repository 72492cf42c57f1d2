"""ImageProcessor -- mirror of scripts/utils/image_processor.py (same constructor, attributes and methods).

`smooth_depth` runs in liblgrasp.so (`lg_smooth_depth`: reflect padding + the separable form of the object's own
Gaussian); the flatness stencil of the scoring path runs inside the fused HIP plane kernel with the Gaussian size of
the ImageProcessor the caller hands to `select_grasp_point` (`flatness_config`).  There is no CPU / torch fallback.
The kernels `get_kernel` hands out are the reference's tensors (API compatibility; the library multiplies with the
1-D factor of the same Gaussian, `lg_gaussian_taps`)."""
import colorsys
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

_SOBEL_X = ((-1.0, 0.0, 1.0), (-2.0, 0.0, 2.0), (-1.0, 0.0, 1.0))
_handles = {}   # device index -> lg_handle for the stateless lg_smooth_depth calls (created once per device, never destroyed)


def _handle(index):
    h = _handles.get(index)
    if h is None:
        h = C.c_void_p()
        check(None, lib.lg_create(index, C.byref(h)), "lg_create")
        _handles[index] = h
    return h


def gaussian_taps(size):
    """The 1-D factor the library multiplies with (outer(t, t) == _create_gaussian_kernel(size) up to rounding)."""
    out = (C.c_float * int(size))()
    if lib.lg_gaussian_taps(int(size), out) != 0:
        raise ValueError(f"gaussian size {size} is outside [1, 15]")
    return np.array(out, np.float32)


def _kernel(image_processor, name):
    k = image_processor.get_kernel(name, "cpu")
    if k is None:
        raise ValueError(f"image_processor has no '{name}' kernel")
    return np.asarray(torch.as_tensor(k).detach().cpu(), np.float64)


def gaussian_config(image_processor):
    """Size of the object's smoothing kernel, after checking that it IS `_create_gaussian_kernel(size)` -- the library
    applies that Gaussian (as its two 1-D factors) and nothing else: a different kernel is refused, not replaced."""
    g = _kernel(image_processor, "gaussian")
    if g.ndim != 2 or g.shape[0] != g.shape[1]:
        raise ValueError(f"gaussian kernel of shape {g.shape} is not square")
    size = int(g.shape[0])
    t = gaussian_taps(size).astype(np.float64)
    if not np.allclose(g, np.outer(t, t), rtol=1e-5, atol=1e-9):
        raise ValueError("image_processor carries a smoothing kernel other than _create_gaussian_kernel(size): unsupported")
    return size


def flatness_config(image_processor):
    """Gaussian size of the smoothing `_calculate_flatness_map` (grasp_point_selector.py:635-657) applies with this
    ImageProcessor; the gradient kernels it reads from the same object must be the reference's Sobel pair."""
    # checked once per object and set of kernel tensors (select_grasp_point calls this on its latency path: three get_kernel
    # calls, an allclose and a library call per frame otherwise)
    ks = getattr(image_processor, "kernels", None)
    key = tuple(id(ks.get(n)) for n in ("gaussian", "sobel_x", "sobel_y")) if isinstance(ks, dict) else None
    cached = getattr(image_processor, "_lg_flatness_config", None)
    if key is not None and cached is not None and cached[0] == key:
        return cached[1]
    size = gaussian_config(image_processor)
    sx, sy = _kernel(image_processor, "sobel_x"), _kernel(image_processor, "sobel_y")
    if sx.shape != (3, 3) or not np.array_equal(sx, np.array(_SOBEL_X)) or not np.array_equal(sy, np.array(_SOBEL_X).T):
        raise ValueError("image_processor carries gradient kernels other than the 3x3 Sobel pair: unsupported")
    if key is not None:
        try:
            image_processor._lg_flatness_config = (key, size)
        except Exception:  # noqa: BLE001   (an object that takes no attributes: checked every time)
            pass
    return size


class ImageProcessor:
    def __init__(self, height, width, kernel_size, gaussian_size):  # image_processor.py:9-13
        self.height = height
        self.width = width
        self.gaussian_size = gaussian_size
        self.kernels = self._initialize_kernels(kernel_size, gaussian_size)
        self.color_map = {}

    def _initialize_kernels(self, kernel_size, gaussian_size):  # :15-23
        kernels = {
            "isolation": torch.ones(kernel_size, kernel_size),
            "sobel_x": torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32),
        }
        kernels["sobel_y"] = kernels["sobel_x"].t()
        kernels["gaussian"] = self._create_gaussian_kernel(gaussian_size)
        return kernels

    def _create_gaussian_kernel(self, size):  # :25-32
        sigma = size / 6.0
        center = size // 2
        x, y = np.meshgrid(np.arange(size), np.arange(size))
        kernel = np.exp(-((x - center) ** 2 + (y - center) ** 2) / (2 * sigma ** 2))
        kernel = kernel / kernel.sum()
        return torch.tensor(kernel, dtype=torch.float32)

    def get_kernel(self, name, device):  # :34-38
        if name in self.kernels:
            return self.kernels[name].to(device)
        return None

    def generate_color(self, leaf_id):  # :40-47
        if leaf_id not in self.color_map:
            hue = (leaf_id * 0.618033988749895) % 1.0
            rgb = colorsys.hsv_to_rgb(hue, 0.8, 0.95)
            self.color_map[leaf_id] = tuple(int(255 * x) for x in rgb)
        return self.color_map[leaf_id]

    def calculate_centroid(self, leaf_mask):  # :49-54
        y_indices, x_indices = torch.where(leaf_mask)
        return (float(x_indices.float().mean()), float(y_indices.float().mean()))

    def smooth_depth(self, depth_patch, device):  # :56-64
        """Reflect padding by size // 2, then the size x size Gaussian: [H, W] -> [H, W] for odd sizes and, as F.conv2d
        gives there, [H + 1, W + 1] for even ones.  Runs in liblgrasp.so on `device` (a HIP device)."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"ImageProcessor.smooth_depth needs a HIP device (got '{dev}'): there is no CPU fallback")
        size = gaussian_config(self)
        d = torch.as_tensor(depth_patch).to(dev, torch.float32).contiguous()
        if d.dim() != 2:
            raise ValueError(f"depth_patch must be [H, W], got {tuple(d.shape)}")
        H, W = d.shape
        p = size // 2
        out = torch.empty((H + 2 * p - size + 1, W + 2 * p - size + 1), dtype=torch.float32, device=dev)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            h = _handle(index)
            # (stateless in the library: the shared per-device handle is only read; a failure's reason is per thread)
            check(None, lib.lg_smooth_depth(h, d.data_ptr(), 1, H, W, size, out.data_ptr(),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "lg_smooth_depth")
        return out.squeeze()
