"""ImageProcessor -- mirror of scripts/utils/image_processor.py (same constructor, attributes and methods).
The kernels it hands out (`get_kernel`) are kept for API compatibility; the flatness stencil itself runs
inside the fused HIP score-map kernel (Gaussian 5x5 sigma=size/6 o Sobel 3x3, reflect padding)."""
import colorsys

import numpy as np
import torch


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

    def smooth_depth(self, depth_patch, device):  # :56-64 (kept for callers; not on the hot path any more)
        import torch.nn.functional as F

        g = self.get_kernel("gaussian", device)
        d = depth_patch.to(device)
        padded = F.pad(d.unsqueeze(0).unsqueeze(0), (g.shape[0] // 2,) * 4, mode="reflect")
        return F.conv2d(padded, g.view(1, 1, *g.shape), padding=0).squeeze()
