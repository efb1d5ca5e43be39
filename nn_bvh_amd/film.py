"""Host-side mirror of the film entry points (include/nnbvh.h, nnbvh_film_*): RGBFilm's pixel
accumulators on the device (/root/reference/src/pbrt/film.h:239-255, 302-307) and the tile
all-gather of a sharded render.  Buffers are torch CUDA tensors or raw device pointers; all work
happens in libnnbvh_hip.so."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check


def _dp(t):
    """device pointer of a torch tensor / int / None"""
    if t is None:
        return None
    return ctypes.c_void_p(t if isinstance(t, int) else t.data_ptr())


class Film:
    def __init__(self, xres, yres, max_component_value=float("inf"), device=0, x0=0, y0=0):
        self.bounds = (x0, y0, x0 + xres, y0 + yres)
        self.device = device
        self._h = _lib.lib().nnbvh_film_create(x0, y0, x0 + xres, y0 + yres,
                                               ctypes.c_float(max_component_value), device)
        if not self._h:
            raise _lib.NNBVHError(f"nnbvh_film_create failed: {_lib.last_error()}")
        self.n_pixels = xres * yres

    def close(self):
        if self._h:
            _lib.lib().nnbvh_film_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self, stream=0):
        check(_lib.lib().nnbvh_film_clear(self._h, ctypes.c_void_p(stream)), "nnbvh_film_clear")

    def add_samples_device(self, px, py, rgb, weight, n_per_pass, n_passes=1, rgb_stride=3, d_size=None,
                           stream=0):
        """UpdateFilm + RGBFilm::AddSample for n_passes samples of n_per_pass distinct pixels."""
        check(_lib.lib().nnbvh_film_add_samples_device(self._h, _dp(px), _dp(py), _dp(rgb), rgb_stride,
                                                       _dp(weight), n_per_pass, n_passes, _dp(d_size),
                                                       ctypes.c_void_p(stream)),
              "nnbvh_film_add_samples_device")

    def pixels_ptr(self):
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        check(_lib.lib().nnbvh_film_pixels_device(self._h, ctypes.byref(p), ctypes.byref(n)),
              "nnbvh_film_pixels_device")
        return p.value, n.value

    def read(self):
        """Synchronous copy: float64 [n_pixels, 4] = rgbSum[3], weightSum per pixel."""
        out = np.zeros((self.n_pixels, 4), np.float64)
        check(_lib.lib().nnbvh_film_read(self._h, _lib.ptr(out)), "nnbvh_film_read")
        return out

    def pack(self, d_index, n, d_out, stream=0):
        check(_lib.lib().nnbvh_film_pack_pixels_device(self._h, _dp(d_index), n, _dp(d_out),
                                                       ctypes.c_void_p(stream)), "nnbvh_film_pack_pixels_device")

    def unpack(self, d_index, n, d_in, stream=0):
        check(_lib.lib().nnbvh_film_unpack_pixels_device(self._h, _dp(d_index), n, _dp(d_in),
                                                         ctypes.c_void_p(stream)), "nnbvh_film_unpack_pixels_device")

    def gather_tiles(self, index_lists, rank, stream=None, group=None):
        """Packs this rank's pixels and all-gathers every rank's (ONE all_gather_into_tensor: RCCL over
        xGMI under the nccl backend; ranks pad to the largest shard).  Returns the float64 tensor
        [world, width, 4].  Everything is ordered on ONE stream: torch's current stream, or `stream` (a raw
        hipStream_t) made current for the duration — the collective orders itself against torch's current
        stream only, and the temporaries are then allocated on the stream that uses them."""
        import contextlib

        import torch
        import torch.distributed as dist
        world = len(index_lists)
        width = max(int(ix.numel()) for ix in index_lists)
        dev = index_lists[rank].device
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)) if stream else contextlib.nullcontext()
        with ctx:
            cur = torch.cuda.current_stream(dev).cuda_stream
            send = torch.zeros((width, 4), dtype=torch.float64, device=dev)
            self.pack(index_lists[rank], int(index_lists[rank].numel()), send, cur)
            out = torch.empty((world * width, 4), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(out, send, group=group)
        return out.view(world, width, 4)

    def all_gather_tiles(self, index_lists, rank, stream=None, group=None):
        """The film all-gather of a tile-sharded render: index_lists[r] = int32 CUDA tensor of the
        linear pixel indices rank r owns.  gather_tiles, then the other ranks' pixels are unpacked into
        this film — on the same stream.  Returns the bytes this rank sent."""
        import contextlib

        import torch
        dev = index_lists[rank].device
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)) if stream else contextlib.nullcontext()
        with ctx:
            out = self.gather_tiles(index_lists, rank, None, group)
            cur = torch.cuda.current_stream(dev).cuda_stream
            for r in range(len(index_lists)):
                if r != rank:
                    self.unpack(index_lists[r], int(index_lists[r].numel()), out[r], cur)
        return int(out.shape[1]) * 32


def pixel_rgb(pixels):
    """RGBFilm::GetPixelRGB (film.h:257-275) without splats and with an identity output transform:
    rgb = Float(rgbSum) / Float(weightSum) where the weight is non-zero.  pixels: float64 [n, 4]."""
    rgb = pixels[:, :3].astype(np.float32)
    w = pixels[:, 3].astype(np.float32)
    nz = w != 0
    rgb[nz] = rgb[nz] / w[nz, None]
    return rgb


def write_pfm(path, rgb, xres, yres):
    """Portable float map (the format pbrt's imgtool reads, util/image.cpp): rows bottom-up,
    little-endian float32."""
    img = np.asarray(rgb, np.float32).reshape(yres, xres, 3)[::-1]
    with open(path, "wb") as f:
        f.write(f"PF\n{xres} {yres}\n-1.0\n".encode())
        f.write(np.ascontiguousarray(img, "<f4").tobytes())
