// film.hip — the film side of the tile-sharded render: RGBFilm's pixel accumulators on the device.
//
// Reference (/root/reference/src/pbrt):
//   RGBFilm::Pixel         film.h:302-307   double rgbSum[3], double weightSum (+ splat atomics, not used here)
//   RGBFilm::AddSample     film.h:239-255   clamp to maxComponentValue, rgbSum[c] += weight * rgb[c], weightSum += weight
//   UpdateFilm             wavefront/film.cpp:13-40  one pixel sample per work item, skipped outside the pixel bounds
// The spectral part of AddSample (sensor->ToSensorRGB(L, lambda), film.h:95-100) belongs to the
// caller's sensor model; samples arrive here as sensor RGB.
//
// Multi-GPU: every rank accumulates the pixels of its own image tiles; nnbvh_film_pack_pixels /
// _unpack_pixels move an index list of pixels (32 B each) between the film and a contiguous
// buffer, which is what the RCCL all-gather of the per-tile film samples sends.
//
// All kernels are streaming passes: HBM-bound, 32 B read + 32 B written per pixel touched.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/nnbvh.h"
#include "nnbvh_internal.h"

struct nnbvh_film {
    int device = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;  // pixel bounds [x0, x1) x [y0, y1)
    float max_component = 0.0f;
    double *d_pixels = nullptr;  // 4 doubles per pixel, row-major over the bounds
};

namespace nnbvh {

static constexpr int kFilmBlock = 256;

// item (pass, i) = sample `pass` of pixel slot i; the slots' pixels are distinct (the reference's
// pixelIndex is unique within a stage), so the read-modify-write needs no atomics, and a slot's
// passes are added in order: the sums are those of the reference's sample loop, bit for bit.
__global__ __launch_bounds__(kFilmBlock) void film_add_samples(
    double *__restrict__ pixels, int x0, int y0, int x1, int y1, float maxComponent,
    const int32_t *__restrict__ px, const int32_t *__restrict__ py, const float *__restrict__ rgb,
    int rgbStride, const float *__restrict__ weight, int nPerPass, int nPasses,
    const int32_t *__restrict__ nDev) {
    int n = nPerPass;
    if (nDev) {
        const int nd = *nDev;
        n = nd < 0 ? 0 : (nd < n ? nd : n);
    }
    for (int i = blockIdx.x * kFilmBlock + threadIdx.x; i < n; i += gridDim.x * kFilmBlock) {
        const int x = px[i], y = py[i];
        if (x < x0 || x >= x1 || y < y0 || y >= y1) continue;  // wavefront/film.cpp:18-19
        double2 *dst = reinterpret_cast<double2 *>(pixels + 4 * ((long)(y - y0) * (x1 - x0) + (x - x0)));
        double2 a = dst[0], b = dst[1];
        for (int pass = 0; pass < nPasses; ++pass) {
            const long k = (long)pass * nPerPass + i;
            float r = rgb[k * rgbStride], g = rgb[k * rgbStride + 1], bl = rgb[k * rgbStride + 2];
            const float w = weight ? weight[k] : 1.0f;
            // film.h:245-247: std::max({r, g, b})
            float m = r;
            if (m < g) m = g;
            if (m < bl) m = bl;
            if (m > maxComponent) {
                const float s = maxComponent / m;
                r *= s;
                g *= s;
                bl *= s;
            }
            a.x += (double)(w * r);  // film.h:252-254: float product, double sum
            a.y += (double)(w * g);
            b.x += (double)(w * bl);
            b.y += (double)w;
        }
        dst[0] = a;
        dst[1] = b;
    }
}

__global__ __launch_bounds__(kFilmBlock) void film_pack(const double2 *__restrict__ pixels,
                                                        const int32_t *__restrict__ index, long n,
                                                        double2 *__restrict__ out, long nPixels) {
    for (long i = (long)blockIdx.x * kFilmBlock + threadIdx.x; i < n; i += (long)gridDim.x * kFilmBlock) {
        const long p = index[i];
        const bool inside = p >= 0 && p < nPixels;  // an index outside the film packs zeros, never faults
        out[2 * i] = inside ? pixels[2 * p] : make_double2(0, 0);
        out[2 * i + 1] = inside ? pixels[2 * p + 1] : make_double2(0, 0);
    }
}

__global__ __launch_bounds__(kFilmBlock) void film_unpack(double2 *__restrict__ pixels,
                                                          const int32_t *__restrict__ index, long n,
                                                          const double2 *__restrict__ in, long nPixels) {
    for (long i = (long)blockIdx.x * kFilmBlock + threadIdx.x; i < n; i += (long)gridDim.x * kFilmBlock) {
        const long p = index[i];
        if (p < 0 || p >= nPixels) continue;  // outside the film: skipped
        pixels[2 * p] = in[2 * i];
        pixels[2 * p + 1] = in[2 * i + 1];
    }
}

static bool film_hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

struct FilmDeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit FilmDeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = film_hip_ok(hipSetDevice(dev), "hipSetDevice");
    }
    ~FilmDeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static int film_grid(long n) {
    long blocks = (n + kFilmBlock - 1) / kFilmBlock;
    if (blocks < 1) blocks = 1;
    return (int)(blocks < 256 * 16 ? blocks : 256 * 16);
}

}  // namespace nnbvh

using namespace nnbvh;

extern "C" {

nnbvh_film *nnbvh_film_create(int32_t x0, int32_t y0, int32_t x1, int32_t y1, float max_component_value,
                              int device) {
    if (x1 <= x0 || y1 <= y0 || (int64_t)(x1 - x0) * (y1 - y0) >= 0x7fffffffLL ||
        !(max_component_value > 0.0f)) {
        set_error("film_create: empty or oversized pixel bounds, or max_component_value <= 0");
        return nullptr;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        set_error("film_create: no usable HIP device");
        return nullptr;
    }
    FilmDeviceGuard guard(device);
    if (!guard.ok) return nullptr;
    nnbvh_film *f = new nnbvh_film;
    f->device = device;
    f->x0 = x0;
    f->y0 = y0;
    f->x1 = x1;
    f->y1 = y1;
    f->max_component = max_component_value;
    const size_t bytes = (size_t)(x1 - x0) * (y1 - y0) * 4 * sizeof(double);
    if (!film_hip_ok(hipMalloc((void **)&f->d_pixels, bytes), "film_create: hipMalloc") ||
        !film_hip_ok(hipMemset(f->d_pixels, 0, bytes), "film_create: hipMemset")) {
        if (f->d_pixels) (void)hipFree(f->d_pixels);
        delete f;
        return nullptr;
    }
    return f;
}

void nnbvh_film_destroy(nnbvh_film *f) {
    if (!f) return;
    FilmDeviceGuard guard(f->device);
    if (f->d_pixels) (void)hipFree(f->d_pixels);
    delete f;
}

int nnbvh_film_clear(nnbvh_film *f, void *stream) {
    if (!f) {
        set_error("film_clear: null film");
        return NNBVH_ERR_ARG;
    }
    FilmDeviceGuard guard(f->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    const size_t bytes = (size_t)(f->x1 - f->x0) * (f->y1 - f->y0) * 4 * sizeof(double);
    return film_hip_ok(hipMemsetAsync(f->d_pixels, 0, bytes, (hipStream_t)stream), "film_clear")
               ? NNBVH_OK
               : NNBVH_ERR_DEVICE;
}

int nnbvh_film_pixels_device(nnbvh_film *f, void **d_pixels, int64_t *n_pixels) {
    if (!f || !d_pixels || !n_pixels) {
        set_error("film_pixels_device: null argument");
        return NNBVH_ERR_ARG;
    }
    *d_pixels = f->d_pixels;
    *n_pixels = (int64_t)(f->x1 - f->x0) * (f->y1 - f->y0);
    return NNBVH_OK;
}

int nnbvh_film_read(nnbvh_film *f, double *out) {
    if (!f || !out) {
        set_error("film_read: null argument");
        return NNBVH_ERR_ARG;
    }
    FilmDeviceGuard guard(f->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    const size_t bytes = (size_t)(f->x1 - f->x0) * (f->y1 - f->y0) * 4 * sizeof(double);
    if (!film_hip_ok(hipDeviceSynchronize(), "film_read") ||
        !film_hip_ok(hipMemcpy(out, f->d_pixels, bytes, hipMemcpyDeviceToHost), "film_read"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_film_add_samples_device(nnbvh_film *f, const int32_t *d_px, const int32_t *d_py,
                                  const float *d_rgb, int32_t rgb_stride, const float *d_weight,
                                  int32_t n_per_pass, int32_t n_passes, const int32_t *d_size,
                                  void *stream) {
    if (!f || n_per_pass < 0 || n_passes < 0 || rgb_stride < 3 ||
        (n_per_pass > 0 && n_passes > 0 && (!d_px || !d_py || !d_rgb))) {
        set_error("film_add_samples_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if ((int64_t)n_per_pass * n_passes >= 0x7fffffffLL) {
        set_error("film_add_samples_device: at most 2^31-1 samples per call");
        return NNBVH_ERR_ARG;
    }
    if (n_per_pass == 0 || n_passes == 0) return NNBVH_OK;
    FilmDeviceGuard guard(f->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    hipLaunchKernelGGL(film_add_samples, dim3(film_grid(n_per_pass)), dim3(kFilmBlock), 0,
                       (hipStream_t)stream, f->d_pixels, f->x0, f->y0, f->x1, f->y1, f->max_component,
                       d_px, d_py, d_rgb, rgb_stride, d_weight, n_per_pass, n_passes, d_size);
    return film_hip_ok(hipGetLastError(), "film_add_samples launch") ? NNBVH_OK : NNBVH_ERR_DEVICE;
}

static int film_move(nnbvh_film *f, const int32_t *d_index, int64_t n, void *d_buf, void *stream,
                     bool pack) {
    if (!f || n < 0 || (n > 0 && (!d_index || !d_buf))) {
        set_error("film_pack/unpack_pixels_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    FilmDeviceGuard guard(f->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    if (pack)
        hipLaunchKernelGGL(film_pack, dim3(film_grid(n)), dim3(kFilmBlock), 0, (hipStream_t)stream,
                           (const double2 *)f->d_pixels, d_index, (long)n, (double2 *)d_buf,
                           (long)(f->x1 - f->x0) * (f->y1 - f->y0));
    else
        hipLaunchKernelGGL(film_unpack, dim3(film_grid(n)), dim3(kFilmBlock), 0, (hipStream_t)stream,
                           (double2 *)f->d_pixels, d_index, (long)n, (const double2 *)d_buf,
                           (long)(f->x1 - f->x0) * (f->y1 - f->y0));
    return film_hip_ok(hipGetLastError(), "film pack/unpack launch") ? NNBVH_OK : NNBVH_ERR_DEVICE;
}

int nnbvh_film_pack_pixels_device(nnbvh_film *f, const int32_t *d_index, int64_t n, void *d_out,
                                  void *stream) {
    return film_move(f, d_index, n, d_out, stream, true);
}

int nnbvh_film_unpack_pixels_device(nnbvh_film *f, const int32_t *d_index, int64_t n, const void *d_in,
                                    void *stream) {
    return film_move(f, d_index, n, const_cast<void *>(d_in), stream, false);
}

}  // extern "C"
