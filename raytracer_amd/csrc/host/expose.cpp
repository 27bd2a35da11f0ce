// expose(): bloom + 8-bit pack on the host, in double, operation for operation as the reference does it
// (/root/reference/src/main.cpp:186-242, /root/reference/src/geometry.cpp:15-23):
//   kernel[ky][kx] = exp(-(x*x + y*y)/18)/18/pi, radius 20;
//   B0 = exposure where colour*exposure > 1 else 0;  Bi = kernel (*) Bi-1 (taps outside the frame skipped,
//   ky outer / kx inner, one rounded product and one rounded sum per tap);  colour += B_num_blurs;
//   pack int(255*c/(1/exposure)) truncated, clamped to 255 from above only, r<<16 + g<<8 + b.
// The reference parallelises the convolution over rows with OpenMP (main.cpp:209); rows are independent,
// so std::threads over rows give the same bits.
#include <cmath>
#include <thread>

#include "rt_host.hpp"

namespace rt {

void expose(int* img, double* colors, int width, int height, double exposure, int num_blurs, int threads) {
    const int radius = 20;
    const double stddev = 3;
    const int K = 2 * radius + 1;
    std::vector<double> kernel((size_t)K * K);
    const double denom = 2 * stddev * stddev;
    for (int y = -radius; y <= radius; y++)
        for (int x = -radius; x <= radius; x++)
            kernel[(size_t)(y + radius) * K + x + radius] = std::exp(-(x * x + y * y) / denom) / denom / M_PI;
    const size_t n = (size_t)width * height;
    std::vector<double> cur(n * 3), nxt;
    for (size_t i = 0; i < n * 3; i++) cur[i] = (colors[i] * exposure > 1) ? exposure : 0.0;
    if (threads < 1) threads = 1;
    for (int b = 0; b < num_blurs; b++) {
        nxt.assign(n * 3, 0.0);
        auto rows = [&](int t) {
            for (int y = t; y < height; y += threads)
                for (int x = 0; x < width; x++) {
                    double c[3] = {0, 0, 0};
                    for (int ky = -radius; ky <= radius; ky++)
                        for (int kx = -radius; kx <= radius; kx++)
                            if (y + ky >= 0 && y + ky < height && x + kx >= 0 && x + kx < width) {
                                const double* s = &cur[((size_t)(y + ky) * width + (x + kx)) * 3];
                                const double k = kernel[(size_t)(ky + radius) * K + kx + radius];
                                for (int i = 0; i < 3; i++) c[i] = c[i] + s[i] * k;
                            }
                    for (int i = 0; i < 3; i++) nxt[((size_t)y * width + x) * 3 + i] = c[i];
                }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(rows, t);
        rows(0);
        for (auto& x : th) x.join();
        cur.swap(nxt);
    }
    for (size_t i = 0; i < n * 3; i++) colors[i] = colors[i] + cur[i];
    const double mx = 1.0 / exposure;
    for (size_t i = 0; i < n; i++) {
        int r = 255 * colors[3 * i] / mx, g = 255 * colors[3 * i + 1] / mx, bl = 255 * colors[3 * i + 2] / mx;
        if (r > 255) r = 255;
        if (g > 255) g = 255;
        if (bl > 255) bl = 255;
        img[i] = (r << 16) + (g << 8) + bl;
    }
}

}  // namespace rt

// plain-C entry for language bindings and tests
extern "C" void rt_host_expose(int* img, double* colors, int width, int height, double exposure, int num_blurs, int threads) {
    rt::expose(img, colors, width, height, exposure, num_blurs, threads);
}
