// late_fine_planes.cpp -- what-if: a ray looks its first K cells up in the coarse field (16 sectors) and the later ones in the fine one (64) (tools only; field caches of
// tools/sweep_model.cpp):  g++ -O2 -std=c++17 -I. tools/diag/late_fine_planes.cpp -o /tmp/sw; /tmp/sw /tmp/track.raw /tmp/poses.bin 16
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/ftgp.h"
#include "../../ft_grandprix_amd/csrc/ftgp_march.h"
static std::vector<uint16_t> load(int W, int H, int ns) {
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> f(cells * ns);
    char cache[256]; snprintf(cache, sizeof cache, "/tmp/sweep_model_field_%dx%d_%d.bin", W, H, ns);
    FILE* cf = fopen(cache, "rb"); if (!cf || fread(f.data(), 2, f.size(), cf) != f.size()) { fprintf(stderr, "no field cache %d\n", ns); exit(2); }
    fclose(cf); return f;
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[3]; if (!f || fread(hdr, 4, 3, f) != 3) return 2;
    const int W = hdr[0], H = hdr[1];
    fclose(f);
    f = fopen(argv[2], "rb"); double ph[6]; if (fread(ph, 8, 6, f) != 6) return 2;
    const int n_cars = (int)ph[0];
    std::vector<double> pose((size_t)n_cars * 4); if (fread(pose.data(), 8, pose.size(), f) != pose.size()) return 2; fclose(f);
    const int R = 1080, COARSE = argc > 3 ? atoi(argv[3]) : 16;
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> fc = load(W, H, COARSE), ff = load(W, H, 64);
    int shift = 0; while ((64 >> shift) > COARSE) ++shift;
    const int fstride = W + 2; const uint32_t plane256 = ftgp_plane256(W, H);
    const float eps = ldexpf(1.0f, -11), isx = (float)(1.0 / ph[1]), isy = (float)(1.0 / ph[2]), r0 = 0.03f;
    for (int K : { 1000, 2, 3, 4, 5, 6, 8 }) {
        long sum_g = 0, ng = 0, fine = 0, total = 0, sum_cap = 0;
        for (int c = 0; c < n_cars; ++c) {
            const double* p = &pose[(size_t)c * 4];
            const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
            const double lcx = p[0] + ch * -0.0525, lcy = p[1] + sh * -0.0525;
            const float u0 = (float)((lcx - ph[3]) / ph[1]), v0 = (float)((ph[4] - lcy) / ph[2]);
            int gmax = 0;
            for (int j = 0; j < R; ++j) {
                const double phi = ((360.0 / R) * j - 90.0) * (M_PI / 180.0);
                const float bx = (float)sin(phi), by = (float)(-cos(phi));
                const float dxw = fmaf((float)ch, bx, -((float)sh * by)), dyw = fmaf((float)sh, bx, (float)ch * by);
                const float du = dxw * isx, dv = -(dyw * isy);
                FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
                const uint32_t s64 = ftgp_ray_sector(du, dv, ftgp_iv(du), ftgp_iv(dv));
                const uint32_t sc = ((s64 >> 3) >> shift) << 3 | (s64 & 7u);
                // offsets relative to the plane: r.offC is for plane s64 of the fine field
                const int rel = r.base - (int)((s64 * plane256) << 8);
                int n = 1;
                for (; n < 100000; ++n) {
                    const int off = r.mx * r.ax + r.my * r.ay + rel;
                    const bool usefine = n > K;
                    const uint32_t wq = usefine ? ff[(size_t)s64 * cells + (off >> 1)] : fc[(size_t)sc * cells + (off >> 1)];
                    fine += usefine; ++total;
                    FtgpStep st; const bool near = ftgp_ray_step(r, wq, eps, st);
                    ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                    if (!st.live) break;
                }
                gmax = std::max(gmax, n);
                if ((j & 63) == 63 || j == R - 1) { sum_g += gmax; sum_cap += std::min(gmax, 8); ++ng; gmax = 0; }
            }
        }
        printf("coarse %d, fine after %4d look-ups: %.1f wave-iterations per car-step (capped at 8 per group: %.1f), %.2f %% of the look-ups in the fine field, %.2f look-ups per ray\n",
               COARSE, K, (double)sum_g / n_cars, (double)sum_cap / n_cars, 100.0 * fine / total, (double)total / ((double)n_cars * R));
    }
    return 0;
}
