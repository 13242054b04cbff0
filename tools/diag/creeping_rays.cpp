// creeping_rays.cpp -- which rays take the most look-ups, and why: trajectories (cell, box, step axis, nearest wall) of the slowest rays of a poses file (tools only;
// inputs from tools/model_inputs.py, the field from the cache tools/sweep_model.cpp leaves in /tmp):  g++ -O2 -std=c++17 -I. -DFTGP_SECTORS=16 tools/diag/creeping_rays.cpp -o /tmp/creep; /tmp/creep /tmp/track.raw /tmp/poses.bin 20 3
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <vector>
#include "../../include/ftgp.h"
#include "../../ft_grandprix_amd/csrc/ftgp_march.h"
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[3]; if (!f || fread(hdr, 4, 3, f) != 3) return 2;
    const int W = hdr[0], H = hdr[1], wpr = hdr[2];
    std::vector<uint32_t> bits((size_t)H * wpr); if (fread(bits.data(), 4, bits.size(), f) != bits.size()) return 2; fclose(f);
    f = fopen(argv[2], "rb"); double ph[6]; if (fread(ph, 8, 6, f) != 6) return 2;
    const int n_cars = (int)ph[0];
    std::vector<double> pose((size_t)n_cars * 4); if (fread(pose.data(), 8, pose.size(), f) != pose.size()) return 2; fclose(f);
    const int R = 1080;
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> field(cells * FTGP_SECTORS);
    char cache[256]; snprintf(cache, sizeof cache, "/tmp/sweep_model_field_%dx%d_%d.bin", W, H, FTGP_SECTORS);
    FILE* cf = fopen(cache, "rb"); if (!cf || fread(field.data(), 2, field.size(), cf) != field.size()) { fprintf(stderr, "no field cache\n"); return 2; }
    const int fstride = W + 2; const uint32_t plane256 = ftgp_plane256(W, H);
    const float eps = ldexpf(1.0f, -11), isx = (float)(1.0 / ph[1]), isy = (float)(1.0 / ph[2]), r0 = 0.03f;
    auto wall = [&](int x, int y) { return x < 0 || y < 0 || x >= W || y >= H ? 2 : (int)((bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u); };
    int shown = 0; const int want = argc > 3 ? atoi(argv[3]) : 12, nshow = argc > 4 ? atoi(argv[4]) : 3;
    for (int c = 0; c < n_cars && shown < nshow; ++c) {
        const double* p = &pose[(size_t)c * 4];
        const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
        const double lcx = p[0] + ch * -0.0525, lcy = p[1] + sh * -0.0525;
        const float u0 = (float)((lcx - ph[3]) / ph[1]), v0 = (float)((ph[4] - lcy) / ph[2]);
        for (int j = 0; j < R && shown < nshow; ++j) {
            const double phi = ((360.0 / R) * j - 90.0) * (M_PI / 180.0);
            const float bx = (float)sin(phi), by = (float)(-cos(phi));
            const float dxw = fmaf((float)ch, bx, -((float)sh * by)), dyw = fmaf((float)sh, bx, (float)ch * by);
            const float du = dxw * isx, dv = -(dyw * isy);
            FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
            std::vector<std::array<int, 6>> tr;
            int n = 1;
            for (; n < 1000; ++n) {
                const uint32_t wq = field[ftgp_ray_offset(r) >> 1];
                FtgpStep st; const bool near = ftgp_ray_step(r, wq, eps, st);
                { const int x0 = (int)floorf(fmaf(du, -r0, u0)), y0 = (int)floorf(fmaf(dv, -r0, v0));      // relative cell -> image cell
                  tr.push_back({ du < 0 ? x0 - r.mx : x0 + r.mx, dv < 0 ? y0 - r.my : y0 + r.my, (int)(wq & 255), (int)(wq >> 8), st.stepx, 0 }); }
                ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                if (!st.live) break;
            }
            if (n >= want) {
                ++shown;
                const uint32_t sec = ftgp_ray_sector(du, dv, ftgp_iv(du), ftgp_iv(dv));
                printf("car %d ray %d: %d iterations, du %.3f dv %.3f slope %.3f sector %u (mirror %u%u dom %u slice %u) range %.2f px\n", c, j, n, du, dv, fminf(fabsf(du), fabsf(dv)) / fmaxf(fabsf(du), fabsf(dv)), sec, sec & 1, (sec >> 1) & 1, (sec >> 2) & 1, sec >> 3, fabsf(r.s) * 40);
                for (auto& t : tr) {
                    const int x = t[0], y = t[1];
                    printf("   cell (%d,%d) box %dx%d step%c", x, y, t[2], t[3], t[4] ? 'x' : 'y');
                    // nearest wall within 6 cells, for orientation
                    int best = 99, bxw = 0, byw = 0; for (int dy = -6; dy <= 6; ++dy) for (int dx = -6; dx <= 6; ++dx) if (wall(x + dx, y + dy) == 1 && std::max(abs(dx), abs(dy)) < best) { best = std::max(abs(dx), abs(dy)); bxw = dx; byw = dy; }
                    if (best < 99) printf("  nearest wall at (%+d,%+d)", bxw, byw);
                    printf("\n");
                }
            }
        }
    }
    return 0;
}
