// Per-car cost of one LiDAR sweep (tools only): total march iterations of the car's n_rays rays with the SHIPPED march
// (ftgp_march.h), for every pose of a poses file.  Formats as tools/sweep_model.cpp.
//   build: g++ -O2 -std=c++17 -I. tools/car_cost.cpp -o /tmp/car_cost      run: /tmp/car_cost track.raw poses.bin n_rays [counts.u8] > costs.txt
//   counts.u8: n_cars x n_rays bytes, the iteration count of every ray (tools/order_probe.py sorts the pool by them)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../include/ftgp.h"
#include "../ft_grandprix_amd/csrc/ftgp_march.h"

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[3]; if (!f || fread(hdr, 4, 3, f) != 3) return 2;
    const int W = hdr[0], H = hdr[1], wpr = hdr[2];
    std::vector<uint32_t> bits((size_t)H * wpr); if (fread(bits.data(), 4, bits.size(), f) != bits.size()) return 2; fclose(f);
    f = fopen(argv[2], "rb"); if (!f) return 2;
    double ph[6]; if (fread(ph, 8, 6, f) != 6) return 2;
    const int n_cars = (int)ph[0];
    std::vector<double> pose((size_t)n_cars * 4); if (fread(pose.data(), 8, pose.size(), f) != pose.size()) return 2; fclose(f);
    const int R = atoi(argv[3]);
    std::vector<uint8_t> wall((size_t)W * H, 0);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) if ((bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u) wall[(size_t)y * W + x] = 1;
    const size_t plane = (size_t)W * H;
    std::vector<uint16_t> runx(2 * plane, 0), runy(2 * plane, 0);
    for (int y = 0; y < H; ++y) {
        int r = 65535; for (int x = W - 1; x >= 0; --x) { r = wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); runx[(size_t)y * W + x] = (uint16_t)r; }
        r = 65535; for (int x = 0; x < W; ++x) { r = wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); runx[plane + (size_t)y * W + x] = (uint16_t)r; }
    }
    for (int x = 0; x < W; ++x) {
        int r = 65535; for (int y = H - 1; y >= 0; --y) { r = wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); runy[(size_t)y * W + x] = (uint16_t)r; }
        r = 65535; for (int y = 0; y < H; ++y) { r = wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); runy[plane + (size_t)y * W + x] = (uint16_t)r; }
    }
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> field(cells * FTGP_SECTORS, (uint16_t)FTGP_FIELD_OUT);
    std::vector<uint8_t> have(cells * FTGP_SECTORS, 0);          // entries are computed on first use: only the corridor is ever touched
    const int fstride = W + 2; const uint32_t plane256 = ftgp_plane256(W, H);
    const float eps = ftgp_snap_eps(W, H);
    std::vector<float> bx(R), by(R);
    for (int j = 0; j < R; ++j) { const double phi = ((360.0 / R) * j - 90.0) * (M_PI / 180.0); bx[j] = (float)sin(phi); by[j] = (float)(-cos(phi)); }
    const float isx = (float)(1.0 / ph[1]), isy = (float)(1.0 / ph[2]), r0 = 0.03f;
    FILE* fc = argc > 4 ? fopen(argv[4], "wb") : nullptr;
    std::vector<uint8_t> cnt(R);
    for (int c = 0; c < n_cars; ++c) {
        const double* p = &pose[(size_t)c * 4];
        const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
        const double lcx = p[0] + ch * -0.0525, lcy = p[1] + sh * -0.0525;
        const float u0 = (float)((lcx - ph[3]) * (1.0 / ph[1])), v0 = (float)((ph[4] - lcy) * (1.0 / ph[2]));
        const float chf = (float)ch, shf = (float)sh;
        long iters = 0; double sumr = 0;
        for (int j = 0; j < R; ++j) {
            const float dxw = fmaf(chf, bx[j], -(shf * by[j])), dyw = fmaf(shf, bx[j], chf * by[j]);
            const float du = dxw * isx, dv = -(dyw * isy);
            FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
            int mine = 0;
            for (int n = 0; n < 100000; ++n) {
                const size_t idx = (uint32_t)ftgp_ray_offset(r) >> 1;
                if (!have[idx]) {
                    const size_t sec = idx / cells, cc = idx % cells; const int X = (int)(cc % (size_t)fstride), Y = (int)(cc / (size_t)fstride);
                    if (X >= 1 && X <= W && Y >= 1 && Y <= H) field[idx] = (uint16_t)ftgp_box_entry(runx.data(), runy.data(), W, H, X - 1, Y - 1, (int)sec);
                    have[idx] = 1;
                }
                FtgpStep st; const bool near = ftgp_ray_step(r, field[idx], eps, st);
                ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                ++iters; ++mine;
                if (!st.live) break;
            }
            sumr += fabsf(r.s);
            cnt[j] = (uint8_t)(mine > 255 ? 255 : mine);
        }
        if (fc) fwrite(cnt.data(), 1, cnt.size(), fc);
        printf("%ld %.3f\n", iters, sumr);
    }
    if (fc) fclose(fc);
    return 0;
}
