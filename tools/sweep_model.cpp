// CPU model of the LiDAR sweep scheduling (tools only): runs the SHIPPED march (ftgp_march.h) for real poses and counts what
// the wave-level schedule of lidar_pool() costs -- wave-iterations, refills, fix branches -- per car-step, for a given
// workgroup shape and refill threshold.  Waves of a workgroup take turns round by round (the real order is timing
// dependent; the totals barely move).
//   build: g++ -O2 -std=c++17 -I. tools/sweep_model.cpp -o /tmp/sweep_model
//   run:   /tmp/sweep_model track.raw poses.bin n_rays cars_per_block waves_per_block refill [eps_log2]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../include/ftgp.h"
#include "../ft_grandprix_amd/csrc/ftgp_march.h"

struct Tables { std::vector<uint8_t> wall; std::vector<uint16_t> runx, runy; };
static void build(const std::vector<uint32_t>& bits, int W, int H, int wpr, Tables& g)
{
    g.wall.assign((size_t)W * H, 0);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) if ((bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u) g.wall[(size_t)y * W + x] = 1;
    const size_t plane = (size_t)W * H;
    g.runx.assign(2 * plane, 0); g.runy.assign(2 * plane, 0);
    for (int y = 0; y < H; ++y) {
        int r = 65535;
        for (int x = W - 1; x >= 0; --x) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runx[(size_t)y * W + x] = (uint16_t)r; }
        r = 65535;
        for (int x = 0; x < W; ++x) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runx[plane + (size_t)y * W + x] = (uint16_t)r; }
    }
    for (int x = 0; x < W; ++x) {
        int r = 65535;
        for (int y = H - 1; y >= 0; --y) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runy[(size_t)y * W + x] = (uint16_t)r; }
        r = 65535;
        for (int y = 0; y < H; ++y) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runy[plane + (size_t)y * W + x] = (uint16_t)r; }
    }
}

#ifndef BOX_ENTRY
#define BOX_ENTRY ftgp_box_entry
#endif

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[3]; if (!f || fread(hdr, 4, 3, f) != 3) return 2;
    const int W = hdr[0], H = hdr[1], wpr = hdr[2];
    std::vector<uint32_t> bits((size_t)H * wpr); if (fread(bits.data(), 4, bits.size(), f) != bits.size()) return 2; fclose(f);
    f = fopen(argv[2], "rb"); if (!f) return 2;
    double ph[6]; if (fread(ph, 8, 6, f) != 6) return 2;      // n_cars, px_size_x, px_size_y, origin_x, origin_y, reserved
    const int n_cars = (int)ph[0];
    std::vector<double> pose((size_t)n_cars * 4); if (fread(pose.data(), 8, pose.size(), f) != pose.size()) return 2; fclose(f);
    const int R = atoi(argv[3]), cpb = atoi(argv[4]), wpb = atoi(argv[5]), refill = atoi(argv[6]);
    const float eps = ldexpf(1.0f, -(argc > 7 ? atoi(argv[7]) : 9));
    Tables g; build(bits, W, H, wpr, g);
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> field(cells * FTGP_SECTORS, (uint16_t)FTGP_FIELD_OUT);
    {   // the field is cached between runs of the model (it only depends on the track and the sector count)
        char cache[256]; snprintf(cache, sizeof cache, "/tmp/sweep_model_field_%dx%d_%d.bin", W, H, FTGP_SECTORS);
        FILE* cf = fopen(cache, "rb");
        if (cf && fread(field.data(), 2, field.size(), cf) == field.size()) fclose(cf);
        else {
            #pragma omp parallel for collapse(2) schedule(dynamic, 8)
            for (int oct = 0; oct < FTGP_SECTORS; ++oct) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x)
                field[(size_t)oct * cells + (size_t)(y + 1) * (W + 2) + (x + 1)] = (uint16_t)BOX_ENTRY(g.runx.data(), g.runy.data(), W, H, x, y, oct);
            cf = fopen(cache, "wb"); if (cf) { fwrite(field.data(), 2, field.size(), cf); fclose(cf); }
        }
    }
    const int fstride = W + 2; const uint32_t plane256 = ftgp_plane256(W, H);
    std::vector<float> bx(R), by(R);
    for (int j = 0; j < R; ++j) { const double phi = ((360.0 / R) * j - 90.0) * (M_PI / 180.0); bx[j] = (float)sin(phi); by[j] = (float)(-cos(phi)); }
    const float isx = (float)(1.0 / ph[1]), isy = (float)(1.0 / ph[2]), r0 = 0.03f;
    long wave_iters = 0, refills = 0, fixes = 0, lane_iters = 0, rays = 0, max_wave_iters_sum = 0, lines = 0;
    // what-if SORTED=1: the pool hands a car's rays out by descending iteration count (as known from this very pose: an upper
    // bound for "sort by the previous step's counts"); SORTED=2: ascending
    const int sorted_mode = getenv("SORTED") ? atoi(getenv("SORTED")) : 0;
    std::vector<std::vector<int>> perm(n_cars);
    std::vector<long> mean_count(R, 0);
    if (sorted_mode) {
        for (int c = 0; c < n_cars; ++c) {
            std::vector<std::pair<int, int>> key(R);
            const double* p = &pose[(size_t)c * 4];
            const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
            const double lcx = p[0] + (ch * -0.0525 - sh * 0.0), lcy = p[1] + (sh * -0.0525 + ch * 0.0);
            const float u0 = (float)((lcx - ph[3]) * (1.0 / ph[1])), v0 = (float)((ph[4] - lcy) * (1.0 / ph[2]));
            const float chf = (float)ch, shf = (float)sh;
            for (int j = 0; j < R; ++j) {
                const float dxw = fmaf(chf, bx[j], -(shf * by[j])), dyw = fmaf(shf, bx[j], chf * by[j]);
                const float du = dxw * isx, dv = -(dyw * isy);
                FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
                int n = 0;
                for (; n < 100000; ++n) {
                    const uint32_t wq = field[ftgp_ray_offset(r) >> 1];
                    FtgpStep st; const bool near = ftgp_ray_step(r, wq, eps, st);
                    ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                    if (!st.live) break;
                }
                // SORTED=1 by count (descending), 2 ascending, 3 by range (descending, quantised to 1/8 world unit)
                const int rq = (int)(fabsf(r.s) * 8.0f);
                // SORTED=4: a static order, no measurement: rays along the car's axis (front and rear) first, sideways rays last
                const int al = (int)(1000.0 * fabs(cos(2.0 * M_PI * j / R)));
                // SORTED=5: static, by the mean iteration count of that ray index over all cars (filled below)
                key[j] = { sorted_mode == 1 ? -n : sorted_mode == 3 ? -rq : sorted_mode == 4 ? -al : n, j };
                if (sorted_mode == 5) { static std::vector<long> acc; if ((int)acc.size() != R) acc.assign(R, 0); acc[j] += n; key[j] = { 0, j }; if (c == n_cars - 1) { for (int q = 0; q < R; ++q) mean_count[q] = acc[q]; } }
            }
            perm[c].resize(R);
            const int BS = getenv("BLOCK") ? atoi(getenv("BLOCK")) : 1;
            if (BS > 1 && R % BS == 0) {          // blocks of BS consecutive rays ordered by their largest key (descending keys are negative: min)
                std::vector<std::pair<int, int>> bk(R / BS);
                for (int b = 0; b < R / BS; ++b) { int m = 1 << 30; for (int o = 0; o < BS; ++o) m = std::min(m, key[b * BS + o].first); bk[b] = { m, b }; }
                std::stable_sort(bk.begin(), bk.end());
                for (int k = 0; k < R / BS; ++k) for (int o = 0; o < BS; ++o) perm[c][k * BS + o] = bk[k].second * BS + o;
            } else {
                std::stable_sort(key.begin(), key.end());
                for (int j = 0; j < R; ++j) perm[c][j] = key[j].second;
            }
        }
    }
    if (sorted_mode == 5) {         // one order for all cars
        std::vector<std::pair<long, int>> k5(R); for (int j = 0; j < R; ++j) k5[j] = { -mean_count[j], j };
        std::stable_sort(k5.begin(), k5.end());
        for (int c = 0; c < n_cars; ++c) for (int j = 0; j < R; ++j) perm[c][j] = k5[j].second;
    }
    const int iso_mode = getenv("ISO") ? atoi(getenv("ISO")) : 0;
    long iso_used = 0, iso_near = 0;
    std::vector<double> pose_next;
    if (getenv("NEXT")) {          // the same cars one step later (tools/model_inputs.py): what-ifs that sort by the PREVIOUS step's counts
        FILE* fn = fopen(getenv("NEXT"), "rb"); double phn[6];
        if (fn && fread(phn, 8, 6, fn) == 6 && (int)phn[0] == n_cars) { pose_next.resize((size_t)n_cars * 4); if (fread(pose_next.data(), 8, pose_next.size(), fn) != pose_next.size()) pose_next.clear(); }
        if (fn) fclose(fn);
    }
    if (getenv("HIST")) {          // iterations per ray, and per group of 64 consecutive rays (= the slowest ray of the group): lidar_groups
        std::vector<long> hr(64, 0), hg(64, 0); long sum_g = 0, ng = 0, sum_r = 0;
        std::vector<long> by_index(R, 0);
        long alt_sum[4] = { 0, 0, 0, 0 }, nx_sum[5] = { 0, 0, 0, 0, 0 };
        for (int c = 0; c < n_cars; ++c) {
            const double* p = &pose[(size_t)c * 4];
            const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
            const double lcx = p[0] + (ch * -0.0525 - sh * 0.0), lcy = p[1] + (sh * -0.0525 + ch * 0.0);
            const float u0 = (float)((lcx - ph[3]) * (1.0 / ph[1])), v0 = (float)((ph[4] - lcy) * (1.0 / ph[2]));
            const float chf = (float)ch, shf = (float)sh;
            int gmax = 0;
            std::vector<int> cnt_of(R); std::vector<float> range_of(R);
            for (int j = 0; j < R; ++j) {
                const float dxw = fmaf(chf, bx[j], -(shf * by[j])), dyw = fmaf(shf, bx[j], chf * by[j]);
                const float du = dxw * isx, dv = -(dyw * isy);
                FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
                int n = 1;
                for (; n < 100000; ++n) {
                    const uint32_t wq = field[ftgp_ray_offset(r) >> 1];
                    FtgpStep st; const bool near = ftgp_ray_step(r, wq, eps, st);
                    ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                    if (!st.live) break;
                }
                cnt_of[j] = n; range_of[j] = fabsf(r.s);
                hr[std::min(n, 63)]++; sum_r += n; by_index[j] += n;
                gmax = std::max(gmax, n);
                if ((j & 63) == 63 || j == R - 1) { hg[std::min(gmax, 63)]++; sum_g += gmax; ++ng; gmax = 0; }
            }
            // what-if: groups of 64 formed from the car's rays in another order -- by the true count (the bound), by the range (what the
            // previous step's scan would offer), by the range quantised to 1/4 unit with the index as tie-break (keeps neighbours together)
            auto grouped = [&](auto key) {
                std::vector<int> idx(R); for (int j = 0; j < R; ++j) idx[j] = j;
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key(a) > key(b); });
                long sum = 0; int m = 0;
                for (int k = 0; k < R; ++k) { m = std::max(m, cnt_of[idx[k]]); if ((k & 63) == 63 || k == R - 1) { sum += m; m = 0; } }
                return sum;
            };
            alt_sum[0] += grouped([&](int j) { return (double)cnt_of[j]; });
            alt_sum[1] += grouped([&](int j) { return (double)range_of[j]; });
            alt_sum[2] += grouped([&](int j) { return floor((double)range_of[j] * 4.0); });
            alt_sum[3] += grouped([&](int j) { return floor((double)range_of[j] * 1.0); });
            if (!pose_next.empty()) {          // counts of THIS pose as the key, counts one step later as the cost
                std::vector<int> cnt_next(R);
                const double* q = &pose_next[(size_t)c * 4];
                const double ch2 = 1.0 - 2.0 * (q[3] * q[3]), sh2 = 2.0 * (q[2] * q[3]);
                const double lx2 = q[0] + (ch2 * -0.0525), ly2 = q[1] + (sh2 * -0.0525);
                const float u2 = (float)((lx2 - ph[3]) * (1.0 / ph[1])), v2 = (float)((ph[4] - ly2) * (1.0 / ph[2]));
                const float c2 = (float)ch2, s2 = (float)sh2;
                for (int j = 0; j < R; ++j) {
                    const float dxw = fmaf(c2, bx[j], -(s2 * by[j])), dyw = fmaf(s2, bx[j], c2 * by[j]);
                    const float du = dxw * isx, dv = -(dyw * isy);
                    FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u2), fmaf(dv, -r0, v2), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
                    int n = 1;
                    for (; n < 100000; ++n) {
                        const uint32_t wq = field[ftgp_ray_offset(r) >> 1];
                        FtgpStep st; const bool near = ftgp_ray_step(r, wq, eps, st);
                        ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                        if (!st.live) break;
                    }
                    cnt_next[j] = n;
                }
                auto cost_next = [&](const std::vector<int>& idx, int n_idx) {
                    long sum = 0; int m = 0;
                    for (int k = 0; k < n_idx; ++k) { m = std::max(m, cnt_next[idx[k]]); if ((k & 63) == 63 || k == n_idx - 1) { sum += m; m = 0; } }
                    return sum;
                };
                std::vector<int> idx(R); for (int j = 0; j < R; ++j) idx[j] = j;
                nx_sum[0] += cost_next(idx, R);                                                     // neighbours, next step (the baseline there)
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cnt_of[a] > cnt_of[b]; });
                nx_sum[1] += cost_next(idx, R);                                                     // sorted by the previous step's count
                std::vector<int> cl(R); for (int j = 0; j < R; ++j) cl[j] = std::min(cnt_of[j], 6);  // ... by its count clipped to 6 (3 bits per ray)
                for (int j = 0; j < R; ++j) idx[j] = j;
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cl[a] > cl[b]; });
                nx_sum[2] += cost_next(idx, R);
                // pairs kept: first-half rays sorted by max(count[j], count[j + R/2]); a group and its opposite are marched one after the other
                const int half = R / 2;
                std::vector<int> hi(half); for (int j = 0; j < half; ++j) hi[j] = j;
                std::stable_sort(hi.begin(), hi.end(), [&](int a, int b) { return std::max(cnt_of[a], cnt_of[a + half]) > std::max(cnt_of[b], cnt_of[b + half]); });
                std::vector<int> opp(half); for (int k = 0; k < half; ++k) opp[k] = hi[k] + half;
                nx_sum[3] += cost_next(hi, half) + cost_next(opp, half);
                for (int j = 0; j < half; ++j) hi[j] = j;
                for (int k = 0; k < half; ++k) opp[k] = k + half;
                nx_sum[4] += cost_next(hi, half) + cost_next(opp, half);                            // neighbours in pairs (what ships, ignoring the mixed group)
            }
        }
        printf("what-if, wave-iterations per car-step with groups of 64 after sorting a car's rays: by true count %.1f, by range %.1f, by range in 1/4-unit bins %.1f, in 1-unit bins %.1f\n",
               (double)alt_sum[0] / n_cars, (double)alt_sum[1] / n_cars, (double)alt_sum[2] / n_cars, (double)alt_sum[3] / n_cars);
        if (!pose_next.empty())
            printf("one step later: neighbours %.1f; sorted by the previous step's count %.1f, by that count clipped to 6: %.1f; pairs kept, sorted by the pair's larger count %.1f (neighbours in pairs %.1f)\n",
                   (double)nx_sum[0] / n_cars, (double)nx_sum[1] / n_cars, (double)nx_sum[2] / n_cars, (double)nx_sum[3] / n_cars, (double)nx_sum[4] / n_cars);
        printf("iterations per ray: mean %.2f; per group of 64: mean %.2f (x %.1f groups per car = %.1f wave-iterations per car-step)\n", (double)sum_r / ((double)n_cars * R), (double)sum_g / ng, (double)ng / n_cars, (double)sum_g / n_cars);
        printf("  n    rays%%  groups%%\n");
        for (int n = 1; n < 64; ++n) if (hr[n] || hg[n]) printf("%3d  %6.2f  %6.2f\n", n, 100.0 * hr[n] / ((double)n_cars * R), 100.0 * hg[n] / ng);
        printf("mean iterations by ray index (every 30th):"); for (int j = 0; j < R; j += 30) printf(" %d:%.1f", j, (double)by_index[j] / n_cars); printf("\n");
        return 0;
    }
    struct Lane { FtgpRay r; int g; bool done; };
    for (int c0 = 0; c0 < n_cars; c0 += cpb) {
        const int nc = std::min(cpb, n_cars - c0), total = nc * R;
        int pool = 0;
        std::vector<std::vector<Lane>> wv(wpb, std::vector<Lane>(64));
        std::vector<int> witers(wpb, 0); std::vector<char> empty(wpb, 0), fin(wpb, 0);
        for (auto& w : wv) for (auto& l : w) { ftgp_ray_park(l.r, -1.0f); l.g = -1; l.done = true; }
        int alive = wpb;
        while (alive > 0) {
            for (int w = 0; w < wpb; ++w) {
                if (fin[w]) continue;
                auto& L = wv[w];
                for (auto& l : L) if (l.done && l.g >= 0) { l.g = -1; ++rays; }
                if (!empty[w]) {
                    int nfree = 0; for (auto& l : L) nfree += l.done;
                    const int base = pool; pool += nfree; ++refills;
                    int rank = 0;
                    for (auto& l : L) if (l.done) {
                        const int mine = base + rank++;
                        if (mine < total) {
                            l.g = mine; const int c = mine / R, j = sorted_mode ? perm[c0 + c][mine % R] : mine % R;
                            const double* p = &pose[(size_t)(c0 + c) * 4];
                            const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
                            const double lcx = p[0] + (ch * -0.0525 - sh * 0.0), lcy = p[1] + (sh * -0.0525 + ch * 0.0);
                            const float u0 = (float)((lcx - ph[3]) * (1.0 / ph[1])), v0 = (float)((ph[4] - lcy) * (1.0 / ph[2]));
                            const float chf = (float)ch, shf = (float)sh;
                            const float dxw = fmaf(chf, bx[j], -(shf * by[j])), dyw = fmaf(shf, bx[j], chf * by[j]);
                            const float du = dxw * isx, dv = -(dyw * isy);
                            ftgp_ray_init(l.r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256);
                            l.done = false;
                            if (iso_mode) {          // what-if: a first jump without a lookup, by the clearance of the LiDAR centre (one value per car)
                                // D = distance (pixels) from the centre to the nearest wall pixel or image edge, margin sqrt(2) + slack
                                const int cx = (int)floorf(u0), cy = (int)floorf(v0);
                                double D = 1e9;
                                const int RAD = 80;
                                for (int yy = std::max(0, cy - RAD); yy <= std::min(H - 1, cy + RAD); ++yy) for (int xx = std::max(0, cx - RAD); xx <= std::min(W - 1, cx + RAD); ++xx)
                                    if (g.wall[(size_t)yy * W + xx]) { const double ddx = xx + 0.5 - u0, ddy = yy + 0.5 - v0; D = std::min(D, sqrt(ddx * ddx + ddy * ddy)); }
                                D = std::min(D, (double)RAD);
                                D = std::min(D, std::min(std::min((double)u0, (double)W - u0), std::min((double)v0, (double)H - v0)));
                                if (iso_mode == 2) D = floor(D);                    // as a u8 plane would hold it
                                const float t0 = (float)((D - 1.5) / std::max(isx, isy));
                                if (t0 > r0) {
                                    const float lx = fmaf(du, t0, u0), ly = fmaf(dv, t0, v0);
                                    const float fx = lx - floorf(lx), fy = ly - floorf(ly);
                                    if (fabsf(fx - 0.5f) <= 0.5f - eps && fabsf(fy - 0.5f) <= 0.5f - eps) {
                                        l.r.mx = abs((int)floorf(lx) - (int)floorf(fmaf(du, -r0, u0))); l.r.my = abs((int)floorf(ly) - (int)floorf(fmaf(dv, -r0, v0))); ++iso_used;
                                    } else ++iso_near;
                                }
                            }
                            if (getenv("PRESTEP")) {          // what-if: the start cell's entry is at hand (no wave-iteration for it)
                                const uint32_t wq = field[ftgp_ray_offset(l.r) >> 1];
                                FtgpStep st; const bool near = ftgp_ray_step(l.r, wq, eps, st);
                                ftgp_ray_commit(l.r, st, near ? ftgp_ray_fix(l.r, st) : st.t);
                            }
                        }
                    }
                    empty[w] = base + nfree >= total;
                }
                bool any = false; for (auto& l : L) any |= l.g >= 0;
                if (!any) { fin[w] = 1; --alive; continue; }
                const int want = empty[w] ? 64 : refill;
                for (int guard = 0; guard < 32768; ++guard) {
                    bool anynear = false; int ndone = 0;
                    std::vector<int> lineset;
                    for (auto& l : L) {
                        if (l.done) { ++ndone; continue; }          // a finished ray is not stepped any more (lidar_pool: `alive`)
                        const int off = ftgp_ray_offset(l.r);
                        lineset.push_back(off >> 7);
                        const uint32_t wq = field[off >> 1];
                        FtgpStep st; const bool near = ftgp_ray_step(l.r, wq, eps, st);
                        anynear |= near;
                        const int t = near ? ftgp_ray_fix(l.r, st) : st.t;
                        ftgp_ray_commit(l.r, st, t);
                        if (!l.done || l.g >= 0) lane_iters += !l.done;
                        l.done = (!st.live); ndone += (!st.live);
                    }
                    std::sort(lineset.begin(), lineset.end()); lines += std::unique(lineset.begin(), lineset.end()) - lineset.begin();
                    ++wave_iters; ++witers[w]; fixes += anynear;
                    if (ndone >= want) break;
                }
            }
        }
        max_wave_iters_sum += *std::max_element(witers.begin(), witers.end());
        if (getenv("DUMPWG")) { long tot = 0; for (int x : witers) tot += x; printf("WG %d sum-wave-iters %ld max-wave-iters %d\n", c0 / cpb, tot, *std::max_element(witers.begin(), witers.end())); }
    }
    const double nsteps = (double)n_cars;
    printf("R %d cpb %d wpb %d refill %d eps 2^-%d: per car-step: wave-iters %.1f  refills %.1f  fix-branches %.1f  useful lane-iters %.0f (util %.2f)  iters/ray %.2f  "
           "max wave-iters per WG %.1f  128B-lines per wave-iter %.1f\n",
           R, cpb, wpb, refill, (int)-log2f(eps), wave_iters / nsteps, refills / nsteps, fixes / nsteps, lane_iters / nsteps, lane_iters / (64.0 * wave_iters),
           (double)lane_iters / rays, (double)max_wave_iters_sum / ((n_cars + cpb - 1) / cpb), (double)lines / wave_iters);
    if (iso_mode) printf("ISO: %ld rays started past their origin cell, %ld fell back (landing near a boundary)\n", iso_used, iso_near);
    return 0;
}
