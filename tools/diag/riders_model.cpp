// riders_model.cpp -- what-if (tools only): the group sweep with an iteration cap per group and RIDERS.  A ray that is still marching when its
// group reaches the cap keeps its lane and rides along with the groups the wave draws next; the rays those lanes would have taken are
// DISPLACED to a queue (un-started) and marched later in gather groups of 64.  Counts what the schedule costs per car-step: wave-iterations
// (look-ups a wave issues, each one a dependent memory round trip), groups set up, the longest chain of a workgroup's waves.
// Field cache of tools/sweep_model.cpp (16 sectors):
//   g++ -O2 -std=c++17 -I. tools/diag/riders_model.cpp -o /tmp/riders; [LPT=1] [SETUP=2.0] /tmp/riders /tmp/track.raw /tmp/poses.bin [cars_per_wg] [waves]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/ftgp.h"
#include "../../ft_grandprix_amd/csrc/ftgp_march.h"

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[3]; if (!f || fread(hdr, 4, 3, f) != 3) return 2;
    const int W = hdr[0], H = hdr[1];
    fclose(f);
    f = fopen(argv[2], "rb"); double ph[6]; if (!f || fread(ph, 8, 6, f) != 6) return 2;
    const int n_cars = (int)ph[0];
    std::vector<double> pose((size_t)n_cars * 4); if (fread(pose.data(), 8, pose.size(), f) != pose.size()) return 2; fclose(f);
    const int R = 1080, half = R / 2, NS = 16, cpb = argc > 3 ? atoi(argv[3]) : 8, wpb = argc > 4 ? atoi(argv[4]) : 16;
    const double SETUP = getenv("SETUP") ? atof(getenv("SETUP")) : 2.0;      // a group's set-up + delivery in units of one march iteration's latency
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> field(cells * NS);
    {
        char cache[256]; snprintf(cache, sizeof cache, "/tmp/sweep_model_field_%dx%d_%d.bin", W, H, NS);
        FILE* cf = fopen(cache, "rb"); if (!cf || fread(field.data(), 2, field.size(), cf) != field.size()) { fprintf(stderr, "no field cache %s (run tools/sweep_model.cpp built with -DFTGP_SECTORS=16 first)\n", cache); return 2; }
        fclose(cf);
    }
    int32_t tab[64][4]; ftgp_sector_table(tab, NS, W + 2, ftgp_plane256(W, H));
    const int fstride = W + 2; const uint32_t plane256 = ftgp_plane256(W, H);
    const float eps = ftgp_snap_eps(W, H), isx = (float)(1.0 / ph[1]), isy = (float)(1.0 / ph[2]), r0 = 0.03f;
    // look-ups per ray (the terminal one included)
    std::vector<std::vector<int>> cnt(n_cars, std::vector<int>(R));
    for (int c = 0; c < n_cars; ++c) {
        const double* p = &pose[(size_t)c * 4];
        const double ch = 1.0 - 2.0 * (p[3] * p[3]), sh = 2.0 * (p[2] * p[3]);
        const double lcx = p[0] + ch * -0.0525, lcy = p[1] + sh * -0.0525;
        const float u0 = (float)((lcx - ph[3]) / ph[1]), v0 = (float)((ph[4] - lcy) / ph[2]);
        for (int j = 0; j < R; ++j) {
            const double phi = ((360.0 / R) * j - 90.0) * (M_PI / 180.0);
            const float bx = (float)sin(phi), by = (float)(-cos(phi));
            const float dxw = fmaf((float)ch, bx, -((float)sh * by)), dyw = fmaf((float)sh, bx, (float)ch * by);
            const float du = dxw * isx, dv = -(dyw * isy);
            FtgpRay r; ftgp_ray_init(r, fmaf(du, -r0, u0), fmaf(dv, -r0, v0), du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256, false, &tab[0][0]);
            int n = 1;
            for (; n < 100000; ++n) {
                const uint32_t wq = field[(uint32_t)ftgp_ray_offset(r) >> 1];
                FtgpStep st; const bool near = ftgp_ray_step(r, wq, eps, st);
                ftgp_ray_commit(r, st, near ? ftgp_ray_fix(r, st) : st.t);
                if (!st.live) break;
            }
            cnt[c][j] = n;
        }
    }
    // the task list of ftgp_create: pairs of opposite groups, long first (|cos| of the middle ray against the car's axis), the two cheapest pairs as singles
    struct Task { int j0, kind; };
    std::vector<Task> order;
    {
        std::vector<std::pair<double, Task>> key;
        int j0 = 0;
        for (; j0 + 64 <= half; j0 += 64) key.push_back({ 0, { j0, 1 } });
        if (j0 < half) key.push_back({ 0, { j0, half - j0 <= 32 ? 2 : 1 } });
        for (auto& k : key) k.first = -fabs(cos(2.0 * M_PI * std::min((double)R - 1.0, k.second.j0 + 31.5) / R));
        std::stable_sort(key.begin(), key.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (size_t k = 0; k < key.size(); ++k) {
            const Task t = key[k].second;
            if (t.kind == 1 && (int)(key.size() - k) <= 2) { order.push_back({ t.j0, 0 }); order.push_back({ t.j0 + half, 0 }); }
            else order.push_back(t);
        }
    }
    for (int CAP : { 1000, 10, 8, 7, 6, 5, 4 }) {
        double wave_iters = 0, groups = 0, gather_groups = 0, displaced = 0, chain_sum = 0, chain_mean_sum = 0, lane_iters = 0, rider_groups = 0;
        int n_wg = 0;
        for (int c0 = 0; c0 + cpb <= n_cars; c0 += cpb, ++n_wg) {
            struct Wave { double clock = 0; int rider[64]; int nr = 0; };
            std::vector<Wave> wv(wpb);
            for (auto& w : wv) for (int l = 0; l < 64; ++l) w.rider[l] = 0;      // remaining look-ups of the ray that rides in lane l (0: lane free)
            std::vector<int> queue;                                            // displaced rays (their look-up counts)
            // one pass of a group: the lanes' new rays (counts; 0 = no ray) beside the riders
            auto pass = [&](Wave& w, const int* fresh) {
                int live[64]; int mx = 0; bool had_riders = false;
                for (int l = 0; l < 64; ++l) {
                    if (w.rider[l]) { had_riders = true; live[l] = w.rider[l]; if (fresh[l]) { queue.push_back(fresh[l]); displaced += 1; } }
                    else live[l] = fresh[l];
                    mx = std::max(mx, live[l]);
                }
                const int it = std::min(mx, CAP);
                for (int l = 0; l < 64; ++l) { lane_iters += std::min(live[l], it); w.rider[l] = std::max(0, live[l] - it); }
                wave_iters += it; groups += 1; rider_groups += had_riders;
                w.clock += it + SETUP;
            };
            const int ntasks = (int)order.size() * cpb;
            // what-if LPT=1: the workgroup's tasks drawn by their TRUE cost (this very pose's look-up counts: the bound for "by the previous step's counts"), longest first
            std::vector<int> draw(ntasks); for (int g = 0; g < ntasks; ++g) draw[g] = g;
            if (getenv("LPT")) {
                auto cost = [&](int g) {
                    const Task t = order[g / cpb]; const int c = c0 + g % cpb; double sum = 0;
                    const int lim = t.kind == 1 ? half : R;
                    if (t.kind == 2) { int m = 0; for (int l = 0; l < 64; ++l) if ((l & 31) < half - t.j0) m = std::max(m, cnt[c][t.j0 + (l & 31) + (l >= 32 ? half : 0)]); return m + SETUP; }
                    for (int p = 0; p < (t.kind == 1 ? 2 : 1); ++p) { int m = 0; for (int l = 0; l < 64 && t.j0 + l < lim; ++l) m = std::max(m, cnt[c][t.j0 + l + p * half]); sum += m + SETUP; }
                    return sum;
                };
                std::vector<double> cs(ntasks); for (int g = 0; g < ntasks; ++g) cs[g] = cost(g);
                std::stable_sort(draw.begin(), draw.end(), [&](int a, int b) { return cs[a] > cs[b]; });
            }
            for (int gi = 0; gi < ntasks; ++gi) {
                const int g = draw[gi];
                Wave& w = *std::min_element(wv.begin(), wv.end(), [](const Wave& a, const Wave& b) { return a.clock < b.clock; });
                const Task t = order[g / cpb]; const int c = c0 + g % cpb;
                int fresh[64];
                if (t.kind == 2) {
                    for (int l = 0; l < 64; ++l) { const int j = t.j0 + (l & 31) + (l >= 32 ? half : 0); fresh[l] = (l & 31) < half - t.j0 ? cnt[c][j] : 0; }
                    pass(w, fresh);
                } else {
                    const int lim = t.kind == 1 ? half : R;
                    for (int l = 0; l < 64; ++l) fresh[l] = t.j0 + l < lim ? cnt[c][t.j0 + l] : 0;
                    pass(w, fresh);
                    if (t.kind == 1) { for (int l = 0; l < 64; ++l) fresh[l] = t.j0 + l < lim ? cnt[c][t.j0 + l + half] : 0; pass(w, fresh); }
                }
            }
            // the list is exhausted: every wave finishes its riders; then the displaced rays in gather groups (drawn by the wave that is free first)
            for (auto& w : wv) { int mx = 0; for (int l = 0; l < 64; ++l) { mx = std::max(mx, w.rider[l]); lane_iters += w.rider[l]; w.rider[l] = 0; } wave_iters += mx; w.clock += mx; }
            for (size_t q = 0; q < queue.size(); q += 64) {
                Wave& w = *std::min_element(wv.begin(), wv.end(), [](const Wave& a, const Wave& b) { return a.clock < b.clock; });
                int mx = 0; for (size_t k = q; k < std::min(queue.size(), q + 64); ++k) { mx = std::max(mx, queue[k]); lane_iters += queue[k]; }
                wave_iters += mx; gather_groups += 1; groups += 1; w.clock += mx + SETUP + 1.0;
            }
            double cmax = 0, csum = 0; for (auto& w : wv) { cmax = std::max(cmax, w.clock); csum += w.clock; }
            chain_sum += cmax; chain_mean_sum += csum / wpb;
        }
        const double ncs = (double)n_wg * cpb;
        printf("cap %4d: wave-iterations per car-step %6.1f  groups %5.2f (of them gather %4.2f, with riders aboard %4.2f)  displaced rays %5.1f  lanes busy %.2f | per workgroup-step: longest wave chain %6.1f, mean %6.1f (iterations + %.1f per group)\n",
               CAP, wave_iters / ncs, groups / ncs, gather_groups / ncs, rider_groups / ncs, displaced / ncs, lane_iters / (64.0 * wave_iters), chain_sum / n_wg, chain_mean_sum / n_wg, SETUP);
    }
    return 0;
}
