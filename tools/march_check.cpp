// Host harness: runs the SHIPPED march (ftgp_march.h), box search and build_tables() on the CPU against a plain DDA (the specification)
// over random rays.  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -std=c++17 -x hip tools/march_check.cpp -o /tmp/march_check
// Input: raw track dump written by tests (int32 W, H, wpr; then uint32 bits[H*wpr]).
#include "../ft_grandprix_amd/csrc/ftgp_api.hip"
#include <float.h>
#include <random>

static bool wall_at(const FtgpTrack& t, int x, int y) { return (t.bits[(size_t)y * t.words_per_row + (x >> 5)] >> (x & 31)) & 1u; }

// the specification, restated here on its own: crossing times relative to the start cell, one fma each
static float plain(const FtgpTrack& t, float pu, float pv, float du, float dv)
{
    const int W = t.width, H = t.height;
    const int ix0 = (int)floorf(pu), iy0 = (int)floorf(pv);
    if (ix0 < 0 || ix0 >= W || iy0 < 0 || iy0 >= H) return -1.0f;
    const float fu = pu - (float)ix0, fv = pv - (float)iy0;
    const float gu = du < 0.0f ? 1.0f - fu : fu, gv = dv < 0.0f ? 1.0f - fv : fv;
    float ivx = fabsf(1.0f / du), ivy = fabsf(1.0f / dv);
    if (!(ivx < FLT_MAX)) ivx = FLT_MAX;
    if (!(ivy < FLT_MAX)) ivy = FLT_MAX;
    const float cx = gu * ivx, cy = gv * ivy;
    const int sx = du < 0.0f ? -1 : 1, sy = dv < 0.0f ? -1 : 1;
    int mx = 0, my = 0;
    float s = 0.0f;
    for (;;) {
        const int ix = ix0 + sx * mx, iy = iy0 + sy * my;
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0f;
        if (wall_at(t, ix, iy)) return fabsf(s);
        const float sX = fmaf((float)(mx + 1), ivx, -cx), sY = fmaf((float)(my + 1), ivy, -cy);
        if (sX < sY) { s = sX; ++mx; } else { s = sY; ++my; }
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: march_check track.raw [n_rays] [seed] [scale] [coarse sectors]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 2; }
    int32_t hdr[3]; if (fread(hdr, 4, 3, f) != 3) return 2;
    std::vector<uint32_t> bits((size_t)hdr[1] * hdr[2]);
    if (fread(bits.data(), 4, bits.size(), f) != bits.size()) return 2;
    fclose(f);
    const long n = argc > 2 ? atol(argv[2]) : 2000000;
    const unsigned seed = argc > 3 ? (unsigned)atoi(argv[3]) : 1;
    const double scale = argc > 4 ? atof(argv[4]) : 40.0;     // pixels per world unit (1/px_size)
    FtgpTrack t{}; t.width = hdr[0]; t.height = hdr[1]; t.words_per_row = hdr[2]; t.bits = bits.data();
    // the shipped host tables + the shipped per-cell box search (ftgp_box_entry, the body of ftgp_box_field_kernel) on the CPU
    HostTables g; build_tables(t, 3, g);
    const int W = t.width, H = t.height;
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;
    std::vector<uint16_t> field(cells * FTGP_SECTORS, (uint16_t)FTGP_FIELD_OUT);
    long gw_bad = 0;
    #pragma omp parallel for collapse(2) schedule(dynamic, 16) reduction(+ : gw_bad)
    for (int oct = 0; oct < FTGP_SECTORS; ++oct)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const uint32_t e = ftgp_box_entry(g.runx.data(), g.runy.data(), W, H, x, y, oct);
                field[(size_t)oct * cells + (size_t)(y + 1) * (W + 2) + (x + 1)] = (uint16_t)e;
                if ((e == 0) != wall_at(t, x, y) || (e != 0 && (e & 255u) == 0)) ++gw_bad;      // 0 <=> wall; a free cell never carries kx = 0
            }
    printf("grid_wall: %ld mismatching pixels\n", gw_bad);
    // argv[5] = a coarse sector count (8, 16, 32): the field as ftgp_create lays it out for large batches -- `coarse` planes, reached through the
    // sector table that maps a ray's sector (always found among FTGP_SECTORS) to its plane
    const int coarse = argc > 5 ? atoi(argv[5]) : 0;
    int32_t tab[FTGP_SECTORS][4];
    if (coarse) {
        const int planes = ftgp_sector_table(tab, coarse, W + 2, ftgp_plane256(W, H));
        field.assign(cells * (size_t)planes, (uint16_t)FTGP_FIELD_OUT);
        #pragma omp parallel for collapse(2) schedule(dynamic, 16)
        for (int oct = 0; oct < coarse; ++oct)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    field[(size_t)oct * cells + (size_t)(y + 1) * (W + 2) + (x + 1)] = (uint16_t)ftgp_box_entry(g.runx.data(), g.runy.data(), W, H, x, y, oct, coarse / 8);
        printf("coarse planes: %d\n", coarse);
    }
    const float eps = ftgp_snap_eps(W, H);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> ux(0, t.width), uy(0, t.height), ua(0, 2 * M_PI), u01(0, 1);
    long bad = 0, hits = 0;
    for (long i = 0; i < n; ++i) {
        double x = ux(rng), y = uy(rng), a = ua(rng);
        const int kind = (int)(i % 8);
        if (kind == 1) { x = floor(x); y = floor(y); }                         // rays from pixel corners
        if (kind == 2) { a = (M_PI / 4) * (double)(int)(u01(rng) * 8); }        // axis-aligned / diagonal directions
        if (kind == 3) { x = floor(x); y = floor(y); a = (M_PI / 4) * (double)(int)(u01(rng) * 8); }
        if (kind == 4) { x = floor(x) + 0.5; a = (M_PI / 2) * (double)(int)(u01(rng) * 4); }
        float pu = (float)x, pv = (float)y;
        float du = (float)(cos(a) * scale), dv = (float)(sin(a) * scale);
        if (kind == 3 || kind == 4) { if (fabsf(du) < 1e-3f) du = 0.0f; if (fabsf(dv) < 1e-3f) dv = 0.0f; }
        if (kind == 2 && (i & 8)) { du = (float)(int)(du); dv = (float)(int)dv; if (du == 0 && dv == 0) du = 1; }
        if ((kind == 3 || kind == 4) && (i & 16)) { if (du == 0.0f) du = -0.0f; if (dv == 0.0f) dv = -0.0f; }      // a rotation's products give -0 as readily as +0
        const float a_ = ftgp_march_one(field.data(), W, H, eps, pu, pv, du, dv, coarse ? &tab[0][0] : nullptr), b_ = plain(t, pu, pv, du, dv);
        hits += b_ >= 0;
        if (memcmp(&a_, &b_, 4) != 0) {
            if (bad < 10) printf("MISMATCH kind %d pu %.9g pv %.9g du %.9g dv %.9g : grid %.9g plain %.9g\n", kind, pu, pv, du, dv, a_, b_);
            ++bad;
        }
    }
    printf("%ld rays, %ld hits, %ld mismatches\n", n, hits, bad);
    return bad ? 1 : 0;
}
