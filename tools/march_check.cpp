// Host harness: runs the SHIPPED march_grid() and build_grid() on the CPU against a plain DDA (the specification)
// over random rays.  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -std=c++17 -x hip tools/march_check.cpp -o /tmp/march_check
// Input: raw track dump written by tests (int32 W, H, wpr; then uint32 bits[H*wpr]).
#include "../ft_grandprix_amd/csrc/ftgp_api.hip"
#include <random>

static bool wall_at(const FtgpTrack& t, int x, int y) { return (t.bits[(size_t)y * t.words_per_row + (x >> 5)] >> (x & 31)) & 1u; }

static float plain(const FtgpTrack& t, float pu, float pv, float du, float dv)
{
    const int W = t.width, H = t.height;
    int ix = (int)floorf(pu), iy = (int)floorf(pv);
    if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0f;
    const float inv_du = (du != 0.0f) ? 1.0f / du : 0.0f, inv_dv = (dv != 0.0f) ? 1.0f / dv : 0.0f;
    float s = 0.0f;
    for (;;) {
        if (wall_at(t, ix, iy)) return fabsf(s);
        float sX = (du != 0.0f) ? ((float)((du > 0.0f) ? ix + 1 : ix) - pu) * inv_du : INFINITY;
        float sY = (dv != 0.0f) ? ((float)((dv > 0.0f) ? iy + 1 : iy) - pv) * inv_dv : INFINITY;
        if (sX < sY) { s = sX; ix += (du > 0.0f) ? 1 : -1; } else { s = sY; iy += (dv > 0.0f) ? 1 : -1; }
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0f;
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: march_check track.raw [n_rays] [seed] [scale]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 2; }
    int32_t hdr[3]; if (fread(hdr, 4, 3, f) != 3) return 2;
    std::vector<uint32_t> bits((size_t)hdr[1] * hdr[2]);
    if (fread(bits.data(), 4, bits.size(), f) != bits.size()) return 2;
    fclose(f);
    const long n = argc > 2 ? atol(argv[2]) : 2000000;
    const unsigned seed = argc > 3 ? (unsigned)atoi(argv[3]) : 1;
    const double scale = argc > 4 ? atof(argv[4]) : 40.0;     // pixels per world unit (1/px_size)
    FtgpTrack t{}; t.width = hdr[0]; t.height = hdr[1]; t.words_per_row = hdr[2]; t.bits = bits.data();
    HostGrid g; build_grid(t, g);
    DeviceParams P{}; P.width = t.width; P.height = t.height; P.nbx = g.nbx; P.nby = g.nby; P.nwpr = g.nwpr; P.n_fine = g.n_fine;
    P.snap_eps = 1.0f / 512.0f;
    P.n_rays = 8; P.eighth = 1; P.scan_floats = 8; P.ray_floats = 8;
    lds_layout(P, 1);
    std::vector<unsigned char> img((size_t)P.lds_bytes, 0);
    memcpy(img.data() + P.off_fine, g.fine.data(), g.fine.size());
    memcpy(img.data() + P.off_rank, g.rank.data(), g.rank.size() * sizeof(uint2));
    memcpy(img.data() + P.off_coarse, g.coarse.data(), g.coarse.size());
    LdsView L{}; L.fine = img.data() + P.off_fine; L.rank = reinterpret_cast<const uint2*>(img.data() + P.off_rank); L.coarse = img.data() + P.off_coarse;
    long gw_bad = 0;
    for (int y = 0; y < t.height; ++y)
        for (int x = 0; x < t.width; ++x)
            if (grid_wall(P, L, x, y) != wall_at(t, x, y)) { if (gw_bad < 5) printf("grid_wall mismatch at %d %d: %d vs %d\n", x, y, (int)grid_wall(P, L, x, y), (int)wall_at(t, x, y)); ++gw_bad; }
    printf("grid_wall: %ld mismatching pixels\n", gw_bad);
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
        const float a_ = march_grid(P, img.data(), pu, pv, du, dv), b_ = plain(t, pu, pv, du, dv);
        hits += b_ >= 0;
        if (memcmp(&a_, &b_, 4) != 0) {
            if (bad < 10) printf("MISMATCH kind %d pu %.9g pv %.9g du %.9g dv %.9g : grid %.9g plain %.9g\n", kind, pu, pv, du, dv, a_, b_);
            ++bad;
        }
    }
    printf("%ld rays, %ld hits, %ld mismatches\n", n, hits, bad);
    return bad ? 1 : 0;
}
