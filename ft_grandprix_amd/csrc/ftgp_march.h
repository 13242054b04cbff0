// ftgp_march.h -- the LiDAR march and its acceleration structure, shared by the HIP kernels and by the host harness
// (tools/march_check.cpp compiles exactly these functions for the CPU and checks them against the plain-DDA specification).
//
// Specification of a ray: plain cell-by-cell DDA in binary32; crossing times
//   sX(b) = ((float)b - pu) * (1/du),  sY(b) = ((float)b - pv) * (1/dv);  x-step iff sX < sY (a tie steps in y);
// the range is |crossing time| of the step that enters the first wall pixel, 0 in a wall, -1 off the image.
// Nothing accumulates, so any march that skips wall-free cells and re-synchronises with these comparisons returns the
// same bits.
//
// Octant box field.  Both axes are mirrored so that every ray travels towards +x', +y' (x' = -x is exact in IEEE
// arithmetic and maps cell i to ~i).  For each pixel and each of the 8 direction octants (mirror x, mirror y, dominant
// axis) one 16-bit entry holds a wall-free box of pixels with its corner at the pixel, extending AHEAD of the ray:
// low byte kx, high byte ky (cells along x' / y').  0 = the pixel is a wall.  The planes carry a one-pixel ring around
// the image whose entries are FTGP_FIELD_OUT (kx = 0 terminates the march, ky = 1 tells it from a wall), and every box
// is clipped at the image edge, so a ray that leaves the image lands exactly on a ring cell: the march needs no
// bounds test and no direction-dependent select.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FTGP_HD __host__ __device__ __forceinline__
#else
#define FTGP_HD static inline
#endif

#define FTGP_FIELD_OUT 0x0100u
#define FTGP_OCTANTS 8

// Entry of pixel (x, y) for octant oct = (mirror x) | (mirror y) << 1 | (y-dominant) << 2.
//   ksq[q][y][x]  side of the largest wall-free square with its corner at the pixel, extending towards quadrant q
//                 (pixels beyond the image count as free; 0 on walls; clamped to 255)
//   runx[d][y][x] wall-free run length starting at the pixel along +x (d = 0) / -x (d = 1); runy likewise (65535 = to the edge and beyond)
// Per octant the box is the square or, when it reaches farther along the dominant axis, the largest 2h x h rectangle
// (2h along the dominant axis), found by walking h rows (columns) with a running minimum of the run lengths.
FTGP_HD uint32_t ftgp_box_entry(const uint8_t* ksq, const uint16_t* runx, const uint16_t* runy, int W, int H, int x, int y, int oct)
{
    const size_t plane = (size_t)W * H, i = (size_t)y * W + x;
    const int q = oct & 3, dom = oct >> 2;
    const int k = ksq[(size_t)q * plane + i];
    if (k == 0) return 0u;
    const int sx = (q & 1) ? -1 : 1, sy = (q & 2) ? -1 : 1;
    const uint16_t* rx = runx + ((q & 1) ? plane : 0);
    const uint16_t* ry = runy + ((q & 2) ? plane : 0);
    int h = 0, m = 65535;
    for (; h < 127; ++h) {
        int r;
        if (dom == 0) { const int yy = y + sy * h; r = (yy >= 0 && yy < H) ? (int)rx[(size_t)yy * W + x] : 65535; }
        else          { const int xx = x + sx * h; r = (xx >= 0 && xx < W) ? (int)ry[(size_t)y * W + xx] : 65535; }
        m = r < m ? r : m;
        if (m < 2 * (h + 1)) break;
    }
    const int ks = k < 127 ? k : 127;
    int kmaj = ks, kmin = ks;
    if (2 * h > ks) { kmaj = 2 * h; kmin = h; }
    int kx = dom == 0 ? kmaj : kmin, ky = dom == 0 ? kmin : kmaj;
    // clip at the image edge: the cell after the box is then a ring cell
    const int ex = sx > 0 ? W - x : x + 1, ey = sy > 0 ? H - y : y + 1;
    kx = kx < ex ? kx : ex; ky = ky < ey ? ky : ey;
    kx = kx < 255 ? kx : 255; ky = ky < 255 ? ky : 255;
    return (uint32_t)kx | ((uint32_t)ky << 8);
}

// One ray in the mirrored frame.  mx / my (0 or -1) turn the mirrored cell back into the true pixel (ix ^ mx, iy ^ my);
// byte offset of its entry = offC + 2 * ((iy ^ my) * fstride + (ix ^ mx)), offC = start of the octant's plane + the ring.
struct FtgpRay {
    float pum, pvm, dum, dvm, ivx, ivy;   // mirrored origin, |direction|, |1 / direction| (+inf where the direction is 0)
    float s, result;                      // crossing time of the last step; range (-1: none yet / off the image)
    int ix, iy;                           // mirrored cell
    int offC, mx, my;
};

// A ray that marches nothing: it sits on ring cell (0, 0) of plane 0, which terminates at once and leaves `result` alone.
FTGP_HD void ftgp_ray_park(FtgpRay& r, float result)
{
    r.pum = r.pvm = r.dum = r.dvm = r.ivx = r.ivy = 0.0f;
    r.s = 0.0f; r.result = result;
    r.ix = r.iy = 0; r.offC = 0; r.mx = r.my = 0;
}

// fstride = W + 2 (cells per plane row), plane_bytes = 2 * (W + 2) * (H + 2)
FTGP_HD void ftgp_ray_init(FtgpRay& r, float pu, float pv, float du, float dv, int W, int H, int fstride, uint32_t plane_bytes)
{
    const float fx = floorf(pu), fy = floorf(pv);
    const bool inside = fx >= 0.0f && fx < (float)W && fy >= 0.0f && fy < (float)H;
    const int ix0 = (int)fx, iy0 = (int)fy;
    const bool mx = du < 0.0f, my = dv < 0.0f;
    const float adu = fabsf(du), adv = fabsf(dv);
    const int dom = adu >= adv ? 0 : 1;
    r.pum = mx ? -pu : pu; r.pvm = my ? -pv : pv;
    r.dum = adu; r.dvm = adv;
    r.ivx = fabsf(1.0f / du);          // IEEE division: +inf where the direction is 0 (that axis is never stepped)
    r.ivy = fabsf(1.0f / dv);
    r.s = 0.0f; r.result = -1.0f;
    const int oct = (mx ? 1 : 0) | (my ? 2 : 0) | (dom << 2);
    r.mx = mx ? -1 : 0; r.my = my ? -1 : 0;
    r.ix = ix0 ^ r.mx; r.iy = iy0 ^ r.my;
    r.offC = (int)((uint32_t)oct * plane_bytes) + 2 * (fstride + 1);
    if (!inside) { r.ix = r.iy = 0; r.offC = 0; r.mx = r.my = 0; }       // starts off the image: ring cell (0, 0), result stays -1
}

FTGP_HD int ftgp_ray_offset(const FtgpRay& r, int fstride)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return r.offC + 2 * (__mul24(r.iy ^ r.my, fstride) + (r.ix ^ r.mx));
#else
    return r.offC + 2 * ((r.iy ^ r.my) * fstride + (r.ix ^ r.mx));
#endif
}

struct FtgpStep { float sn; int t, cur, hi, xe, ye; bool stepx, done; };

// One generic iteration, first half: w is the field entry of the ray's cell.  Jumps to the far edge of the box
// (s = min(sX, sY) of its exit boundaries) and estimates the transverse cell as floor(p + d * s).  Returns true when the
// landing point is within `eps` of a pixel boundary: the caller then runs ftgp_ray_fix() before ftgp_ray_commit().
FTGP_HD bool ftgp_ray_step(FtgpRay& r, uint32_t w, float eps, FtgpStep& st)
{
    const int kx = (int)(w & 255u), ky = (int)(w >> 8);
    st.done = kx == 0;                                        // wall or ring cell
    r.result = (w == 0u) ? fabsf(r.s) : r.result;             // |s|: a ray that starts on a boundary can produce -0
    st.xe = r.ix + kx; st.ye = r.iy + ky;
    const float sX = ((float)st.xe - r.pum) * r.ivx;
    const float sY = ((float)st.ye - r.pvm) * r.ivy;
    st.stepx = sX < sY;
    st.sn = st.stepx ? sX : sY;
    const float tp = st.stepx ? r.pvm : r.pum, td = st.stepx ? r.dvm : r.dum;
    const float v = fmaf(td, st.sn, tp);
    const float fl = floorf(v);
    int t = (int)fl;
    st.cur = st.stepx ? r.iy : r.ix; st.hi = (st.stepx ? st.ye : st.xe) - 1;
    t = t < st.cur ? st.cur : t; t = t > st.hi ? st.hi : t;
    st.t = t;
    const float frac = v - fl;
    return !st.done & (fabsf(frac - 0.5f) > 0.5f - eps);    // within eps of a boundary (and never for a NaN)
}

// the specification's comparisons for a landing point close to a boundary (sY(b) <= s after an x-jump, sX(b) < s after a y-jump)
FTGP_HD int ftgp_ray_fix(const FtgpRay& r, const FtgpStep& st)
{
    const float tp = st.stepx ? r.pvm : r.pum, tinv = st.stepx ? r.ivy : r.ivx;
    const int t = st.t;
    const float Sa = ((float)t - tp) * tinv, Sb = ((float)(t + 1) - tp) * tinv;
    const bool ca = st.stepx ? (Sa <= st.sn) : (Sa < st.sn), cb = st.stepx ? (Sb <= st.sn) : (Sb < st.sn);
    const bool dec = (t > st.cur) & !ca;
    const bool inc = !dec & (t < st.hi) & cb;
    return t + (inc ? 1 : 0) - (dec ? 1 : 0);
}

FTGP_HD void ftgp_ray_commit(FtgpRay& r, const FtgpStep& st, int t)
{
    const int nix = st.stepx ? st.xe : t, niy = st.stepx ? t : st.ye;
    r.ix = st.done ? r.ix : nix; r.iy = st.done ? r.iy : niy;     // a finished ray stays on its terminal cell
    r.s = st.done ? r.s : st.sn;
}

// single ray against a field image (host harness, tests)
FTGP_HD float ftgp_march_one(const uint16_t* field, int W, int H, float eps, float pu, float pv, float du, float dv)
{
    const int fstride = W + 2;
    const uint32_t plane_bytes = 2u * (uint32_t)fstride * (uint32_t)(H + 2);
    FtgpRay r; ftgp_ray_init(r, pu, pv, du, dv, W, H, fstride, plane_bytes);
    for (int guard = 0; guard < 4 * 8192; ++guard) {
        const uint32_t w = field[ftgp_ray_offset(r, fstride) >> 1];
        FtgpStep st;
        const bool near = ftgp_ray_step(r, w, eps, st);
        const int t = near ? ftgp_ray_fix(r, st) : st.t;
        ftgp_ray_commit(r, st, t);
        if (st.done) break;
    }
    return r.result;
}
