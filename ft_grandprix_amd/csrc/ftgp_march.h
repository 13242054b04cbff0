// ftgp_march.h -- the LiDAR march and its acceleration structure, shared by the HIP kernels and by the host harness
// (tools/march_check.cpp compiles exactly these functions for the CPU and checks them against the plain-DDA specification).
//
// Specification of a ray (round 5: crossing times in coordinates RELATIVE to the ray's start cell, one fused
// multiply-add each).  Origin (pu, pv) in pixels, direction (du, dv) in pixels per unit of range.  Start cell (ix0, iy0) = floor of the
// origin.  Per axis, with p the origin's coordinate, i0 its floor and d the direction component:
//   f  = p - (float)i0                          the origin's offset inside the start cell (exact)
//   g  = d < 0 (or -0) ? 1.0f - f : f           the same seen in the direction of travel (one rounding when mirrored), 0 <= g <= 1
//   iv = |1 / d|, correctly rounded; +inf (d = 0: that axis is never stepped) is replaced by FLT_MAX so that 0 * iv stays 0
//   c  = g * iv                                 (one rounding)
//   S(k) = fma((float)k, iv, -c)                crossing time of the k-th cell boundary the ray meets on this axis, k = 1, 2, ...
// Plain cell-by-cell DDA: after mx steps in x and my steps in y the next step is in x iff Sx(mx + 1) < Sy(my + 1) (a tie steps in y);
// the range is |crossing time| of the step that enters the first wall pixel, 0 in a wall, -1 off the image.  Nothing accumulates, so any
// march that skips wall-free cells and re-synchronises with these comparisons returns the same bits.  (Rounds 1-4 specified
// S = ((float)b - p) * (1 / d) on absolute boundary coordinates b: a conversion, a subtraction and a multiplication per axis and look-up
// where this form needs a conversion and one fma, and the cancellation of two coordinates of a thousand pixels in every time.)
//
// Sector box field.  Both axes are mirrored so that every ray travels towards +x', +y': the march counts cells from the start cell in the
// direction of travel (mx, my >= 0).  Directions are binned into FTGP_SECTORS = 8 * NS sectors: the octant (mirror x, mirror y,
// dominant axis) and, inside it, NS equal slices of the slope minor/major (NS = 1, 2, 4 or 8: a handle's choice at ftgp_create --
// more slices mean fewer march iterations and a larger field; FTGP_SECTORS = 64 is the largest and what the host tools use).  For each pixel and sector one 16-bit
// entry holds a box of pixels with its corner at the pixel, extending AHEAD of the ray: low byte kx, high byte ky (cells
// along x' / y'); 0 = the pixel is a wall.  Only the part of the box that a ray of the sector can reach from anywhere
// inside the pixel has to be wall-free (the cone of the sector, widened by one cell so that rays through pixel corners --
// ties of the DDA -- and binary32 rounding stay inside it); walls beside or behind the cone never shorten a jump.
// The planes carry a one-pixel ring around the image whose entries are FTGP_FIELD_OUT (kx = 0 terminates the march,
// ky = 1 tells it from a wall), and every box is clipped at the image edge, so a ray that leaves the image lands exactly
// on a ring cell: the march needs no bounds test and no direction-dependent select.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FTGP_HD __host__ __device__ __forceinline__
#else
#define FTGP_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
FTGP_HD uint32_t ftgp_bits(float x) { return __float_as_uint(x); }
FTGP_HD float ftgp_float(uint32_t b) { return __uint_as_float(b); }
#else
FTGP_HD uint32_t ftgp_bits(float x) { uint32_t b; memcpy(&b, &x, 4); return b; }
FTGP_HD float ftgp_float(uint32_t b) { float x; memcpy(&x, &b, 4); return x; }
#endif

#define FTGP_FIELD_OUT 0x0100u
#ifndef FTGP_SECTORS
#define FTGP_SECTORS 64
#endif

#define FTGP_SLOPE_SLICES (FTGP_SECTORS / 8)
static_assert(FTGP_SECTORS == 8 || FTGP_SECTORS == 16 || FTGP_SECTORS == 32 || FTGP_SECTORS == 64 || FTGP_SECTORS == 128, "8, 16, 32, 64 or 128 sectors");

// Entry of pixel (x, y) for sector = (mirror x) | (mirror y) << 1 | (y-dominant) << 2 | (slope slice) << 3.
//   runx[d][y][x] wall-free run length starting at the pixel along +x (d = 0) / -x (d = 1); runy likewise
//                 (0 on walls; 65535 = to the image edge and beyond)
// In (major A, minor B) coordinates of the sector a ray that starts anywhere in cell (0, 0) with slope in [lo, hi] =
// [qa / NS, qb / NS] can only be in row j of column i if  lo * (i - 1) - 1 <= j <= hi * (i + 1) + 1  (closed: corner touches count).  For every
// box height kB the widest admissible kA is the first wall met by the reachable part of rows 0 .. kB - 1; the pair that
// maximises the travel min(kA, kB / slope) summed over four slopes of the sector is stored.
FTGP_HD uint32_t ftgp_box_entry(const uint16_t* runx, const uint16_t* runy, int W, int H, int x, int y, int sector, int NS = FTGP_SLOPE_SLICES)
{
    const size_t plane = (size_t)W * H;
    if (runx[(size_t)y * W + x] == 0) return 0u;                 // wall
    const int q = sector & 3, dom = (sector >> 2) & 1;
    const int qa = sector >> 3, qb = qa + 1;                     // NS = slope slices per octant (a handle's choice: ftgp_create)
    const int sx = (q & 1) ? -1 : 1, sy = (q & 2) ? -1 : 1;
    const int ax = dom == 0 ? sx : 0, ay = dom == 0 ? 0 : sy;    // unit step along the major / minor axis
    const int bx = dom == 0 ? 0 : sx, by = dom == 0 ? sy : 0;
    const uint16_t* run = dom == 0 ? runx + ((q & 1) ? plane : 0) : runy + ((q & 2) ? plane : 0);
    // distance to the image edge: boxes stop there, so that the cell after the box is a ring cell
    const int eX = sx > 0 ? W - x : x + 1, eY = sy > 0 ? H - y : y + 1;
    const int eA = dom == 0 ? eX : eY, eB = dom == 0 ? eY : eX;
    // four sample slopes n_k / (8 NS), n_k = 8 qa + 1, 3, 5, 7, as travel = min(cA * kA, cB_k * kB) with integer coefficients
    const long n0 = 8 * qa + 1, n1 = 8 * qa + 3, n2 = 8 * qa + 5, n3 = 8 * qa + 7;
    const long cA = n0 * n1 * n2 * n3;
    const long cB0 = 8 * NS * n1 * n2 * n3, cB1 = 8 * NS * n0 * n2 * n3, cB2 = 8 * NS * n0 * n1 * n3, cB3 = 8 * NS * n0 * n1 * n2;
    long best = -1; int bA = 1, bB = 1, m = 65535;
    for (int j = 0; j < 255 && j < eB; ++j) {
        // first column of row j that a ray of the sector can reach:  j <= hi * (i + 1) + 1  <=>  i >= ceil(NS (j - 1) / qb) - 1
        int cj = j >= 1 ? (NS * (j - 1) + qb - 1) / qb - 1 : 0;
        cj = cj < 0 ? 0 : cj;
        const int px = x + ax * cj + bx * j, py = y + ay * cj + by * j;
        int lim = 65535;                                          // column of the first wall of the reachable part of row j
        if (px >= 0 && px < W && py >= 0 && py < H) { const int r = run[(size_t)py * W + px]; if (r < 65535) lim = cj + r; }
        if (qa > 0 && lim > (NS * (j + 1)) / qa + 1) lim = 65535;            // ... and the last one: j >= lo * (i - 1) - 1  <=>  i <= NS (j + 1) / qa + 1
        m = lim < m ? lim : m;
        if (4 * cA * (long)m <= best) break;                      // no taller box can do better
        int kA = m < 255 ? m : 255; kA = kA < eA ? kA : eA;
        const int kB = j + 1;
        const long a = cA * kA;
        const long b0 = cB0 * kB, b1 = cB1 * kB, b2 = cB2 * kB, b3 = cB3 * kB;
        const long score = (a < b0 ? a : b0) + (a < b1 ? a : b1) + (a < b2 ? a : b2) + (a < b3 ? a : b3);
        if (score > best) { best = score; bA = kA; bB = kB; }
    }
    const int kx = dom == 0 ? bA : bB, ky = dom == 0 ? bB : bA;
    return (uint32_t)kx | ((uint32_t)ky << 8);
}

// size of one sector plane in units of 256 bytes ((W + 2) x (H + 2) 16-bit entries, padded)
FTGP_HD uint32_t ftgp_plane256(int W, int H) { return (2u * (uint32_t)(W + 2) * (uint32_t)(H + 2) + 255u) >> 8; }

// Distance to a pixel boundary below which the landing estimate floor(g + |d| * s) is not trusted and the specification's
// comparison decides.  With unit roundoff u = 2^-24 and M the largest coordinate (relative to the start cell: below the image size): the
// estimate fl(g + |d| * s) is off by at most u * M from g + |d| * s (one rounding), and the specification's own decision -- S(k) <= s with
// S(k) = fl(k * iv - c), c = fl(g * iv), iv = fl(1 / d): three roundings, each at most u of a time that is at most the time to the
// boundary -- flips at a point that is off by at most 3u * M from where g + |d| * s crosses k; both use the same binary32 s.  So the
// two can only disagree within 4u * M of a boundary:  eps = 4u * pow2ceil(max(W, H) + 2) = 2^-22 * pow2ceil(...), 2^-11 px for a
// 1600-px track (the power of two above M leaves another 1.0 .. 2.0 x).
FTGP_HD float ftgp_snap_eps(int W, int H)
{
    const int m = (W > H ? W : H) + 2;
    int p = 1; while (p < m) p <<= 1;
    return (float)p * (1.0f / 4194304.0f);
}

#define FTGP_IV_MAX 3.40282347e+38f      /* FLT_MAX: |1 / d| of a direction component that is 0 (or so small that 1 / d overflows) */

// One ray in the mirrored frame, relative to its start cell.  The byte offset of the entry of relative cell (mx, my) is linear in it:
// base + mx * ax + my * ay  with ax = +-2, ay = +-2 * fstride (the signs undo the mirrors) and base = the start cell's own entry.
struct FtgpRay {
    float gu, gv;                         // origin inside the start cell, seen in the direction of travel (specification: g)
    float du, dv;                         // |direction|
    float ivx, ivy;                       // |1 / direction| (FTGP_IV_MAX where the direction is 0)
    float cx, cy;                         // g * iv: the crossing time of boundary k is fma(k, iv, -c)
    float s, result;                      // crossing time of the last step; range when the ray ends on no wall (-1, see ftgp_ray_range)
    int mx, my;                           // cells travelled along x' / y'
    int base, ax, ay;                     // byte offset of cell (mx, my) in the field = base + mx * ax + my * ay
};

// A ray that marches nothing: every cell of it maps to ring cell (0, 0) of plane 0, which terminates at once and leaves `result` alone.
FTGP_HD void ftgp_ray_park(FtgpRay& r, float result)
{
    r.gu = r.gv = r.du = r.dv = r.ivx = r.ivy = r.cx = r.cy = 0.0f;
    r.s = 0.0f; r.result = result;
    r.mx = r.my = 0; r.base = 0; r.ax = r.ay = 0;
}

// fstride = W + 2 (cells per plane row); plane256 = bytes per sector plane / 256 (planes are padded to a multiple of 256 B).
// Entry of the sector table: { byte offset of pixel (0, 0) in the sector's plane, ax, ay, 0 }.
// `plane`: which plane of the field serves a ray of this sector (a handle may keep fewer planes than FTGP_SECTORS: ftgp_sector_table).
FTGP_HD void ftgp_sector_entry(int32_t* e, uint32_t sector, int fstride, uint32_t plane256, uint32_t plane)
{
    const int mxm = -(int)(sector & 1u), mym = -(int)((sector >> 1) & 1u);
    e[0] = (int)((plane * plane256) << 8) + 2 * (fstride + 1);      // the ring: one row and one column
    e[1] = (2 ^ mxm) - mxm;                                        // +-2
    e[2] = ((2 * fstride) ^ mym) - mym;                            // +-2 * fstride
    e[3] = 0;
}
FTGP_HD void ftgp_sector_entry(int32_t* e, uint32_t sector, int fstride, uint32_t plane256) { ftgp_sector_entry(e, sector, fstride, plane256, sector); }

// The sector table of a field that holds `coarse` sectors' planes.  A ray's sector is always found among FTGP_SECTORS; the coarse sector it
// belongs to has the same mirror / axis bits and its slope slice shifted down (the coarse slices are unions of fine ones).  Returns the
// number of planes.  (Round 4 also kept all FTGP_SECTORS planes for a ray's FIRST look-up as an experiment -- fewer iterations, slower:
// profiles/round4/ab_fine_first.log -- which round 5 removed.)
FTGP_HD int ftgp_sector_table(int32_t (*tab)[4], int coarse, int fstride, uint32_t plane256)
{
    int shift = 0; while ((FTGP_SECTORS >> shift) > coarse) ++shift;
    for (uint32_t s = 0; s < (uint32_t)FTGP_SECTORS; ++s)
        ftgp_sector_entry(tab[s], s, fstride, plane256, ((s >> 3) >> shift) << 3 | (s & 7u));
    return coarse;
}

// Sector of a direction: (mirror x) | (mirror y) << 1 | (y-dominant) << 2 | (slope slice) << 3.  Mirrors and dominant axis come out of
// sign bits and one compare; the OPPOSITE direction (-du, -dv) has exactly this sector with the two mirror bits flipped (sector ^ 3).
// nsf = (slope slices per octant) * 0.99999: the slice of a ray is floor(minor / major * nsf).
#define FTGP_SLICE_FACTOR(ns) ((float)(ns) * 0.99999f)
FTGP_HD uint32_t ftgp_ray_sector(float du, float dv, float ivx, float ivy, float nsf = FTGP_SLICE_FACTOR(FTGP_SLOPE_SLICES))
{
    // A direction component of -0 mirrors its axis too: harmless, the ray never steps along it.  (The dominant axis stays a compare
    // and two selects: fminf / fmaxf bring canonicalising v_max x, x, x along, and integer min / max on the bit patterns cost what the
    // selects cost.)
    const uint32_t bu = ftgp_bits(du), bv = ftgp_bits(dv);
    const float adu = fabsf(du), adv = fabsf(dv);
    const bool ydom = !(adu >= adv);
    uint32_t sector = (bu >> 31) | ((bv >> 31) << 1) | (ydom ? 4u : 0u);
    {
        // slope slice = floor(NS * minor / major), with the reciprocal of the major component that is at hand anyway.  A few units
        // of rounding may put a ray that runs along a slice boundary into the neighbouring slice: harmless, the boxes of a
        // slice hold for every slope within a cell's margin of it (ftgp_box_entry), and a box that holds gives the
        // specification's result whichever slice it came from.  The factor just below NS keeps slope 1 in the last slice.
        const float mn = ydom ? adu : adv, inv_mj = ydom ? ivy : ivx;
        const float scaled = (mn * inv_mj) * nsf;             // (one slice per octant: the factor is below 1 and the slice 0)
#if defined(__HIP_DEVICE_COMPILE__)
        int slice; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(slice) : "v"(scaled));
#else
        int slice = (int)floorf(scaled);
#endif
        // (a direction of (0, 0) or a NaN cannot occur: the components are a rotated unit vector, scaled)
        sector |= (uint32_t)slice << 3;
    }
    return sector;
}

// Puts a ray of known sector on its start cell.  (ftgp_ray_init = ftgp_ray_sector + ftgp_ray_place; the step kernel places the ray
// OPPOSITE to one it has just marched -- same |direction|, reciprocals and slope slice -- with sector ^ 3 and skips the first half.)
// ivx, ivy = |1 / du|, |1 / dv| correctly rounded (IEEE division), FTGP_IV_MAX where that is +inf.
// assume_inside (a compile-time constant at every call site): the caller guarantees that (pu, pv) is finite and lies on the
// image -- the step kernel does when every LiDAR centre of the workgroup is a ring radius plus two pixels away from the image
// edge (frame_write) -- so the test, and the selects that park an off-image ray, are not needed.
// sector_tab (optional, [FTGP_SECTORS][4] as ftgp_sector_table() fills it): plane offset and strides of every sector precomputed, one
// 16-byte read instead of six integer instructions.
FTGP_HD void ftgp_ray_place(FtgpRay& r, float pu, float pv, float du, float dv, float ivx, float ivy, uint32_t sector, int W, int H, int fstride, uint32_t plane256,
                            bool assume_inside = false, const int32_t* sector_tab = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int ix0, iy0;                                             // floor and convert in one instruction; the conversion saturates
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix0) : "v"(pu));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy0) : "v"(pv));
    const bool inside = assume_inside || (!__builtin_isunordered(pu, pv) && (unsigned)ix0 < (unsigned)W && (unsigned)iy0 < (unsigned)H);   // a NaN converts to 0: test it
    const float fu = __builtin_amdgcn_fractf(pu), fv = __builtin_amdgcn_fractf(pv);      // p - floor(p): exact for every p >= 0 (an origin off the image is parked)
#else
    const float fx = floorf(pu), fy = floorf(pv);
    const bool inside = assume_inside || (fx >= 0.0f && fx < (float)W && fy >= 0.0f && fy < (float)H);
    const int ix0 = (int)fx, iy0 = (int)fy;
    const float fu = pu - fx, fv = pv - fy;
#endif
    const bool negu = du < 0.0f, negv = dv < 0.0f;           // (a component of -0 counts as not negative here although its sector mirrors the axis: it never steps)
    r.gu = negu ? 1.0f - fu : fu;
    r.gv = negv ? 1.0f - fv : fv;
    r.du = fabsf(du); r.dv = fabsf(dv);
    r.ivx = ivx; r.ivy = ivy;
    r.cx = r.gu * ivx; r.cy = r.gv * ivy;
    r.s = 0.0f;
    if (!assume_inside) r.result = -1.0f;
    r.mx = 0; r.my = 0;
    int off;
    if (sector_tab) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int4 e4 = reinterpret_cast<const int4*>(sector_tab)[sector];       // one 16-byte LDS read
        off = e4.x; r.ax = e4.y; r.ay = e4.z;
#else
        const int32_t* e = sector_tab + 4 * sector;
        off = e[0]; r.ax = e[1]; r.ay = e[2];
#endif
    } else {
        int32_t e[4]; ftgp_sector_entry(e, sector, fstride, plane256);
        off = e[0]; r.ax = e[1]; r.ay = e[2];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    int cell;                                                 // iy0 * fstride + ix0 in one 24-bit multiply-add (the compiler prefers a 64-bit one)
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cell) : "v"(iy0), "s"(fstride), "v"(ix0));
#else
    const int cell = iy0 * fstride + ix0;
#endif
    r.base = off + 2 * cell;                                  // the start cell's entry, in the image's own orientation (the strides carry the mirrors)
    if (!inside) { r.base = 0; r.ax = r.ay = 0; }             // starts off the image: every cell maps to ring cell (0, 0) of plane 0, result stays -1
}

FTGP_HD void ftgp_ray_init(FtgpRay& r, float pu, float pv, float du, float dv, float ivx, float ivy, int W, int H, int fstride, uint32_t plane256,
                           bool assume_inside = false, const int32_t* sector_tab = nullptr, float nsf = FTGP_SLICE_FACTOR(FTGP_SLOPE_SLICES))
{
    ftgp_ray_place(r, pu, pv, du, dv, ivx, ivy, ftgp_ray_sector(du, dv, ivx, ivy, nsf), W, H, fstride, plane256, assume_inside, sector_tab);
}

// |1 / d| as the specification wants it, on the host (the device has rcp_abs2)
FTGP_HD float ftgp_iv(float d) { const float z = fabsf(1.0f / d); return z < FTGP_IV_MAX ? z : FTGP_IV_MAX; }

// The on-image test of ftgp_ray_init for a caller that initialised with assume_inside although it could not promise it:
// a ray whose origin is off the image (or not a number) is parked exactly as ftgp_ray_init would have parked it.
FTGP_HD void ftgp_ray_park_if_outside(FtgpRay& r, float pu, float pv, int W, int H)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int ix0, iy0;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix0) : "v"(pu));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy0) : "v"(pv));
    const bool inside = !__builtin_isunordered(pu, pv) && (unsigned)ix0 < (unsigned)W && (unsigned)iy0 < (unsigned)H;
#else
    const float fx = floorf(pu), fy = floorf(pv);
    const bool inside = fx >= 0.0f && fx < (float)W && fy >= 0.0f && fy < (float)H;
#endif
    if (!inside) { r.base = 0; r.ax = r.ay = 0; }
}

FTGP_HD int ftgp_ray_offset(const FtgpRay& r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int a, b;                                                 // two multiply-adds (the compiler prefers two multiplies and a three-operand add)
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a) : "v"(r.my), "v"(r.ay), "v"(r.base));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(b) : "v"(r.mx), "v"(r.ax), "v"(a));
    return b;
#else
    return r.mx * r.ax + (r.my * r.ay + r.base);
#endif
}

struct FtgpStep { float sn; int t, xe, ye; bool stepx, live; };       // live: the ray is not on its terminal cell yet

// One generic iteration, first half: w is the field entry of the ray's cell.  Jumps to the far edge of the box
// (s = min(sX, sY) of its exit boundaries) and estimates the transverse cell as floor(g + |d| * s).  Returns true when the
// landing point is within `eps` of a pixel boundary: the caller then runs ftgp_ray_fix() before ftgp_ray_commit().
// Otherwise the estimate needs no clamp: the ray leaves the box through the edge it reaches first, i.e. at a transverse
// coordinate inside the box's span, and the estimate is off by far less than eps.
// Vector instructions are chosen by what they cost on gfx950 (profiles/round2/valu_cost.log): add / logic / shift / mov and binary32
// add / mul / fma issue in 2 cycles per wave, everything else used here (selects, compares, conversions, 24-bit multiplies,
// three-operand integer forms, fract) in 4.
FTGP_HD bool ftgp_ray_step(const FtgpRay& r, uint32_t w, float eps, FtgpStep& st)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // byte operands straight out of the entry (SDWA), floor-and-convert in one instruction, hardware fract
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(st.xe) : "v"(r.mx), "v"(w));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(st.ye) : "v"(r.my), "v"(w));
    st.live = st.xe != r.mx;                                  // kx == 0: wall or ring cell
#else
    const int kx = (int)(w & 255u), ky = (int)(w >> 8) & 255;
    st.live = kx != 0;                                        // kx == 0: wall or ring cell
    st.xe = r.mx + kx; st.ye = r.my + ky;
#endif
    const float sX = fmaf((float)st.xe, r.ivx, -r.cx);       // the box's far boundaries are the xe-th / ye-th the ray meets
    const float sY = fmaf((float)st.ye, r.ivy, -r.cy);
    st.stepx = sX < sY;
    st.sn = st.stepx ? sX : sY;
    // both landing estimates, then the one that applies: fma(dv, sX, gv) after an x-jump, fma(du, sY, gu) after a y-jump
    // (two scalar fmas: packed binary32 instructions cost more than two scalar ones on gfx950, measured -- the library is
    // also built with -fno-slp-vectorize for that reason)
    const float vX = fmaf(r.dv, sX, r.gv), vY = fmaf(r.du, sY, r.gu);
    const float v = st.stepx ? vX : vY;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(st.t) : "v"(v));
    const float frac = __builtin_amdgcn_fractf(v);            // v - floor(v), kept below 1: only ever compared with eps and 1 - eps
#else
    const float fl = floorf(v);
    st.t = (int)fl;
    const float frac = v - fl;
#endif
    return st.live & (fabsf(frac - 0.5f) > 0.5f - eps);    // within eps of a boundary (and never for a NaN)
}

// The specification's comparison for a landing point close to a boundary.  "Close" means within eps of ONE pixel boundary -- the
// nearest integer to the estimate -- and eps is far below half a pixel, so that boundary is the only one in doubt: has the ray
// crossed it by the time sn of the jump (Sy(b) <= sn after an x-jump, Sx(b) < sn after a y-jump: a tie steps in y first)?  The cell
// is the one beyond it if so, the one before it if not, kept inside the transverse span [cur, hi] of the box.  (b = 0 is the start
// cell's own rear boundary: S(0) = -c <= 0 says "crossed", the cell is 0.)
FTGP_HD int ftgp_ray_fix(const FtgpRay& r, const FtgpStep& st)
{
    const float tc = st.stepx ? r.cy : r.cx, tinv = st.stepx ? r.ivy : r.ivx;
    const int cur = st.stepx ? r.my : r.mx, hi = (st.stepx ? st.ye : st.xe) - 1;
    const float v = st.stepx ? fmaf(r.dv, st.sn, r.gv) : fmaf(r.du, st.sn, r.gu);       // the landing estimate of ftgp_ray_step()
    const float bf = rintf(v);                                // nearest boundary (ties cannot occur here: the fraction is within eps of 0 or 1)
    const float S = fmaf(bf, tinv, -tc);
    const bool crossed = st.stepx ? (S <= st.sn) : (S < st.sn);
    int t = (int)bf - (crossed ? 0 : 1);
    t = t < cur ? cur : t; t = t > hi ? hi : t;
    return t;
}

// Second half.  With hold (the default) a finished ray stays on its terminal cell and keeps the crossing time into it.
// hold = false (a compile-time constant at the call site) is for a caller that never looks a finished ray's cell up again and
// has put the crossing time aside before the lookup that ended the ray: cell and time of a finished ray then drift, which saves the
// three selects that would hold them.
FTGP_HD void ftgp_ray_commit(FtgpRay& r, const FtgpStep& st, int t, bool hold = true)
{
    const int nix = st.stepx ? st.xe : t, niy = st.stepx ? t : st.ye;
    if (hold) { r.mx = st.live ? nix : r.mx; r.my = st.live ? niy : r.my; r.s = st.live ? st.sn : r.s; }
    else { r.mx = nix; r.my = niy; r.s = st.sn; }
}

// Range of a ray that sits on its terminal cell, whose entry is w: the crossing time into a wall cell, `result` (-1, or 0 for a
// switched-off rangefinder) on a ring cell.  |s|: a ray that starts on a boundary can produce -0.
FTGP_HD float ftgp_ray_range(const FtgpRay& r, uint32_t w) { return w == 0u ? fabsf(r.s) : r.result; }

// single ray against a field image (host harness, tests)
// (sector_tab: as ftgp_create builds it -- the field may hold fewer planes than FTGP_SECTORS)
FTGP_HD float ftgp_march_one(const uint16_t* field, int W, int H, float eps, float pu, float pv, float du, float dv, const int32_t* sector_tab = nullptr)
{
    const int fstride = W + 2;
    const uint32_t plane256 = ftgp_plane256(W, H);
    FtgpRay r; ftgp_ray_init(r, pu, pv, du, dv, ftgp_iv(du), ftgp_iv(dv), W, H, fstride, plane256, false, sector_tab);
    uint32_t w = FTGP_FIELD_OUT;
    for (int guard = 0; guard < 4 * 8192; ++guard) {
        w = field[(uint32_t)ftgp_ray_offset(r) >> 1];        // byte offsets are 32-bit unsigned (ftgp_create keeps the field below 4 GiB)
        FtgpStep st;
        const bool near = ftgp_ray_step(r, w, eps, st);
        const int t = near ? ftgp_ray_fix(r, st) : st.t;
        ftgp_ray_commit(r, st, t);
        if (!st.live) return ftgp_ray_range(r, w);
    }
    return r.result;
}
