// pg_math.h -- libm functions of the reference's state path, restated so that the device reproduces the host bits.
//
// atan2f: Entity::face_direction (reference src/entity.cpp:84-88) stores -atan2f(dy, dx) + offset in the entity's
// rotation, which is part of the serialized state, so the float result must match glibc's bit for bit.  glibc 2.35
// (the libm the compiled reference links here and on the GPU box) implements atan2f/atanf with the fdlibm single
// precision algorithm (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c): argument reduction to one of four intervals,
// an odd/even split degree-11 polynomial in float arithmetic, hi/lo constants for atan(0.5), atan(1), atan(1.5),
// pi/2.  Restated here from the published algorithm; tests/test_device_math.py checks it against the host libm on
// a few million inputs.  Plain float operations only -- compile with -ffp-contract=off.
#pragma once
#include "wave.h"

namespace pgamd {

PG_DEV float pg_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = __builtin_bit_cast(int32_t, x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25 (or NaN)
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

PG_DEV float pg_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = __builtin_bit_cast(int32_t, x), hy = __builtin_bit_cast(int32_t, y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
    if (hx == 0x3f800000) return pg_atanf(y);              // x = 1.0
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);     // 2*sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = pg_atanf(__builtin_fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// Double precision atan2 for BasicAbstractGame::get_theta (reference src/basic-abstract-game.cpp:233-238: float dx, dy
// promoted to double, the result narrowed to float; only jumper's compass uses it, and only to place a line).  The
// published fdlibm algorithm (e_atan2.c, s_atan.c; error < 1 ulp): the same four-interval reduction as the float
// version above with a degree-11 odd polynomial in double.  The device libm's atan2 costs 41 VGPRs more in the render
// kernel (a wave per SIMD in jumper); after the narrowing to float both agree with glibc's correctly rounded atan2
// except on a ~2^-29 fraction of inputs (tests/test_device_math.py measures it on the host).
PG_DEV double pg_atan_d(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                           9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                           4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    const int64_t bits = __builtin_bit_cast(int64_t, x);
    const int32_t hx = (int32_t)(bits >> 32);
    const int32_t ix = hx & 0x7fffffff;
    double hi = 0, lo = 0;
    bool reduced = true;
    if (ix >= 0x44100000) {  // |x| >= 2^66 (or NaN)
        if (x != x) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3fdc0000) {  // |x| < 0.4375
        if (ix < 0x3e200000) return x;  // |x| < 2^-29
        reduced = false;
    } else {
        x = __builtin_fabs(x);
        if (ix < 0x3ff30000) {  // |x| < 1.1875
            if (ix < 0x3fe60000) {  // 7/16 <= |x| < 11/16
                hi = atanhi[0]; lo = atanlo[0];
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else {  // 11/16 <= |x| < 19/16
                hi = atanhi[1]; lo = atanlo[1];
                x = (x - 1.0) / (x + 1.0);
            }
        } else if (ix < 0x40038000) {  // |x| < 2.4375
            hi = atanhi[2]; lo = atanlo[2];
            x = (x - 1.5) / (1.0 + 1.5 * x);
        } else {  // 2.4375 <= |x| < 2^66
            hi = atanhi[3]; lo = atanlo[3];
            x = -1.0 / x;
        }
    }
    const double z = x * x;
    const double w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (!reduced) return x - x * (s1 + s2);
    const double r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}
PG_DEV double pg_atan2_d(double y, double x) {
    const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00, pi = 3.1415926535897931160E+00,
                 pi_lo = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    const int64_t bx = __builtin_bit_cast(int64_t, x), by = __builtin_bit_cast(int64_t, y);
    const int32_t hx = (int32_t)(bx >> 32), hy = (int32_t)(by >> 32);
    const uint32_t lx = (uint32_t)bx, ly = (uint32_t)by;
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if ((((uint32_t)hx - 0x3ff00000u) | lx) == 0) return pg_atan_d(y);  // x = 1.0
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                       // 2 * sign(x) + sign(y)
    if ((iy | ly) == 0) {                                              // y = 0
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if ((ix | lx) == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
    if (ix == 0x7ff00000) {                                                // x = +-inf
        if (iy == 0x7ff00000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            return m == 2 ? 3.0 * pi_o_4 + tiny : -3.0 * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0;
        if (m == 1) return -0.0;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7ff00000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y = +-inf
    const int32_t k = (iy - ix) >> 20;
    double z;
    if (k > 60) {  // |y / x| > 2^60
        z = pi_o_2 + 0.5 * pi_lo;
        m &= 1;
    } else if (hx < 0 && k < -60) z = 0.0;  // 0 > |y| / x > -2^-60
    else z = pg_atan_d(__builtin_fabs(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// Double precision sin / cos (rotated sprites' QTransform, bullet and thrust directions in bossfight / caveflyer / ninja /
// starpilot, jumper's compass): the published fdlibm algorithm (k_sin.c, k_cos.c, the Cody-Waite part of e_rem_pio2.c),
// error < 1 ulp like the device libm's -- but the same IEEE operations on the host and on the GPU, so the CPU tests
// (tests/emu) exercise exactly the arithmetic the kernels run, and several VGPRs lighter than the library routines.
// Every result is narrowed to float or truncated to a pixel / 16.16 coefficient by its caller; glibc's sin / cos are
// correctly rounded in nearly all cases, so the two differ after the narrowing about once in 2^29 calls.
// Domain: |x| <= 2^19 * pi/2 (8.2e5 rad; rotations are a few thousand rad at most); beyond it the reduction below loses
// accuracy gracefully instead of running the Payne-Hanek path.
PG_DEV double pg_ksin_d(double x, double y, bool have_tail) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const int32_t ix = (int32_t)(__builtin_bit_cast(int64_t, x) >> 32) & 0x7fffffff;
    if (ix < 0x3e400000 && (int)x == 0) return x;  // |x| < 2^-27
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (!have_tail) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
PG_DEV double pg_kcos_d(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const int32_t ix = (int32_t)(__builtin_bit_cast(int64_t, x) >> 32) & 0x7fffffff;
    if (ix < 0x3e400000 && (int)x == 0) return 1.0;  // |x| < 2^-27
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));  // |x| < 0.3
    const double qx = ix > 0x3fe90000 ? 0.28125 : __builtin_bit_cast(double, (int64_t)(ix - 0x00200000) << 32);  // x / 4
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
// x = n * pi/2 + (y0 + y1), |y0 + y1| <= pi/4; returns n mod 4 as a non-negative residue
PG_DEV int pg_rem_pio2_d(double x, double &y0, double &y1) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    const int32_t hx = (int32_t)(__builtin_bit_cast(int64_t, x) >> 32);
    const int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3fe921fb) {  // |x| <= pi/4
        y0 = x;
        y1 = 0;
        return 0;
    }
    const double t0 = __builtin_fabs(x);
    const int n = (int)(t0 * invpio2 + 0.5);
    const double fn = (double)n;
    double r = t0 - fn * pio2_1;
    double w = fn * pio2_1t;  // first round, good to 85 bits
    const int j = ix >> 20;
    y0 = r - w;
    int i = j - (((int32_t)(__builtin_bit_cast(int64_t, y0) >> 32) >> 20) & 0x7ff);
    if (i > 16) {  // cancellation: second round, good to 118 bits
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y0 = r - w;
        i = j - (((int32_t)(__builtin_bit_cast(int64_t, y0) >> 32) >> 20) & 0x7ff);
        if (i > 49) {  // third round, 151 bits
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            y0 = r - w;
        }
    }
    y1 = (r - y0) - w;
    if (hx < 0) {
        y0 = -y0;
        y1 = -y1;
        return (-n) & 3;
    }
    return n & 3;
}
PG_DEV double pg_sin_d(double x) {
    if (!(__builtin_fabs(x) <= 1.7976931348623157e308)) return x - x;  // inf / NaN
    double y0, y1;
    const int n = pg_rem_pio2_d(x, y0, y1);
    const bool tail = !(y1 == 0 && y0 == x);
    if (n == 0) return pg_ksin_d(y0, y1, tail);
    if (n == 1) return pg_kcos_d(y0, y1);
    if (n == 2) return -pg_ksin_d(y0, y1, true);
    return -pg_kcos_d(y0, y1);
}
PG_DEV double pg_cos_d(double x) {
    if (!(__builtin_fabs(x) <= 1.7976931348623157e308)) return x - x;
    double y0, y1;
    const int n = pg_rem_pio2_d(x, y0, y1);
    if (n == 0) return pg_kcos_d(y0, y1);
    if (n == 1) return -pg_ksin_d(y0, y1, true);
    if (n == 2) return -pg_kcos_d(y0, y1);
    return pg_ksin_d(y0, y1, true);
}

}  // namespace pgamd
