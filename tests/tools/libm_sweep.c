/* TEST HARNESS: host-libm (glibc) side of the device math sweeps (tests/test_device_math.py, -m gpu). */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* the reference's expression, reference src/games/bigfish.cpp:84 (FISH_MAX_R = 2, FISH_MIN_R = .25) */
void ref_bigfish_radius(const float *r01, float *out, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) out[i] = (float)((double)(2.0f - .25f) * pow((double)r01[i], 1.4) + (double).25f);
}

/* counts[0..3] += mismatches of sin / cos as doubles, and after narrowing to float, for the floats with bit patterns
 * first_bits .. first_bits + n - 1 */
void count_sincos_mismatches(const double *dev_sin, const double *dev_cos, uint32_t first_bits, long n, long *counts) {
    long ds = 0, dc = 0, fs = 0, fc = 0;
#pragma omp parallel for schedule(static) reduction(+ : ds, dc, fs, fc)
    for (long i = 0; i < n; i++) {
        uint32_t b = first_bits + (uint32_t)i;
        float xf;
        memcpy(&xf, &b, 4);
        const double x = (double)xf, s = sin(x), c = cos(x);
        if (memcmp(&s, &dev_sin[i], 8) != 0) ds++;
        if (memcmp(&c, &dev_cos[i], 8) != 0) dc++;
        if ((float)s != (float)dev_sin[i]) fs++;
        if ((float)c != (float)dev_cos[i]) fc++;
    }
    counts[0] += ds; counts[1] += dc; counts[2] += fs; counts[3] += fc;
}

/* appends the bit patterns (of the same range) whose sin or cos differ as doubles to out[*count ...] (capacity cap) */
void collect_sincos_mismatches(const double *dev_sin, const double *dev_cos, uint32_t first_bits, long n, uint32_t *out, long cap, long *count) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        uint32_t b = first_bits + (uint32_t)i;
        float xf;
        memcpy(&xf, &b, 4);
        const double x = (double)xf, s = sin(x), c = cos(x);
        if (memcmp(&s, &dev_sin[i], 8) != 0 || memcmp(&c, &dev_cos[i], 8) != 0) {
            long k;
#pragma omp atomic capture
            k = (*count)++;
            if (k < cap) out[k] = b;
        }
    }
}
/* host side of sincos_scaled */
void ref_sincos_scaled(const uint32_t *bits, long n, double scale, float *out_sin, float *out_cos) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        float xf;
        memcpy(&xf, &bits[i], 4);
        out_sin[i] = (float)(sin((double)xf) * scale);
        out_cos[i] = (float)(cos((double)xf) * scale);
    }
}
