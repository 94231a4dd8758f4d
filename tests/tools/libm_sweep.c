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

/* caveflyer's exhaust puff (reference src/games/caveflyer.cpp:275): add_entity(agent->x - agent->rx * cos(theta), agent->y - agent->ry * sin(theta), ...),
 * i.e. float(double(x) - double(r) * trig(double(theta))) -- serialized state with a position-dependent operand.  The device's sin / cos
 * (dev_sin / dev_cos: doubles of the float angles first_bits ..) and the host libm's may differ in the last bit of the double; everything
 * after the trig call is IEEE double arithmetic and a narrowing, identical on both sides, so the comparison runs here, on the device's
 * doubles.  For every angle where a double differs, and both signs of the angle (sin odd, cos even in both implementations), the
 * expression is evaluated for the agent radius r and for x over (a) `samples` evenly spaced positions of [r, xmax - r], shifted per angle,
 * and (b) the 2 x 32 floats next to the wall contact x = r, where the difference x - r trig is smallest.  counts[0] += angles examined,
 * counts[1] += products r * trig that differ as doubles, counts[2] += evaluations, counts[3] += evaluations whose float results differ. */
void count_puff_mismatches(const double *dev_sin, const double *dev_cos, uint32_t first_bits, long n, float r, float xmax, int samples, long *counts) {
    long na = 0, np = 0, ne = 0, nbad = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : na, np, ne, nbad)
    for (long i = 0; i < n; i++) {
        uint32_t b = first_bits + (uint32_t)i;
        float af;
        memcpy(&af, &b, 4);
        const double a = (double)af, hs = sin(a), hc = cos(a);
        if (memcmp(&hs, &dev_sin[i], 8) == 0 && memcmp(&hc, &dev_cos[i], 8) == 0) continue;
        na++;
        for (int comp = 0; comp < 2; comp++) {
            for (int sign = 0; sign < 2; sign++) {
                /* trig(-a): sin changes sign, cos does not */
                const double th = comp == 0 ? (sign ? -hs : hs) : hc, td = comp == 0 ? (sign ? -dev_sin[i] : dev_sin[i]) : dev_cos[i];
                if (comp == 1 && sign == 1) continue;
                const double ph = (double)r * th, pd = (double)r * td;
                if (memcmp(&ph, &pd, 8) == 0) continue;
                np++;
                const float lo = r, hi = xmax - r;
                const float step = (hi - lo) / (float)samples;
                const float shift = step * (float)((b * 2654435761u) >> 8) / 16777216.0f;
                for (int k = 0; k < samples; k++) {
                    const float x = lo + shift + step * (float)k;
                    ne++;
                    if ((float)((double)x - ph) != (float)((double)x - pd)) nbad++;
                }
                uint32_t rb;
                memcpy(&rb, &r, 4);
                for (int k = -32; k <= 32; k++) {
                    const uint32_t xb = rb + (uint32_t)k;
                    float x;
                    memcpy(&x, &xb, 4);
                    ne++;
                    if ((float)((double)x - ph) != (float)((double)x - pd)) nbad++;
                }
            }
        }
    }
    counts[0] += na; counts[1] += np; counts[2] += ne; counts[3] += nbad;
}
