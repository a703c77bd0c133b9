/* oracle/deblock_oracle.c — CPU restatement of the reference's block-boundary filter.
 * TEST INFRASTRUCTURE ONLY (see siren_oracle.h).
 *   mode 1: deblock.py arithmetic (Python floats, true division)      deblock.py:7-50, 52-78
 *   mode 0: deblock.cpp arithmetic (C ints, truncating division)      deblock.cpp:12-71, 277-319
 * The Python flavour is pinned by golden vectors produced by running deblock.py; deblock.cpp needs
 * libtiff headers that this image lacks, so its flavour is restated from the source only. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static double alpha_(double x) { return 0.8 * (pow(2.0, x / 6.0) - 1.0); }
static double beta_(double x) { return 0.5 * x - 7.0; }
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void filter_px(uint16_t *p, int64_t stride, double index_a, double index_b, double thres, int mode)
{   /* p points at q0; pixels p2,p1,p0,q0,q1,q2 are p[-3*stride] .. p[2*stride] */
    const int P2 = p[-3 * stride], P1 = p[-2 * stride], P0 = p[-stride], Q0 = p[0], Q1 = p[stride], Q2 = p[2 * stride];
    if (mode == 1) {
        const double p2 = P2, q2 = Q2;
        double p1 = P1, p0 = P0, q0 = Q0, q1 = Q1;
        if ((p1 + p0 + q0 + q1) / 4 > thres) return;
        if (!(fabs(p0 - q0) < alpha_(index_a) && fabs(p1 - p0) < beta_(index_b) && fabs(q1 - q0) < beta_(index_b))) return;
        double delta0 = (4 * (q0 - p0) + (p1 - q1) + 4) / 8;
        double deltap1 = (p2 + (p0 + q0 + 1) / 2 - 2 * p1) / 2;
        double deltaq1 = (q2 + (q0 + p0 + 1) / 2 - 2 * q1) / 2;
        double c1 = 20, c0 = c1;
        if (fabs(p2 - p0) < beta_(index_b)) c0 += 1;
        if (fabs(q2 - q0) < beta_(index_b)) c0 += 1;
        delta0 = clipd(delta0, -c0, c0);
        deltap1 = clipd(deltap1, -c1, c1);
        deltaq1 = clipd(deltaq1, -c1, c1);
        p1 += deltap1; p0 += delta0; q0 -= delta0; q1 += deltaq1;
        p[-2 * stride] = (uint16_t)(int64_t)p1; p[-stride] = (uint16_t)(int64_t)p0;
        p[0] = (uint16_t)(int64_t)q0; p[stride] = (uint16_t)(int64_t)q1;
    } else {
        if ((P1 + P0 + Q0 + Q1) / 4 > (int)thres) return;
        const float al = (float)(0.8 * (pow(2.0, (double)(float)index_a / 6) - 1)), be = (float)(0.5 * (double)(float)index_b - 7);
        if (!((float)abs(P0 - Q0) < al && (float)abs(P1 - P0) < be && (float)abs(Q1 - Q0) < be)) return;
        float delta0 = (float)((4 * (Q0 - P0) + (P1 - Q1) + 4) / 8);
        float deltap1 = (float)((P2 + (P0 + Q0 + 1) / 2 - 2 * P1) / 2);
        float deltaq1 = (float)((Q2 + (Q0 + P0 + 1) / 2 - 2 * Q1) / 2);
        uint16_t c1 = 20, c0 = c1;
        if ((float)abs(P2 - P0) < be) c0 += 1;
        if ((float)abs(Q2 - Q0) < be) c0 += 1;
        delta0 = delta0 < -(float)c0 ? -(float)c0 : (delta0 > (float)c0 ? (float)c0 : delta0);
        deltap1 = deltap1 < -(float)c1 ? -(float)c1 : (deltap1 > (float)c1 ? (float)c1 : deltap1);
        deltaq1 = deltaq1 < -(float)c1 ? -(float)c1 : (deltaq1 > (float)c1 ? (float)c1 : deltaq1);
        p[-2 * stride] = (uint16_t)((float)P1 + deltap1); p[-stride] = (uint16_t)((float)P0 + delta0);
        p[0] = (uint16_t)((float)Q0 - delta0); p[stride] = (uint16_t)((float)Q1 + deltaq1);
    }
}

/* one boundary line of one slice z, exactly as filter2d / the C++ inner loops walk it */
void oracle_deblock_line(uint16_t *img, int64_t D, int64_t H, int64_t W, int64_t z, int64_t x1, int64_t y1, int64_t x2, int64_t y2,
                         double index_a, double index_b, double thres, int mode)
{
    (void)D;
    uint16_t *sl = img + z * H * W;
    if (x1 == x2) {
        if (x1 - 3 < 0 || x1 + 3 > W - 1) return;
        for (int64_t y = y1; y <= y2; ++y) filter_px(sl + y * W + x1, 1, index_a, index_b, thres, mode);
    } else if (y1 == y2) {
        if (y1 - 3 < 0 || y1 + 3 > H - 1) return;
        for (int64_t x = x1; x <= x2; ++x) filter_px(sl + y1 * W + x, W, index_a, index_b, thres, mode);
    }
}
