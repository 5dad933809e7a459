/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see slic_oracle.c header).
 *
 * orc_resize_bicubic_u8: the input stage's resize, i.e. what
 *   datasets/resize_image_dataset.py:31-34  chainercv.transforms.resize(image, shape, 3)
 * computes on an 8-bit image when Pillow does the work:
 *   PIL.Image.fromarray(channel).resize((w, h), PIL.Image.BICUBIC)      (mode "L", one channel at a time)
 * Pillow is a third-party dependency absent from /root/reference (README.md pins no version; 8.4.0 is in
 * the conda environment, 12.2.0 in the default one: identical outputs on the fixtures).  Restated from
 * the published algorithm of Pillow's src/libImaging/Resample.c [3p]:
 *   - coefficients (precompute_coeffs): scale = in/out; filterscale = max(scale, 1); support = 2 *
 *     filterscale; for every output position xx: center = (xx + 0.5) * scale, taps xmin = max(0,
 *     (int)(center - support + 0.5)) .. xmax = min(in, (int)(center + support + 0.5)), weight w =
 *     bicubic((x + xmin - center + 0.5) / filterscale) with a = -0.5, normalised by their sum;
 *   - 8-bit path (normalize_coeffs_8bpc): k = (int)(+-0.5 + w * 2^22);
 *   - horizontal pass over every input row, then vertical pass, each  clip8((2^21 + sum pixel * k) >> 22).
 * chainercv's Pillow branch would pass mode "F", which raises for the uint8 image the reference hands it,
 * so the reference itself ran its OpenCV branch (cv2.INTER_CUBIC): that variant cannot be pinned here
 * (no cv2 in the image, no fixture in the reference) — DESIGN.md section 7.2.
 * Pinned by tests/golden/resize_*.npz (outputs of Pillow 8.4.0 itself).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double bicubic_filter(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

/* returns ksize; bounds (out*2) and integer coefficients (out*ksize) are malloc'ed */
int orc_resize_coeffs(int in_size, int out_size, int32_t **bounds_out, int32_t **kk_out)
{
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    int32_t *bounds = (int32_t *)malloc((size_t)out_size * 2 * sizeof(int32_t));
    int32_t *kk = (int32_t *)malloc((size_t)out_size * ksize * sizeof(int32_t));
    double *k = (double *)malloc((size_t)ksize * sizeof(double));
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0;
        for (x = 0; x < ksize; ++x) {
            if (k[x] < 0) kk[xx * ksize + x] = (int)(-0.5 + k[x] * (1 << 22));
            else kk[xx * ksize + x] = (int)(0.5 + k[x] * (1 << 22));
        }
        bounds[xx * 2 + 0] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    free(k);
    *bounds_out = bounds;
    *kk_out = kk;
    return ksize;
}

static inline uint8_t clip8(int v)
{
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* src (H, W) uint8 one channel -> dst (h, w) uint8 */
void orc_resize_bicubic_u8(const uint8_t *src, int H, int W, uint8_t *dst, int h, int w)
{
    int32_t *bx, *kx, *by, *ky;
    const int ksx = orc_resize_coeffs(W, w, &bx, &kx);
    const int ksy = orc_resize_coeffs(H, h, &by, &ky);
    uint8_t *tmp = (uint8_t *)malloc((size_t)H * w);
    const int need_h = (W != w), need_v = (H != h);
    /* Pillow skips a pass whose size does not change */
    if (need_h) {
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < w; ++xx) {
                const int xmin = bx[xx * 2], xmax = bx[xx * 2 + 1];
                int ss = 1 << 21;
                for (int x = 0; x < xmax; ++x) ss += src[(size_t)yy * W + x + xmin] * kx[xx * ksx + x];
                tmp[(size_t)yy * w + xx] = clip8(ss);
            }
    } else memcpy(tmp, src, (size_t)H * W);
    if (need_v) {
        for (int yy = 0; yy < h; ++yy) {
            const int ymin = by[yy * 2], ymax = by[yy * 2 + 1];
            for (int xx = 0; xx < w; ++xx) {
                int ss = 1 << 21;
                for (int y = 0; y < ymax; ++y) ss += tmp[(size_t)(y + ymin) * w + xx] * ky[yy * ksy + y];
                dst[(size_t)yy * w + xx] = clip8(ss);
            }
        }
    } else memcpy(dst, tmp, (size_t)h * w);
    free(tmp); free(bx); free(kx); free(by); free(ky);
}
