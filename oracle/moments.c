/* TEST INFRASTRUCTURE ONLY -- CPU restatement of cv::moments for single-channel images of depth CV_8U / CV_16U / CV_16S (moments.cpp:309-357
 * momentsInTile, :483-575 the tile loop of cv::moments), the part cv_hal_imageMoments replaces: the ten spatial moments m00 .. m03.  Never linked into the product.
 *
 * The image is cut into 32 x 32 tiles; inside a tile the raw moments are exact integers (int for CV_8U, int64 for the 16-bit depths); they are converted to
 * double, shifted to the tile's origin (x, y) with the reference's expressions -- whose grouping of the double operations is kept, because the sums over the
 * tiles exceed 2^53 for large images and round -- and added tile by tile in raster order.  `binary`: the tile is first replaced by (pixel != 0) ? 255 : 0 and
 * its moments are scaled by the double 1. / 255. */
#include "oracle.h"
#include <stdint.h>

int orc_imageMoments(const uint8_t* src, size_t sstep, int depth /*0 8U, 2 16U, 3 16S*/, int w, int h, int binary, double* m)
{
    if ((depth != 0 && depth != 2 && depth != 3) || w <= 0 || h <= 0) return 1;
    for (int k = 0; k < 10; k++) m[k] = 0;
    for (int y = 0; y < h; y += 32) {
        const int th = h - y < 32 ? h - y : 32;
        for (int x = 0; x < w; x += 32) {
            const int tw = w - x < 32 ? w - x : 32;
            int64_t mom[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < th; r++) {
                const uint8_t* row = src + (size_t)(y + r) * sstep;
                int64_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
                for (int c = 0; c < tw; c++) {
                    int64_t p = depth == 0 ? row[x + c] : depth == 2 ? ((const uint16_t*)row)[x + c] : ((const int16_t*)row)[x + c];
                    if (binary) p = p != 0 ? 255 : 0;
                    x0 += p; x1 += c * p; x2 += (int64_t)c * c * p; x3 += (int64_t)c * c * c * p;
                }
                const int64_t py = r * x0, sy = (int64_t)r * r;
                mom[9] += py * sy; mom[8] += x1 * sy; mom[7] += x2 * r; mom[6] += x3; mom[5] += x0 * sy;
                mom[4] += x1 * r; mom[3] += x2; mom[2] += py; mom[1] += x1; mom[0] += x0;
            }
            double mo[10];
            for (int k = 0; k < 10; k++) mo[k] = (double)mom[k];
            if (binary) { const double s = 1. / 255; for (int k = 0; k < 10; k++) mo[k] *= s; }
            const double xm = x * mo[0], ym = y * mo[0];
            m[0] += mo[0];
            m[1] += mo[1] + xm;
            m[2] += mo[2] + ym;
            m[3] += mo[3] + x * (mo[1] * 2 + xm);
            m[4] += mo[4] + x * (mo[2] + ym) + y * mo[1];
            m[5] += mo[5] + y * (mo[2] * 2 + ym);
            m[6] += mo[6] + x * (3. * mo[3] + x * (3. * mo[1] + xm));
            m[7] += mo[7] + x * (2 * (mo[4] + y * mo[1]) + x * (mo[2] + ym)) + y * mo[3];
            m[8] += mo[8] + y * (2 * (mo[4] + x * mo[2]) + y * (mo[1] + xm)) + x * mo[5];
            m[9] += mo[9] + y * (3. * mo[5] + y * (3. * mo[2] + ym));
        }
    }
    return 0;
}

/* CV_32F / CV_64F images: momentsInTile<float, double, double> / <double, double, double> (moments.cpp:307-357, no vector form) -- every sum is a chain of
 * double additions in raster order inside the 32 x 32 tile, rows first, then the ten moments row by row; the tile loop is the one above. */
int orc_imageMomentsF(const uint8_t* src, size_t sstep, int depth /*5 32F, 6 64F*/, int w, int h, int binary, double* m)
{
    if ((depth != 5 && depth != 6) || w <= 0 || h <= 0) return 1;
    for (int k = 0; k < 10; k++) m[k] = 0;
    for (int y = 0; y < h; y += 32) {
        const int th = h - y < 32 ? h - y : 32;
        for (int x = 0; x < w; x += 32) {
            const int tw = w - x < 32 ? w - x : 32;
            double mom[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < th; r++) {
                const uint8_t* row = src + (size_t)(y + r) * sstep;
                double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
                for (int c = 0; c < tw; c++) {
                    double p = depth == 5 ? (double)((const float*)row)[x + c] : ((const double*)row)[x + c];
                    if (binary) p = p != 0 ? 255 : 0;                       /* the tile is first compared with zero into a CV_8U tile (:523-528) */
                    const double xp = c * p, xxp = xp * c;
                    x0 += p; x1 += xp; x2 += xxp; x3 += xxp * c;
                }
                const double py = r * x0, sy = (double)(r * r);
                mom[9] += py * sy; mom[8] += x1 * sy; mom[7] += x2 * r; mom[6] += x3; mom[5] += x0 * sy;
                mom[4] += x1 * r; mom[3] += x2; mom[2] += py; mom[1] += x1; mom[0] += x0;
            }
            double mo[10];
            for (int k = 0; k < 10; k++) mo[k] = mom[k];
            if (binary) { const double s = 1. / 255; for (int k = 0; k < 10; k++) mo[k] *= s; }
            const double xm = x * mo[0], ym = y * mo[0];
            m[0] += mo[0];
            m[1] += mo[1] + xm;
            m[2] += mo[2] + ym;
            m[3] += mo[3] + x * (mo[1] * 2 + xm);
            m[4] += mo[4] + x * (mo[2] + ym) + y * mo[1];
            m[5] += mo[5] + y * (mo[2] * 2 + ym);
            m[6] += mo[6] + x * (3. * mo[3] + x * (3. * mo[1] + xm));
            m[7] += mo[7] + x * (2 * (mo[4] + y * mo[1]) + x * (mo[2] + ym)) + y * mo[3];
            m[8] += mo[8] + y * (2 * (mo[4] + x * mo[2]) + y * (mo[1] + xm)) + x * mo[5];
            m[9] += mo[9] + y * (3. * mo[5] + y * (3. * mo[2] + ym));
        }
    }
    return 0;
}
