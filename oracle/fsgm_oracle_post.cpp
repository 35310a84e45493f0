/*
 * fsgm_oracle_post.cpp -- CPU restatement of the reference's post-processing functions, the MATLAB
 * files the evaluation script chains after SGM (test.m:45-50): speckle_filter.m, calc_disp_from_first.m,
 * forward_backward_check.m, scanline_in_fill.m, vzInd2Disp.m  (reference: /root/reference, cited as
 * file:line).  Written in the reference's own order of operations: raster scans, a FIFO flood fill,
 * first-come writes.
 *
 * TEST INFRASTRUCTURE ONLY (see fsgm_oracle.h).  PARITY UNPINNED: the originals are MATLAB scripts
 * (no MATLAB/Octave in this image) and the reference ships no input/output pair for them.
 *
 * Memory order: maps f64 [y][x] (x fastest), two-plane maps [plane][y][x]; MATLAB's NaN = invalid
 * carries over as an IEEE NaN.  Pixel coordinates inside the maps (Pd0) are 1-based like MATLAB's.
 */
#include "fsgm_oracle.h"
#include <math.h>
#include <string.h>
#include <vector>

extern "C" {

/* vzInd2Disp.m:1-5 */
void fsgm_oracle_vzind2disp(double* D, const double* w, const double* O, int n_px, double vMax, double n) {
    for (int i = 0; i < n_px; i++) {
        const double vzRatio = w[i] / n * vMax;
        const double vzInd = vzRatio / (1 - vzRatio);
        D[i] = O[i] * vzInd;
    }
}

/* speckle_filter.m:1-103.  labels (may be NULL): i32 [H][W], 0 where the input is NaN. */
void fsgm_oracle_speckle_filter(double* out, int32_t* labels_out, const double* image, int W, int H,
                                double maxDiff, double maxSpeckleSize) {
    const size_t NP = (size_t)W * H;
    std::vector<double> img(image, image + NP);
    std::vector<int32_t> labels(NP, 0);                            /* :18 */
    std::vector<char> regionTypes(1, 0);                           /* :20, indexed by label-1 */
    std::vector<int32_t> queue(NP);
    int32_t curLabel = 0;
    for (int y = 0; y < H; y++)                                    /* :23 */
        for (int x = 0; x < W; x++) {
            const size_t ind = (size_t)y * W + x;                  /* :25 sub2ind([width,height], x, y) */
            if (isnan(img[ind])) continue;                         /* :26 */
            if (labels[ind] > 0) {                                 /* :27-30 */
                if (regionTypes[labels[ind] - 1]) img[ind] = NAN;
                continue;
            }
            size_t head = 0, tail = 0;
            queue[tail++] = (int32_t)ind;                          /* :35 */
            curLabel++;                                            /* :37 */
            if ((size_t)curLabel > regionTypes.size()) regionTypes.resize(curLabel, 0);
            regionTypes[curLabel - 1] = 0;                         /* :38 */
            labels[ind] = curLabel;                                /* :40 */
            long regionPixelNum = 0;
            while (head < tail) {                                  /* :43 */
                const int32_t cur = queue[head++];                 /* :45 */
                const int cury = cur / W, curx = cur - cury * W;
                regionPixelNum++;
                const double v = img[cur];                         /* :50 */
                auto visit = [&](int32_t nb) {                     /* :53-91 */
                    if (labels[nb] == 0 && !isnan(img[nb]) && fabs(v - img[nb]) < maxDiff) {
                        labels[nb] = curLabel;
                        queue[tail++] = nb;
                    }
                };
                if (curx < W - 1) visit(cur + 1);                  /* right  :53 */
                if (curx > 0) visit(cur - 1);                      /* left   :63 */
                if (cury < H - 1) visit(cur + W);                  /* bottom :73 */
                if (cury > 0) visit(cur - W);                      /* top    :83 */
            }
            if ((double)regionPixelNum < maxSpeckleSize) {         /* :94-97 */
                regionTypes[curLabel - 1] = 1;
                img[ind] = NAN;
            }
        }
    memcpy(out, img.data(), NP * sizeof(double));
    if (labels_out) memcpy(labels_out, labels.data(), NP * sizeof(int32_t));     /* :101 */
}

/* calc_disp_from_first.m:1-52.  Pd0, nd: [2][H][W], plane 0 = x; O: [H][W]. */
void fsgm_oracle_calc_disp_from_first(double* D2, const double* D1, int W, int H, const double* Pd0,
                                      const double* nd, const double* O, double vMax, double n) {
    const size_t NP = (size_t)W * H;
    for (size_t i = 0; i < NP; i++) D2[i] = -1.0;                  /* :6 */
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            const size_t p = (size_t)j * W + i;
            const double vzInd = D1[p];
            double disp;
            fsgm_oracle_vzind2disp(&disp, &vzInd, &O[p], 1, vMax, n);          /* :11 */
            const double p2x = Pd0[p] + disp * nd[p], p2y = Pd0[NP + p] + disp * nd[NP + p];   /* :13-14 */
            const double sx0 = floor(p2x), sy0 = floor(p2y), sx1 = sx0 + 1, sy1 = sy0 + 1;      /* :16-22 */
            auto put = [&](double sx, double sy) {                 /* :24-46 */
                if (sx >= 1 && sx <= W && sy >= 1 && sy <= H) {
                    double& t = D2[(size_t)((int)sy - 1) * W + ((int)sx - 1)];
                    if (t == 0 || t < D1[p]) t = D1[p];
                }
            };
            put(sx0, sy0); put(sx1, sy0); put(sx0, sy1); put(sx1, sy1);
        }
}

/* forward_backward_check.m:1-39 */
void fsgm_oracle_forward_backward_check(double* out, const double* D1, const double* D2, int W, int H,
                                        const double* Pd0, const double* nd, const double* O, double vMax, double n) {
    const size_t NP = (size_t)W * H;
    const double thr = 2.0;                                        /* :6 */
    memcpy(out, D1, NP * sizeof(double));
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            const size_t p = (size_t)j * W + i;
            const double vzInd = out[p];
            if (isnan(vzInd)) continue;                            /* :12-14 */
            double disp;
            fsgm_oracle_vzind2disp(&disp, &vzInd, &O[p], 1, vMax, n);          /* :15 */
            const double p2x = round(Pd0[p] + disp * nd[p]), p2y = round(Pd0[NP + p] + disp * nd[NP + p]);   /* :17-20 */
            /* :22-25; a NaN target (NaN geometry) passes every test of :22 and then indexes D2(NaN,NaN), an
             * error in MATLAB: treated as outside here and in the kernel */
            if (!(p2x >= 1 && p2x <= W && p2y >= 1 && p2y <= H)) { out[p] = NAN; continue; }
            const double d2 = D2[(size_t)((int)p2y - 1) * W + ((int)p2x - 1)];
            if (d2 == -1) { out[p] = NAN; continue; }              /* :27-30 */
            if (fabs(out[p] - d2) > thr) out[p] = NAN;             /* :32-34 */
        }
}

/* scanline_in_fill.m:1-70, one channel (test.m:49 passes a 2-D map) */
void fsgm_oracle_scanline_in_fill(double* out, const double* in, int W, int H) {
    const size_t NP = (size_t)W * H;
    memcpy(out, in, NP * sizeof(double));
    auto at = [&](int v, int u) -> double& { return out[(size_t)v * W + u]; };
    for (int v = 0; v < H; v++) {                                  /* :6 */
        int count = 0;
        for (int u = 0; u < W; u++) {                              /* :9 */
            if (!isnan(at(v, u))) {
                if (count >= 1) {                                  /* :11-22 */
                    const int u1 = u - count, u2 = u - 1;          /* 0-based */
                    if (u1 > 0 && u2 < W - 1) {                    /* :14: u1 > 1 && u2 < width (1-based) */
                        const double f = fmin(at(v, u1 - 1), at(v, u2 + 1));
                        for (int c = u1; c <= u2; c++) at(v, c) = f;
                    }
                }
                count = 0;
            } else count++;
        }
        for (int u = 0; u < W; u++)                                /* :30-37 extrapolate to the left */
            if (!isnan(at(v, u))) { for (int u2 = 0; u2 < u; u2++) at(v, u2) = at(v, u); break; }
        for (int u = W - 1; u >= 0; u--)                           /* :39-46 to the right */
            if (!isnan(at(v, u))) { for (int u2 = u + 1; u2 < W; u2++) at(v, u2) = at(v, u); break; }
    }
    for (int u = 0; u < W; u++) {                                  /* :50 */
        for (int v = 0; v < H; v++)                                /* :52-59 to the top */
            if (!isnan(at(v, u))) { for (int v2 = 0; v2 < v; v2++) at(v2, u) = at(v, u); break; }
        for (int v = H - 1; v >= 0; v--)                           /* :61-68 to the bottom */
            if (!isnan(at(v, u))) { for (int v2 = v + 1; v2 < H; v2++) at(v2, u) = at(v, u); break; }
    }
}

/* vmf.m:1-14: medfilt2(flow(:,:,c), [5 5]) per channel.  medfilt2 (toolbox, not in the reference tree)
 * pads with zeros and returns the median of the 25 window values: the 13th smallest. */
void fsgm_oracle_vmf(double* out, const double* flow, int W, int H, int channels) {
    const size_t NP = (size_t)W * H;
    for (int c = 0; c < channels; c++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double w[25];
                int k = 0;
                for (int dy = -2; dy <= 2; dy++)
                    for (int dx = -2; dx <= 2; dx++) {
                        const int yy = y + dy, xx = x + dx;
                        w[k++] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? flow[c * NP + (size_t)yy * W + xx] : 0.0;
                    }
                for (int i = 1; i < 25; i++) {                         /* insertion sort */
                    const double v = w[i];
                    int j = i - 1;
                    while (j >= 0 && w[j] > v) { w[j + 1] = w[j]; j--; }
                    w[j + 1] = v;
                }
                out[c * NP + (size_t)y * W + x] = w[12];
            }
}

/* test.m:45-50: the chain the evaluation script runs on the vz-index map D1.  filterD2 (may be NULL)
 * receives calc_disp_from_first's map; disp (may be NULL) receives vzInd2Disp of the result. */
void fsgm_oracle_postprocess(double* filterD1, double* filterD2, double* disp, const double* D1, int W, int H,
                             const double* Pd0, const double* nd, const double* O, double vMax, double n, double dMax) {
    const size_t NP = (size_t)W * H;
    std::vector<double> a(NP), b(NP), d2(NP);
    fsgm_oracle_speckle_filter(a.data(), nullptr, D1, W, H, 2, 100);                                   /* :45 */
    fsgm_oracle_calc_disp_from_first(d2.data(), a.data(), W, H, Pd0, nd, O, vMax, n);                  /* :46 */
    fsgm_oracle_forward_backward_check(b.data(), a.data(), d2.data(), W, H, Pd0, nd, O, vMax, n);      /* :47 */
    fsgm_oracle_speckle_filter(a.data(), nullptr, b.data(), W, H, dMax, (double)H * (double)W / 10);    /* :48 rows*cols/10 */
    fsgm_oracle_scanline_in_fill(filterD1, a.data(), W, H);                                            /* :49 */
    if (filterD2) memcpy(filterD2, d2.data(), NP * sizeof(double));
    if (disp) fsgm_oracle_vzind2disp(disp, filterD1, O, (int)NP, vMax, n);                             /* :50 */
}

}  // extern "C"
