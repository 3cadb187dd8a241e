/*
 * ptx_output.c -- what the user sees: <name>.pfm with the raw accumulator and <name> as a tone-mapped 8-bit P6.
 * Follows dataOutput() of the reference (utils_path.c:1989-2076) step for step, because the cut-off selection is
 * data dependent and the PPM is the product: global min/max normalisation, a 1000-bin histogram, a low and a high
 * cut found by scanning it, clip, gamma 0.75, truncate to bytes.  Runs on the host between render calls.
 */
#include "ptx_kit.h"

/* The two files of dataOutput() from an accumulator and bytes that were tone-mapped elsewhere (on the GPU, by
 * ptx_tonemap): <name>.pfm and <name>, with the two lines the reference prints in between. */
void ptxh_write_output(const double *im, const unsigned char *bits, int sx, const char *name, double mi, double mx,
                       double low_cut, double high_cut) {
    const size_t n = (size_t)sx * sx * 3;
    char pfmname[1100];
    snprintf(pfmname, sizeof(pfmname), "%s.pfm", name);
    FILE *f = fopen(pfmname, "w");
    if (f) {
        fprintf(f, "PF\n%d %d\n%1.1f\n", sx, sx, -1.0);
        fwrite(im, n * sizeof(double), 1, f);
        fclose(f);
    }
    fprintf(stderr, "Image stats: Minimum=%f, maximum=%f\n", mi, mx);
    fprintf(stderr, "Limit values chosen at min=%f, max=%f... normalizing image\n", low_cut, high_cut);
    f = fopen(name, "wb+");
    if (!f) {
        fprintf(stderr, "Unable to open file %s for output! No image written\n", name);
    } else {
        fprintf(f, "P6\n# Output from PathTracer.c\n%d %d\n255\n", sx, sx);
        fwrite(bits, n, 1, f);
        fclose(f);
    }
}

void dataOutput(double *im, int sx, char *name) {
    const size_t n = (size_t)sx * sx * 3;
    double *t = (double *)malloc(n * sizeof(double));
    unsigned char *bits = (unsigned char *)malloc(n);
    char pfmname[1100];
    if (!t || !bits) { free(t); free(bits); return; }
    memcpy(t, im, n * sizeof(double));

    /* raw doubles under a PF header: not a standard float PFM, but what the reference writes (:2003-2009) */
    snprintf(pfmname, sizeof(pfmname), "%s.pfm", name);
    FILE *f = fopen(pfmname, "w");
    if (f) {
        fprintf(f, "PF\n%d %d\n%1.1f\n", sx, sx, -1.0);
        fwrite(t, n * sizeof(double), 1, f);
        fclose(f);
    }

    double mi = 10e6, mx = -10e6;
    for (size_t i = 0; i < n; i++) {
        if (t[i] < mi) mi = t[i];
        if (t[i] > mx) mx = t[i];
    }
    for (size_t i = 0; i < n; i++) {
        t[i] = t[i] - mi;
        t[i] = t[i] / (mx - mi);
    }
    fprintf(stderr, "Image stats: Minimum=%f, maximum=%f\n", mi, mx);

    /* histogram with bins [biw*j, biw*(j+1)); the reference finds the bin by a linear scan with exactly these
     * comparisons, so the bin is searched the same way starting from the arithmetic guess */
    const double biw = 1.000001 / 1000.0;
    double hist[1000];
    for (int j = 0; j < 1000; j++) hist[j] = 0;
    for (size_t i = 0; i < n; i++) {
        double v = t[i];
        if (!(v >= 0)) continue; /* NaN or negative: no bin, as in the reference's scan */
        int j = (int)(v / biw);
        if (j > 999) j = 999;
        while (j > 0 && !(v >= biw * j)) j--;
        while (j < 999 && !(v < biw * (j + 1))) j++;
        if (v >= biw * j && v < biw * (j + 1)) hist[j]++;
    }

    const double pct = .005 * (double)((long)sx * sx * 3);
    int j;
    mx = 0;
    for (j = 5; j < 990; j++) {
        mx += hist[j];
        if (hist[j + 5] - hist[j - 5] > pct) break;
        if (mx > pct) break;
    }
    mi = (biw * (.90 * j));
    for (j = 990; j > 5; j--)
        if (hist[j - 5] - hist[j + 5] > pct) break;
    mx = (biw * (j + (.25 * (999 - j))));
    fprintf(stderr, "Limit values chosen at min=%f, max=%f... normalizing image\n", mi, mx);

    for (size_t i = 0; i < n; i++) {
        double v = t[i] - mi;
        v = v / (mx - mi);
        if (v < 0.0) v = 0.0;
        if (v > 1.0) v = 1.0;
        v = pow(v, .75);
        bits[i] = (unsigned char)(255.0 * v);
    }
    f = fopen(name, "wb+");
    if (!f) {
        fprintf(stderr, "Unable to open file %s for output! No image written\n", name);
    } else {
        fprintf(f, "P6\n# Output from PathTracer.c\n%d %d\n255\n", sx, sx);
        fwrite(bits, n, 1, f);
        fclose(f);
    }
    free(bits);
    free(t);
}
