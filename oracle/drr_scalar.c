/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Independent float64 scalar restatement of the two DRR renderers, one ray at a time, written
 * from the published algorithms (Siddon 1985; DiffDRR's trilinear ray-marching) and the call-site
 * contract of the reference:
 *   renderer(volume[D0,D1,D2], source[B,1,3], target[B,n,3], img[B,1,n], mask=None|[D0,D1,D2])
 *     -> [B,C,n]                      /root/reference/src/xvr/model/trainer.py:283-289
 *   C = max(label)+1, channel = label of the sample's nearest voxel
 *                                     /root/reference/src/xvr/model/trainer.py:288,292-302
 *
 * PARITY UNPINNED: the arithmetic it restates lives in diffdrr==0.6.0
 * (/root/reference/uv.lock:955-977), which is absent from /root/reference and from this image;
 * the reference has no tests or golden vectors for the path.  This file pins the *build's own*
 * torch restatement (oracle/diffdrr_restated.py) and the HIP kernels against an implementation
 * that shares no code and no op sequence with either: no grid_sample, no incremental traversal --
 * Siddon here literally enumerates every plane crossing, sorts, and looks up segment midpoints.
 *
 * Build: gcc -O2 -fPIC -shared -o oracle/_build/libdrr_scalar.so oracle/drr_scalar.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double voxel_shift;   /* planes of voxel i: [i - shift, i + 1 - shift]                 */
    double eps;           /* added to (target - source)                                       */
    double a[3], b[3];    /* sampling index = a * x + b (grid_sample un-normalisation)        */
    int    n_points;      /* trilinear: samples per ray                                       */
    double near, far;     /* trilinear: linspace(near, far, n_points)                         */
    double denom;         /* trilinear: out = len * sum / denom                               */
    int    clip;          /* trilinear: rescale alphas to [alphamin, alphamax]                */
    int    per_ray_clamp; /* siddon: integrate over [max(amin,0), min(amax,1)] only           */
} oracle_spec;

static double vol_at(const float *vol, const int *S, long i, long j, long k) {
    if (i < 0 || j < 0 || k < 0 || i >= S[0] || j >= S[1] || k >= S[2]) return 0.0;
    return (double)vol[((size_t)i * S[1] + j) * S[2] + k];
}

/* label of the voxel nearest to index-space point p (round half to even, zeros padding -> 0) */
static long nearest_label(const float *mask, const int *S, const double *p) {
    long i = (long)nearbyint(p[0]), j = (long)nearbyint(p[1]), k = (long)nearbyint(p[2]);
    return (long)vol_at(mask, S, i, j, k);
}

static void slab(const double *s, const double *d, const int *S, double shift, double *amin, double *amax) {
    double lo = -INFINITY, hi = INFINITY;
    for (int ax = 0; ax < 3; ++ax) {
        double a0 = (0.0 - shift - s[ax]) / d[ax];
        double a1 = ((double)S[ax] - shift - s[ax]) / d[ax];
        double mn = a0 < a1 ? a0 : a1, mx = a0 < a1 ? a1 : a0;
        if (mn > lo) lo = mn;
        if (mx < hi) hi = mx;
    }
    *amin = lo < 0.0 ? 0.0 : lo;
    *amax = hi > 1.0 ? 1.0 : hi;
}

int oracle_trilinear(const float *vol, const int *S, const double *src, const double *tgt,
                     const double *len, int B, int n, const oracle_spec *sp, const float *mask,
                     int C, double *out) {
    memset(out, 0, sizeof(double) * (size_t)B * C * n);
    for (int bi = 0; bi < B; ++bi) {
        const double *s = src + 3 * (size_t)bi;
        for (int r = 0; r < n; ++r) {
            const double *t = tgt + 3 * ((size_t)bi * n + r);
            double d[3] = {t[0] - s[0] + sp->eps, t[1] - s[1] + sp->eps, t[2] - s[2] + sp->eps};
            double amin = 0, amax = 1;
            slab(s, d, S, sp->voxel_shift, &amin, &amax);
            double span = amax - amin;
            if (span < 0) span = 0;
            for (int k = 0; k < sp->n_points; ++k) {
                double u = sp->n_points > 1 ? sp->near + (sp->far - sp->near) * k / (sp->n_points - 1) : sp->near;
                double al = sp->clip ? amin + u * (amax - amin) : u;
                double p[3];
                long i0[3];
                double f[3];
                for (int ax = 0; ax < 3; ++ax) {
                    p[ax] = sp->a[ax] * (s[ax] + al * d[ax]) + sp->b[ax];
                    double fl = floor(p[ax]);
                    i0[ax] = (long)fl;
                    f[ax] = p[ax] - fl;
                }
                double v = 0.0;
                for (int c = 0; c < 8; ++c) {
                    int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
                    double w = (dx ? f[0] : 1 - f[0]) * (dy ? f[1] : 1 - f[1]) * (dz ? f[2] : 1 - f[2]);
                    v += w * vol_at(vol, S, i0[0] + dx, i0[1] + dy, i0[2] + dz);
                }
                long ch = 0;
                if (mask) {
                    ch = nearest_label(mask, S, p);
                    if (ch < 0 || ch >= C) return -2;
                }
                out[((size_t)bi * C + ch) * n + r] += v;
            }
            double scale = len[(size_t)bi * n + r] / sp->denom * (sp->clip ? span : 1.0);
            for (int ch = 0; ch < C; ++ch) out[((size_t)bi * C + ch) * n + r] *= scale;
        }
    }
    return 0;
}

static int cmp_double(const void *x, const void *y) {
    double a = *(const double *)x, b = *(const double *)y;
    return (a > b) - (a < b);
}

int oracle_siddon(const float *vol, const int *S, const double *src, const double *tgt,
                  const double *len, int B, int n, const oracle_spec *sp, const float *mask,
                  int C, double *out, long long *n_segments) {
    int P = S[0] + S[1] + S[2] + 3;
    double *al = (double *)malloc(sizeof(double) * P);
    if (!al) return -1;
    long long segs = 0;
    memset(out, 0, sizeof(double) * (size_t)B * C * n);
    for (int bi = 0; bi < B; ++bi) {
        const double *s = src + 3 * (size_t)bi;
        for (int r = 0; r < n; ++r) {
            const double *t = tgt + 3 * ((size_t)bi * n + r);
            double d[3] = {t[0] - s[0] + sp->eps, t[1] - s[1] + sp->eps, t[2] - s[2] + sp->eps};
            int m = 0;
            for (int ax = 0; ax < 3; ++ax)
                for (int i = 0; i <= S[ax]; ++i) al[m++] = ((double)i - sp->voxel_shift - s[ax]) / d[ax];
            qsort(al, P, sizeof(double), cmp_double);
            double amin = 0, amax = 1;
            slab(s, d, S, sp->voxel_shift, &amin, &amax);
            if (sp->per_ray_clamp && !(amax > amin)) continue; /* ray misses the volume */
            for (int j = 0; j + 1 < P; ++j) {
                double a0 = al[j], a1 = al[j + 1];
                if (sp->per_ray_clamp) { /* clip the segment into [amin, amax] */
                    a0 = a0 < amin ? amin : (a0 > amax ? amax : a0);
                    a1 = a1 < amin ? amin : (a1 > amax ? amax : a1);
                }
                double mid = 0.5 * (a0 + a1);
                double seg = a1 - a0;
                if (!(seg > 0) || !isfinite(seg)) continue;
                double p[3];
                for (int ax = 0; ax < 3; ++ax) p[ax] = sp->a[ax] * (s[ax] + mid * d[ax]) + sp->b[ax];
                long i = (long)nearbyint(p[0]), jj = (long)nearbyint(p[1]), k = (long)nearbyint(p[2]);
                if (i < 0 || jj < 0 || k < 0 || i >= S[0] || jj >= S[1] || k >= S[2]) continue;
                double v = vol_at(vol, S, i, jj, k);
                long ch = 0;
                if (mask) {
                    ch = (long)vol_at(mask, S, i, jj, k);
                    if (ch < 0 || ch >= C) { free(al); return -2; }
                }
                out[((size_t)bi * C + ch) * n + r] += v * seg;
                ++segs;
            }
            double L = len[(size_t)bi * n + r];
            for (int ch = 0; ch < C; ++ch) out[((size_t)bi * C + ch) * n + r] *= L;
        }
    }
    free(al);
    if (n_segments) *n_segments = segs;
    return 0;
}

int oracle_abi_version(void) { return 1; }
