/* rtdd_mg_oracle.c -- TEST INFRASTRUCTURE, not product code (see rtdd_oracle.c's header).
 *
 * CPU restatement of the multigrid V-cycle EXTENSION (BASELINE config 5; the reference has no multigrid, so there is
 * nothing of the reference's to pin this against: PARITY UNPINNED by the reference; pinned instead by scipy's direct
 * solution in tests/ and by the residual it reaches).  The algorithm is Dendy's black-box multigrid on the image grid:
 *   - level 0 = the image, smoothed by the red-black Gauss-Seidel sweep of rtdd_oracle.c on the true operator;
 *   - hierarchy operator at level 0 = that operator with links < theta moved to the diagonal only;
 *   - coarse point (I,J) = fine point (2I,2J); interpolation weights from the stencil (edge points: stencil collapsed
 *     across the edge; cell centres: own equation with edge neighbours replaced by their interpolants);
 *   - Galerkin coarse operators P^T A P (symmetric 9-point: couplings E,S,SE,SW + diagonal D; D == 0 = inactive);
 *   - four-colour Gauss-Seidel on the coarse levels, 30 sweeps on the coarsest;
 *   - V(2,2) cycles, residual max|J(x)-x| checked every `check_every` cycles;
 *   - with a check every cycle: vector extrapolation x + lambda/(1-lambda) (x - x_prev) whenever the residual ratio lambda of
 *     two consecutive cycles agrees within 5 % (0.3 < lambda < 0.995), at most every third cycle.
 * Plain scalar f32 C, every sum in a fixed order, compiled with -ffp-contract=off. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

void orc_rbgs_sweep(float *x, const int32_t *index2, const uint8_t *mask, size_t maskPitch, int rows, int cols, const float *lut, int contract, float omega);
float orc_residual(const float *in, const int32_t *index2, const uint8_t *mask, size_t maskPitch, int rows, int cols, const float *lut, int contract);

#define MG_THETA 1e-4f
#define MG_NU 2
#define MG_COARSEST 30
#define MG_MAXLEV 16

typedef struct {
    int rows, cols;
    float *a[12];                 /* E S SE SW D P0 P1 P2 P3 e b r */
} mg_level;

static mg_level g_lv[MG_MAXLEV];
static int g_nlev = 0;

static float at(const float *a, int rows, int cols, int y, int x) { return (y >= 0 && y < rows && x >= 0 && x < cols) ? a[(size_t)y * cols + x] : 0.0f; }

static float coupling(const mg_level *l, int y, int x, int dy, int dx) {
    const int R = l->rows, C = l->cols;
    if (dy == 0) return dx > 0 ? at(l->a[0], R, C, y, x) : at(l->a[0], R, C, y, x - 1);
    if (dy > 0) {
        if (dx == 0) return at(l->a[1], R, C, y, x);
        return dx > 0 ? at(l->a[2], R, C, y, x) : at(l->a[3], R, C, y, x);
    }
    if (dx == 0) return at(l->a[1], R, C, y - 1, x);
    return dx < 0 ? at(l->a[2], R, C, y - 1, x - 1) : at(l->a[3], R, C, y - 1, x + 1);
}

static float pweight(const mg_level *l, int y, int x, int I, int J) {
    if (y < 0 || y >= l->rows || x < 0 || x >= l->cols) return 0.0f;
    const int di = I - (y >> 1), dj = J - (x >> 1);
    if (di < 0 || di > 1 || dj < 0 || dj > 1) return 0.0f;
    return l->a[5 + di * 2 + dj][(size_t)y * l->cols + x];
}

static void mg_free(void) {
    for (int l = 0; l < g_nlev; l++) for (int i = 0; i < 12; i++) { free(g_lv[l].a[i]); g_lv[l].a[i] = NULL; }
    g_nlev = 0;
}

static void build_p(mg_level *l) {
    const int R = l->rows, C = l->cols;
    float *P0 = l->a[5], *P1 = l->a[6], *P2 = l->a[7], *P3 = l->a[8];
    for (int y = 0; y < R; y++)
        for (int x = 0; x < C; x++) {
            const size_t q = (size_t)y * C + x;
            const float d = l->a[4][q];
            const int act = d > 0.0f, oy = y & 1, ox = x & 1;
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
            if (!oy && !ox) p0 = act ? 1.0f : 0.0f;
            else if (!oy && ox) {
                const float den = (d - coupling(l, y, x, -1, 0)) - coupling(l, y, x, 1, 0);
                if (act && den > 0.0f) {
                    p0 = ((coupling(l, y, x, 0, -1) + coupling(l, y, x, -1, -1)) + coupling(l, y, x, 1, -1)) / den;
                    p1 = ((coupling(l, y, x, 0, 1) + coupling(l, y, x, -1, 1)) + coupling(l, y, x, 1, 1)) / den;
                }
            } else if (oy && !ox) {
                const float den = (d - coupling(l, y, x, 0, -1)) - coupling(l, y, x, 0, 1);
                if (act && den > 0.0f) {
                    p0 = ((coupling(l, y, x, -1, 0) + coupling(l, y, x, -1, -1)) + coupling(l, y, x, -1, 1)) / den;
                    p2 = ((coupling(l, y, x, 1, 0) + coupling(l, y, x, 1, -1)) + coupling(l, y, x, 1, 1)) / den;
                }
            }
            P0[q] = p0; P1[q] = p1; P2[q] = p2; P3[q] = 0.0f;
        }
    for (int y = 1; y < R; y += 2)
        for (int x = 1; x < C; x += 2) {
            const size_t q = (size_t)y * C + x;
            const float d = l->a[4][q];
            if (!(d > 0.0f)) continue;
            const float wn = coupling(l, y, x, -1, 0), ws = coupling(l, y, x, 1, 0), ww = coupling(l, y, x, 0, -1), we = coupling(l, y, x, 0, 1);
            const float n0 = at(P0, R, C, y - 1, x), n1 = at(P1, R, C, y - 1, x);
            const float s0 = at(P0, R, C, y + 1, x), s1 = at(P1, R, C, y + 1, x);
            const float w0 = at(P0, R, C, y, x - 1), w2 = at(P2, R, C, y, x - 1);
            const float e0 = at(P0, R, C, y, x + 1), e2 = at(P2, R, C, y, x + 1);
            P0[q] = ((coupling(l, y, x, -1, -1) + wn * n0) + ww * w0) / d;
            P1[q] = ((coupling(l, y, x, -1, 1) + wn * n1) + we * e0) / d;
            P2[q] = ((coupling(l, y, x, 1, -1) + ws * s0) + ww * w2) / d;
            P3[q] = ((coupling(l, y, x, 1, 1) + ws * s1) + we * e2) / d;
        }
}

static void galerkin(const mg_level *f, mg_level *c) {
    static const int tdi[5] = {0, 0, 1, 1, 1}, tdj[5] = {0, 1, 0, 1, -1};
    const int R = c->rows, C = c->cols;
    for (int I = 0; I < R; I++)
        for (int J = 0; J < C; J++) {
            float out[5];
            for (int t = 0; t < 5; t++) {
                const int TI = I + tdi[t], TJ = J + tdj[t];
                float acc = 0.0f;
                if (TI < R && TJ >= 0 && TJ < C)
                    for (int py = -1; py <= 1; py++)
                        for (int px = -1; px <= 1; px++) {
                            const int y = 2 * I + py, x = 2 * J + px;
                            const float wp = pweight(f, y, x, I, J);
                            if (wp == 0.0f) continue;
                            for (int qy = -1; qy <= 1; qy++)
                                for (int qx = -1; qx <= 1; qx++) {
                                    const float wq = pweight(f, y + qy, x + qx, TI, TJ);
                                    if (wq == 0.0f) continue;
                                    const float a = (qy == 0 && qx == 0) ? f->a[4][(size_t)y * f->cols + x] : -coupling(f, y, x, qy, qx);
                                    acc += wp * (a * wq);
                                }
                        }
                out[t] = acc;
            }
            const size_t q = (size_t)I * C + J;
            const int act = out[0] > 0.0f;
            c->a[4][q] = act ? out[0] : 0.0f;
            c->a[0][q] = act ? -out[1] : 0.0f; c->a[1][q] = act ? -out[2] : 0.0f; c->a[2][q] = act ? -out[3] : 0.0f; c->a[3][q] = act ? -out[4] : 0.0f;
        }
    for (int y = 0; y < R; y++)                       /* couplings that end at an inactive point */
        for (int x = 0; x < C; x++) {
            const size_t q = (size_t)y * C + x;
            if (!(at(c->a[4], R, C, y, x + 1) > 0.0f)) c->a[0][q] = 0.0f;
            if (!(at(c->a[4], R, C, y + 1, x) > 0.0f)) c->a[1][q] = 0.0f;
            if (!(at(c->a[4], R, C, y + 1, x + 1) > 0.0f)) c->a[2][q] = 0.0f;
            if (!(at(c->a[4], R, C, y + 1, x - 1) > 0.0f)) c->a[3][q] = 0.0f;
        }
}

static float gs_sum(const mg_level *l, int y, int x) {
    const int R = l->rows, C = l->cols;
    const float *e = l->a[9];
    float v = l->a[10][(size_t)y * C + x];
    v += coupling(l, y, x, 0, -1) * at(e, R, C, y, x - 1);
    v += coupling(l, y, x, 0, 1) * at(e, R, C, y, x + 1);
    v += coupling(l, y, x, -1, 0) * at(e, R, C, y - 1, x);
    v += coupling(l, y, x, 1, 0) * at(e, R, C, y + 1, x);
    v += coupling(l, y, x, -1, -1) * at(e, R, C, y - 1, x - 1);
    v += coupling(l, y, x, -1, 1) * at(e, R, C, y - 1, x + 1);
    v += coupling(l, y, x, 1, -1) * at(e, R, C, y + 1, x - 1);
    v += coupling(l, y, x, 1, 1) * at(e, R, C, y + 1, x + 1);
    return v;
}

static void smooth(mg_level *l, int nsweeps, int reverse) {
    for (int s = 0; s < nsweeps; s++)
        for (int c = 0; c < 4; c++) {
            const int colour = reverse ? 3 - c : c;
            for (int y = colour >> 1; y < l->rows; y += 2)
                for (int x = colour & 1; x < l->cols; x += 2) {
                    const size_t q = (size_t)y * l->cols + x;
                    const float d = l->a[4][q];
                    if (d > 0.0f) l->a[9][q] = gs_sum(l, y, x) / d;
                }
        }
}

static void residual(mg_level *l) {
    for (int y = 0; y < l->rows; y++)
        for (int x = 0; x < l->cols; x++) {
            const size_t q = (size_t)y * l->cols + x;
            const float d = l->a[4][q];
            l->a[11][q] = d > 0.0f ? gs_sum(l, y, x) - d * l->a[9][q] : 0.0f;
        }
}

static void restrict_to(const mg_level *f, mg_level *c) {
    for (int I = 0; I < c->rows; I++)
        for (int J = 0; J < c->cols; J++) {
            float acc = 0.0f;
            for (int py = -1; py <= 1; py++)
                for (int px = -1; px <= 1; px++) {
                    const int y = 2 * I + py, x = 2 * J + px;
                    const float wp = pweight(f, y, x, I, J);
                    if (wp != 0.0f) acc += wp * f->a[11][(size_t)y * f->cols + x];
                }
            c->a[10][(size_t)I * c->cols + J] = acc;
            c->a[9][(size_t)I * c->cols + J] = 0.0f;
        }
}

static void prolong_add(const mg_level *c, const mg_level *f, float *target) {
    for (int y = 0; y < f->rows; y++)
        for (int x = 0; x < f->cols; x++) {
            const size_t q = (size_t)y * f->cols + x;
            const int I = y >> 1, J = x >> 1;
            float v = f->a[5][q] * at(c->a[9], c->rows, c->cols, I, J);
            v += f->a[6][q] * at(c->a[9], c->rows, c->cols, I, J + 1);
            v += f->a[7][q] * at(c->a[9], c->rows, c->cols, I + 1, J);
            v += f->a[8][q] * at(c->a[9], c->rows, c->cols, I + 1, J + 1);
            target[q] += v;
        }
}

/* x: rows x cols dense, Dirichlet values in place where mask == 255; index2 as orc_index_to_weight writes it. */
ORC_API int orc_mg_solve(float *x, const int32_t *index2, const uint8_t *mask, size_t maskPitch, int rows, int cols, const float *lut, int contract,
                         int max_cycles, float tolerance, int check_every, double alternative_seconds, double cycle_seconds, int *cycles_done, float *residual_out) {
    mg_free();
    int r = rows, c = cols;
    for (int l = 0; l < MG_MAXLEV; l++) {
        g_lv[l].rows = r; g_lv[l].cols = c;
        for (int i = 0; i < 12; i++) g_lv[l].a[i] = (float *)calloc((size_t)r * c, sizeof(float));
        g_nlev = l + 1;
        if ((size_t)r * c <= 256 || (r == 1 && c == 1)) break;
        r = (r + 1) / 2; c = (c + 1) / 2;
    }
    mg_level *L0 = &g_lv[0];
    for (int y = 0; y < rows; y++)
        for (int xx = 0; xx < cols; xx++) {
            const size_t p = (size_t)y * cols + xx;
            const int fr = mask[(size_t)y * maskPitch + xx] != 255;
            const float wl = lut[index2[2 * p] / 1000], wr = lut[index2[2 * p] % 1000], wu = lut[index2[2 * p + 1] / 1000], wd = lut[index2[2 * p + 1] % 1000];
            float d = 0.0f;
            d += wl; d += wr; d += wu; d += wd;
            const int rfree = xx + 1 < cols && mask[(size_t)y * maskPitch + xx + 1] != 255;
            const int dfree = y + 1 < rows && mask[(size_t)(y + 1) * maskPitch + xx] != 255;
            L0->a[0][p] = (fr && rfree && wr >= MG_THETA) ? wr : 0.0f;
            L0->a[1][p] = (fr && dfree && wd >= MG_THETA) ? wd : 0.0f;
            L0->a[4][p] = fr ? d : 0.0f;
        }
    for (int l = 0; l + 1 < g_nlev; l++) { build_p(&g_lv[l]); galerkin(&g_lv[l], &g_lv[l + 1]); }

    const int last = g_nlev - 1;
    *cycles_done = 0;
    *residual_out = NAN;
    float before = INFINITY, before2 = INFINITY;
    int since = 0;
    float *xprev = (float *)malloc((size_t)rows * cols * sizeof(float));
    while (*cycles_done < max_cycles) {
        memcpy(xprev, x, (size_t)rows * cols * sizeof(float));                  /* x_{k-1} for the extrapolation */
        if (last == 0) {
            for (int s = 0; s < 2 * MG_NU; s++) orc_rbgs_sweep(x, index2, mask, maskPitch, rows, cols, lut, contract, 1.0f);
        } else {
            for (int s = 0; s < MG_NU; s++) orc_rbgs_sweep(x, index2, mask, maskPitch, rows, cols, lut, contract, 1.0f);
            for (int y = 0; y < rows; y++)                 /* residual from exact differences */
                for (int xx = 0; xx < cols; xx++) {
                    const size_t p = (size_t)y * cols + xx;
                    float rr = 0.0f;
                    if (mask[(size_t)y * maskPitch + xx] != 255) {
                        const float xc = x[p];
                        if (xx > 0) rr += lut[index2[2 * p] / 1000] * (x[p - 1] - xc);
                        if (xx + 1 < cols) rr += lut[index2[2 * p] % 1000] * (x[p + 1] - xc);
                        if (y > 0) rr += lut[index2[2 * p + 1] / 1000] * (x[p - cols] - xc);
                        if (y + 1 < rows) rr += lut[index2[2 * p + 1] % 1000] * (x[p + cols] - xc);
                    }
                    L0->a[11][p] = rr;
                }
            for (int l = 0; l < last; l++) {
                restrict_to(&g_lv[l], &g_lv[l + 1]);
                if (l + 1 == last) { smooth(&g_lv[l + 1], MG_COARSEST, 0); break; }
                smooth(&g_lv[l + 1], MG_NU, 0);
                residual(&g_lv[l + 1]);
            }
            for (int l = last - 1; l >= 1; l--) { prolong_add(&g_lv[l + 1], &g_lv[l], g_lv[l].a[9]); smooth(&g_lv[l], MG_NU, 1); }
            prolong_add(&g_lv[1], L0, x);
            for (int s = 0; s < MG_NU; s++) orc_rbgs_sweep(x, index2, mask, maskPitch, rows, cols, lut, contract, 1.0f);
        }
        (*cycles_done)++;
        if (tolerance > 0.0f && (*cycles_done % check_every == 0 || *cycles_done == max_cycles)) {
            *residual_out = orc_residual(x, index2, mask, maskPitch, rows, cols, lut, contract);
            if (*residual_out <= tolerance) break;
            if (alternative_seconds > 0.0 && before2 < INFINITY) {   /* RTDD_METHOD_AUTO: leave when the cycles still needed cost more */
                const double rate = sqrt((double)*residual_out / (double)before2);
                if (!(rate < 1.0)) break;
                const double needed = ceil(log((double)*residual_out / (double)tolerance) / -log(rate));
                if (needed * cycle_seconds > alternative_seconds) break;       /* both prices are the caller's constants */
            }
            /* vector extrapolation: the residual shrank by the same factor lambda twice in a row -> remove that family */
            since++;
            if (check_every == 1 && since >= 3 && before2 < INFINITY && last > 0 && *cycles_done < max_cycles) {
                const double l1 = (double)*residual_out / (double)before, l0 = (double)before / (double)before2;
                if (l1 > 0.3 && l1 < 0.995 && fabs(l1 - l0) <= 0.05 * l1) {
                    const float alpha = (float)(l1 / (1.0 - l1));
                    for (size_t p = 0; p < (size_t)rows * cols; p++) {
                        const float v = x[p], d = v - xprev[p];
                        const float w = contract ? fmaf(alpha, d, v) : v + alpha * d;
                        x[p] = fminf(fmaxf(w, 0.0f), 255.0f);
                    }
                    since = 0;
                }
            }
            before2 = before; before = *residual_out;
        }
    }
    free(xprev);
    return g_nlev;
}

/* plane `which` of level `level` of the last hierarchy built (dense rows x cols); returns 0, or -1 when out of range */
ORC_API int orc_mg_level(int level, int which, float *out, int *rows, int *cols) {
    if (level < 0 || level >= g_nlev || which < 0 || which >= 12) return -1;
    *rows = g_lv[level].rows; *cols = g_lv[level].cols;
    if (out) memcpy(out, g_lv[level].a[which], (size_t)g_lv[level].rows * g_lv[level].cols * sizeof(float));
    return 0;
}
