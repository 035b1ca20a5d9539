/* CPU oracle: rectangular linear sum assignment (TEST INFRASTRUCTURE - not product code).
 *
 * Restates the algorithm behind scipy.optimize.linear_sum_assignment (third-party dependency
 * of the reference, pinned scipy=1.7.3 in AGQA/requirements.txt:82; call sites
 * AGQA/src/lxrt/matcher.py:79 and :103): Crouse's modified Jonker-Volgenant
 * shortest-augmenting-path method, with the details that decide WHICH optimal assignment
 * comes back when there are ties:
 *   - a matrix with more rows than columns is solved transposed;
 *   - unscanned columns are kept in a list initialised in reverse order and swap-removed;
 *   - among equal reduced path costs a still-unassigned column wins, else the first seen.
 * All arithmetic is float64 and comparisons are exact.
 * Pinned by tests/golden/lsap_*.npz (generated with SciPy 1.15.3 in the build container).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int solve_wide(const double *cost, int nr, int nc, int64_t *col4row)
{
    /* nr <= nc; cost is row-major nr x nc */
    double *u = calloc(nr, sizeof(double)), *v = calloc(nc, sizeof(double));
    double *spc = malloc(nc * sizeof(double));
    int64_t *row4col = malloc(nc * sizeof(int64_t)), *path = malloc(nc * sizeof(int64_t));
    int *remaining = malloc(nc * sizeof(int));
    char *in_sr = malloc(nr), *in_sc = malloc(nc);
    for (int j = 0; j < nc; j++) row4col[j] = -1;
    for (int i = 0; i < nr; i++) col4row[i] = -1;

    for (int cur = 0; cur < nr; cur++) {
        double min_val = 0.0;
        int i = cur, sink = -1, n_rem = nc;
        memset(in_sr, 0, nr);
        memset(in_sc, 0, nc);
        for (int j = 0; j < nc; j++) { spc[j] = INFINITY; path[j] = -1; remaining[j] = nc - 1 - j; }
        while (sink == -1) {
            int best = -1;
            double lowest = INFINITY;
            in_sr[i] = 1;
            for (int it = 0; it < n_rem; it++) {
                int j = remaining[it];
                double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; best = it; }
            }
            min_val = lowest;
            if (best < 0 || isinf(min_val)) { sink = -2; break; }   /* infeasible */
            int j = remaining[best];
            if (row4col[j] == -1) sink = j; else i = (int)row4col[j];
            in_sc[j] = 1;
            remaining[best] = remaining[--n_rem];
        }
        if (sink < 0) { free(u); free(v); free(spc); free(row4col); free(path); free(remaining); free(in_sr); free(in_sc); return -1; }
        u[cur] += min_val;
        for (int r = 0; r < nr; r++)
            if (in_sr[r] && r != cur) u[r] += min_val - spc[col4row[r]];
        for (int c = 0; c < nc; c++)
            if (in_sc[c]) v[c] -= min_val - spc[c];
        int j = sink;
        for (;;) {
            int r = (int)path[j];
            row4col[j] = r;
            int64_t prev = col4row[r];
            col4row[r] = j;
            j = (int)prev;
            if (r == cur) break;
        }
    }
    free(u); free(v); free(spc); free(row4col); free(path); free(remaining); free(in_sr); free(in_sc);
    return 0;
}

/* cost: row-major nr x nc float64. rows/cols: int64[min(nr,nc)] outputs sorted by row.
 * Returns the number of assigned pairs, or -1 if infeasible. */
int lsap_solve(const double *cost, int nr, int nc, int64_t *rows, int64_t *cols)
{
    if (nr == 0 || nc == 0) return 0;
    if (nc >= nr) {
        if (solve_wide(cost, nr, nc, cols)) return -1;
        for (int i = 0; i < nr; i++) rows[i] = i;
        return nr;
    }
    /* more rows than columns: solve the transpose, then order by original row */
    double *t = malloc(sizeof(double) * (size_t)nr * nc);
    for (int i = 0; i < nr; i++)
        for (int j = 0; j < nc; j++) t[(size_t)j * nr + i] = cost[(size_t)i * nc + j];
    int64_t *c4r = malloc(sizeof(int64_t) * nc);           /* for each original column: its row */
    int rc = solve_wide(t, nc, nr, c4r);
    free(t);
    if (rc) { free(c4r); return -1; }
    /* stable sort of columns by assigned row (insertion sort; sizes are tiny) */
    int *order = malloc(sizeof(int) * nc);
    for (int j = 0; j < nc; j++) order[j] = j;
    for (int a = 1; a < nc; a++) {
        int key = order[a], b = a - 1;
        while (b >= 0 && c4r[order[b]] > c4r[key]) { order[b + 1] = order[b]; b--; }
        order[b + 1] = key;
    }
    for (int k = 0; k < nc; k++) { rows[k] = c4r[order[k]]; cols[k] = order[k]; }
    free(order); free(c4r);
    return nc;
}
