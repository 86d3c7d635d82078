/*
 * CPU ORACLE helper (plain C) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Restates two scipy.ndimage (scipy==1.15.3) operations the reference calls on
 * its Label path:
 *   - scipy.ndimage.label(mask, structure)           labelling.py:489, 507
 *       connected components; ids are int32, 1..K, increasing with the C-order
 *       (raster) index of each component's first voxel.
 *   - scipy.ndimage.binary_fill_holes(mask)          labelling.py:486
 *       complement of the 6-connected background region reachable from the
 *       volume border (scipy implements it as binary_dilation of the border
 *       seed inside ~mask until convergence, then inverts).
 * Classic two-pass union-find raster scan and a BFS flood fill: deliberately
 * different algorithms from the GPU kernels they check.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t uf_find(int64_t *p, int64_t x) {
    int64_t r = x;
    while (p[r] != r) r = p[r];
    while (p[x] != r) { int64_t n = p[x]; p[x] = r; x = n; }
    return r;
}

static void uf_union(int64_t *p, int64_t a, int64_t b) {
    a = uf_find(p, a); b = uf_find(p, b);
    if (a == b) return;
    if (a < b) p[b] = a; else p[a] = b;
}

/* conn = 26 (full 3x3x3) or 6 (cross).  Returns the number of components. */
int64_t orc_label(const uint8_t *mask, int32_t *out, int64_t nz, int64_t ny, int64_t nx, int conn) {
    int64_t n = nz * ny * nx;
    int64_t *parent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!parent) return -1;
    for (int64_t i = 0; i < n; ++i) parent[i] = i;
    for (int64_t z = 0; z < nz; ++z)
        for (int64_t y = 0; y < ny; ++y)
            for (int64_t x = 0; x < nx; ++x) {
                int64_t i = (z * ny + y) * nx + x;
                if (!mask[i]) continue;
                /* the 13 raster-preceding neighbours (26-conn) or 3 (6-conn) */
                for (int dz = -1; dz <= 0; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue;
                            if (conn == 6 && (abs(dz) + abs(dy) + abs(dx)) != 1) continue;
                            int64_t zz = z + dz, yy = y + dy, xx = x + dx;
                            if (zz < 0 || yy < 0 || yy >= ny || xx < 0 || xx >= nx) continue;
                            int64_t j = (zz * ny + yy) * nx + xx;
                            if (mask[j]) uf_union(parent, i, j);
                        }
            }
    /* roots are minimal raster indices => a raster walk numbers components in scipy's order */
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!mask[i]) { out[i] = 0; continue; }
        int64_t r = uf_find(parent, i);
        if (r == i) { out[i] = (int32_t)(++k); }
        else out[i] = out[r];
    }
    free(parent);
    return k;
}

void orc_fill_holes(const uint8_t *mask, uint8_t *out, int64_t nz, int64_t ny, int64_t nx) {
    int64_t n = nz * ny * nx;
    uint8_t *reach = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t *queue = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t head = 0, tail = 0;
    for (int64_t z = 0; z < nz; ++z)
        for (int64_t y = 0; y < ny; ++y)
            for (int64_t x = 0; x < nx; ++x) {
                if (!(z == 0 || z == nz - 1 || y == 0 || y == ny - 1 || x == 0 || x == nx - 1)) continue;
                int64_t i = (z * ny + y) * nx + x;
                if (!mask[i] && !reach[i]) { reach[i] = 1; queue[tail++] = i; }
            }
    while (head < tail) {
        int64_t i = queue[head++];
        int64_t x = i % nx, y = (i / nx) % ny, z = i / (nx * ny);
        const int64_t cand[6] = {
            z > 0 ? i - nx * ny : -1, z < nz - 1 ? i + nx * ny : -1,
            y > 0 ? i - nx : -1,      y < ny - 1 ? i + nx : -1,
            x > 0 ? i - 1 : -1,       x < nx - 1 ? i + 1 : -1 };
        for (int c = 0; c < 6; ++c) {
            int64_t j = cand[c];
            if (j >= 0 && !mask[j] && !reach[j]) { reach[j] = 1; queue[tail++] = j; }
        }
    }
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)(mask[i] || !reach[i]);
    free(reach);
    free(queue);
}
