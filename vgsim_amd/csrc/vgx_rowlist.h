// vgx_rowlist.h — occupancy-list maintenance of the four-replicates-per-wavefront kernels whose lists may hold ZERO-COUNT entries
// during a launch (vgx_quadf.hip: FAST mode; vgx_quad.hip: the long-list exact kernel).  A count that drops to 0 stays in the list
// (a term of +0.0 in every sum, weight 0 in every choice; the reference's dense arrays hold such zeros everywhere), a later
// insertion of that haplotype revives it, an insertion next to it takes its slot: nothing moves on removal.  An insertion moves
// the entries from its position up to the NEXT zero-count entry (or the end of the list) one slot up, a 64-entry tile per step in
// registers.  After the launch vgx_lists_settle_kernel squeezes the zero entries out and rewrites what the other kernels read.
// A row = 16 lanes = one replicate (vgx_rowprim.h); all lanes of the wavefront call together.
#pragma once
#include "vgx_rowprim.h"

struct __attribute__((packed, aligned(4))) QV4 { int x, y, z, w; };   // four list entries at any 4-byte boundary
// inclusive int32 prefix inside each row
static __device__ __forceinline__ int row_iscan32(int v) {
    v += VGX_DPP_SHR(v, 1); v += VGX_DPP_SHR(v, 2); v += VGX_DPP_SHR(v, 4); v += VGX_DPP_SHR(v, 8);
    return v;
}
// A row squeezes the zero-count entries out of its list (in place, tile by tile) and rewrites the tile sums; returns the new length.
// Rows with on = false pass through (all rows of the wavefront must call).  UN tiles are loaded before any of them is stored (the
// stores of a tile never reach beyond it, so the later tiles may already be in registers); c64: the 8-byte counts are written too.
struct __attribute__((aligned(16))) QL2 { int64_t a, b; };
template <int UN>
static __device__ __forceinline__ int q_compact_list(int32_t *lh, int32_t *l3, int64_t *lt, int n, bool on, int64_t *c64 = nullptr) {
    const int rl = threadIdx.x & 15;
    const int maxn = rows_max(on ? n : 0);
    int out = 0;
    for (int tb0 = 0; tb0 < maxn; tb0 += 64 * UN) {
        QV4 hv[UN], cv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int tb = tb0 + 64 * u;
            hv[u] = QV4{0, 0, 0, 0}; cv[u] = QV4{0, 0, 0, 0};
            if (on && tb < n) { hv[u] = *(const QV4 *)(lh + tb + 4 * rl); cv[u] = *(const QV4 *)(l3 + tb + 4 * rl); }
        }
        WSYNC();
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int tb = tb0 + 64 * u;
            const bool in = on && tb < n;
            const int e0 = tb + 4 * rl;
            const bool k0 = in && e0 + 0 < n && cv[u].x != 0, k1 = in && e0 + 1 < n && cv[u].y != 0, k2 = in && e0 + 2 < n && cv[u].z != 0,
                       k3 = in && e0 + 3 < n && cv[u].w != 0;
            const int cnt = (int)k0 + (int)k1 + (int)k2 + (int)k3;
            const int incl = row_iscan32(cnt);
            int pos = out + incl - cnt;
            if (cnt == 4 && pos == e0) {      // an untouched chunk of four stays where it is
                if (c64) { *(QL2 *)(c64 + pos) = QL2{cv[u].x, cv[u].y}; *(QL2 *)(c64 + pos + 2) = QL2{cv[u].z, cv[u].w}; }
            } else {
                if (k0) { lh[pos] = hv[u].x; l3[pos] = cv[u].x; if (c64) c64[pos] = cv[u].x; pos += 1; }
                if (k1) { lh[pos] = hv[u].y; l3[pos] = cv[u].y; if (c64) c64[pos] = cv[u].y; pos += 1; }
                if (k2) { lh[pos] = hv[u].z; l3[pos] = cv[u].z; if (c64) c64[pos] = cv[u].z; pos += 1; }
                if (k3) { lh[pos] = hv[u].w; l3[pos] = cv[u].w; if (c64) c64[pos] = cv[u].w; pos += 1; }
            }
            out += rowget_i32(incl, 15);
        }
    }
    WSYNC();
    const int nn = on ? out : 0;
    // the tile sums (maintained by the event loop while nothing moves: rewritten only for lists that lost entries)
    const bool redo = on && nn != n && nn > 64;
    const int maxn2 = rows_max(redo ? nn : 0);
    for (int tb0 = 0; tb0 < maxn2; tb0 += 64 * UN) {
        QV4 cv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int tb = tb0 + 64 * u;
            cv[u] = QV4{0, 0, 0, 0};
            if (redo && tb < nn) cv[u] = *(const QV4 *)(l3 + tb + 4 * rl);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int tb = tb0 + 64 * u;
            const int e0 = tb + 4 * rl;
            const int sm = (e0 + 0 < nn ? cv[u].x : 0) + (e0 + 1 < nn ? cv[u].y : 0) + (e0 + 2 < nn ? cv[u].z : 0) + (e0 + 3 < nn ? cv[u].w : 0);
            const int tot = rowget_i32(row_iscan32(sm), 15);
            if (redo && tb < nn && rl == 0) lt[tb >> 6] = tot;
        }
    }
    // the tile sums behind the shortened list go back to 0 (vgx_dev.h: "0 beyond the list"; an insertion that opens a tile adds to them)
    {
        const int j0 = nn > 64 ? (nn + 63) >> 6 : 0, j1 = on ? n >> 6 : -1;
        const int mj = rows_max(on && nn != n ? j1 + 1 : 0);
        for (int jb = 0; jb < mj; jb += 16) {
            const int j = jb + rl;
            if (on && nn != n && j >= j0 && j <= j1) lt[j] = 0;
        }
    }
    WSYNC();
    return nn;
}


// infectious[population of the lists given, hap] += 1 for the rows with `act` (a mutant or a migrant arrives).  lh / lc / l3 / lt:
// the population's haplotypes, rate classes (all 0 in these kernels), 4-byte counts and tile sums; l8 (BYTE8): its one-byte counts
// (min(count, 255)).  n: the list's length incl. zero-count entries, updated.  Returns false where the list is full (cap < H and no
// zero-count entry left to squeeze out).  zeros: the number of zero-count entries the caller believes the list holds (kept up to
// date here: -1 when the arrival takes one, 0 after a squeeze).
template <bool BYTE8>
static __device__ __forceinline__ bool q_list_add_one(bool act, int hap, int32_t *lh, int32_t *lc, int32_t *l3, int64_t *lt, uint8_t *l8,
                                                      int &n, int cap, int H, int &zeros) {
    const int rl = threadIdx.x & 15;
    const int delta = 1;
    bool ok = true;
    if (__builtin_expect(__ballot(act && n >= cap && cap < H) != 0, 0)) {   // a full list: squeeze its zero-count entries out first
        const bool cm = act && n >= cap && cap < H;
        const int n2 = q_compact_list<1>(lh, l3, lt, n, cm);
        if (BYTE8) {
            const int m8 = rows_max(cm ? n2 : 0);
            for (int k = rl; k < m8; k += 16) if (cm && k < n2) l8[k] = (uint8_t)min(l3[k], 255);
        }
        if (cm) { n = n2; zeros = 0; }
        WSYNC();
    }
    // ---- lower bound: first index whose haplotype is >= hap ----
    int posn = 0;
    bool found = false;
    int cur = 0;
    {
        const int ns = act ? n : 0;
        const int maxn = rows_max(ns);
        int lo = 0;             // first entry of the 64^k-aligned window known to contain the bound
        // 64-ary descent over the sorted list (a lane probes four of the 64 sub-windows): strides 64^3, 64^2, 64
        for (int stride = 1 << 18; stride >= 64; stride >>= 6) {
            if (maxn <= stride) continue;
            int c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t k = lo + (int64_t)(4 * rl + j) * stride;
                const int h = k < ns ? lh[k] : 0x7fffffff;
                c += h <= hap ? 1 : 0;
            }
            // probes are sorted: the bound lies at or after the last probe <= hap and before the next probe
            const int nle = rowget_i32(row_iscan32(c), 15);
            lo += nle > 0 ? (nle - 1) * stride : 0;
        }
        // the tile at lo: four consecutive entries per lane, the counts with them
        const QV4 hv = *(const QV4 *)(lh + lo + 4 * rl);
        const QV4 cv = *(const QV4 *)(l3 + lo + 4 * rl);
        const int e0 = lo + 4 * rl;
        const int h0 = e0 + 0 < ns ? hv.x : 0x7fffffff, h1 = e0 + 1 < ns ? hv.y : 0x7fffffff;
        const int h2 = e0 + 2 < ns ? hv.z : 0x7fffffff, h3 = e0 + 3 < ns ? hv.w : 0x7fffffff;
        const int jl = h0 >= hap ? 0 : h1 >= hap ? 1 : h2 >= hap ? 2 : h3 >= hap ? 3 : 4;
        const int q = row_min(jl < 4 ? 4 * rl + jl : 64);
        const int qq = min(q, 63), j = qq & 3;
        const int hq = rowget_i32(j == 0 ? h0 : j == 1 ? h1 : j == 2 ? h2 : h3, qq >> 2);
        const int cq = rowget_i32(j == 0 ? cv.x : j == 1 ? cv.y : j == 2 ? cv.z : cv.w, qq >> 2);
        if (act) {
            posn = min(lo + q, n);
            found = q < 64 && hq == hap;
            cur = found ? cq : 0;
        }
    }
    const bool bump = act && found;                           // count changes in place (a zero-count entry comes back)
    const bool ins = act && !found;                           // a new entry
    if (ins && n >= cap) ok = false;
    const bool ins_ok = ins && ok;
    if (bump && cur == 0) zeros -= 1;
    if (bump && rl == 0) {
        l3[posn] = cur + delta;
        if (BYTE8) l8[posn] = (uint8_t)min(cur + delta, 255);
        if (n > 64) lt[posn >> 6] += delta;
    }
    // ---- insertion: the entries from posn up to the next zero-count entry (or the end of the list) move one slot up, a tile
    // (16 lanes x 4 entries) per step, the entry pushed out of a tile carried into the next ----
    if (__builtin_expect(__ballot(ins_ok) != 0, 0)) {
        int b = ins_ok ? posn : 0, nn = n, ch = hap, cc = delta;
        bool going = ins_ok;
        while (__ballot(going)) {
            const int tb = b & ~63;
            QV4 hv = {0, 0, 0, 0}, cv = {0, 0, 0, 0};
            int64_t tsum = 0;
            if (going) {
                hv = *(const QV4 *)(lh + tb + 4 * rl); cv = *(const QV4 *)(l3 + tb + 4 * rl);
                if (n > 64 && rl == 0) tsum = lt[tb >> 6];
            }
            const int i0 = 4 * rl;                       // index of the lane's first entry inside the tile
            const int bq = b - tb, lim = min(64, nn - tb);
            const int jl = (i0 + 0 >= bq && i0 + 0 < lim && cv.x == 0) ? 0 : (i0 + 1 >= bq && i0 + 1 < lim && cv.y == 0) ? 1
                         : (i0 + 2 >= bq && i0 + 2 < lim && cv.z == 0) ? 2 : (i0 + 3 >= bq && i0 + 3 < lim && cv.w == 0) ? 3 : 4;
            const int zq = row_min(going && jl < 4 ? i0 + jl : 64);
            const int endq = zq < 64 ? zq : lim;         // the slot that takes the last moved entry (64: the next tile's first)
            const int ph = VGX_DPP_SHR(hv.w, 1), pc = VGX_DPP_SHR(cv.w, 1);    // the entry below the lane's first
            const int h63 = rowget_i32(hv.w, 15), c63 = rowget_i32(cv.w, 15);
            QV4 nh = hv, nc = cv;
            if (i0 + 0 == bq) { nh.x = ch; nc.x = cc; } else if (i0 + 0 > bq && i0 + 0 <= endq) { nh.x = ph; nc.x = pc; }
            if (i0 + 1 == bq) { nh.y = ch; nc.y = cc; } else if (i0 + 1 > bq && i0 + 1 <= endq) { nh.y = hv.x; nc.y = cv.x; }
            if (i0 + 2 == bq) { nh.z = ch; nc.z = cc; } else if (i0 + 2 > bq && i0 + 2 <= endq) { nh.z = hv.y; nc.z = cv.y; }
            if (i0 + 3 == bq) { nh.w = ch; nc.w = cc; } else if (i0 + 3 > bq && i0 + 3 <= endq) { nh.w = hv.z; nc.w = cv.z; }
            if (going && i0 + 3 >= bq && i0 <= endq) {
                if (tb + i0 + 3 < cap) {
                    *(QV4 *)(lh + tb + i0) = nh; *(QV4 *)(l3 + tb + i0) = nc;
                    if (BYTE8) *(uint32_t *)(l8 + tb + i0) = (uint32_t)min(nc.x, 255) | ((uint32_t)min(nc.y, 255) << 8) | ((uint32_t)min(nc.z, 255) << 16) |
                                                             ((uint32_t)min(nc.w, 255) << 24);
                } else {      // (a list of fewer than four slots: the lane's entries beyond it belong to another list)
                    if (i0 + 0 >= bq && i0 + 0 <= endq) { lh[tb + i0 + 0] = nh.x; l3[tb + i0 + 0] = nc.x; if (BYTE8) l8[tb + i0 + 0] = (uint8_t)min(nc.x, 255); }
                    if (i0 + 1 >= bq && i0 + 1 <= endq) { lh[tb + i0 + 1] = nh.y; l3[tb + i0 + 1] = nc.y; if (BYTE8) l8[tb + i0 + 1] = (uint8_t)min(nc.y, 255); }
                    if (i0 + 2 >= bq && i0 + 2 <= endq) { lh[tb + i0 + 2] = nh.z; l3[tb + i0 + 2] = nc.z; if (BYTE8) l8[tb + i0 + 2] = (uint8_t)min(nc.z, 255); }
                    if (i0 + 3 >= bq && i0 + 3 <= endq) { lh[tb + i0 + 3] = nh.w; l3[tb + i0 + 3] = nc.w; if (BYTE8) l8[tb + i0 + 3] = (uint8_t)min(nc.w, 255); }
                }
            }
            const bool cont = going && endq == 64;
            if (going && n > 64 && rl == 0) lt[tb >> 6] = tsum + cc - (cont ? c63 : 0);
            if (cont) { ch = h63; cc = c63; b = tb + 64; }
            else if (going) { if (zq >= 64) nn += 1; else zeros -= 1; going = false; }
        }
        if (ins_ok && nn != n && rl == 0) lc[n] = 0;
        WSYNC();
        if (ins_ok && n == 64 && nn == 65) {   // the list outgrows one tile: start its tile sums
            int64_t s0 = 0;
            for (int c4 = 0; c4 < 4; ++c4) s0 += rowget_i64(row_iscan((int64_t)l3[c4 * 16 + rl]), 15);
            if (rl == 0) { lt[0] = s0; lt[1] = l3[64]; }
        }
        if (ins_ok) n = nn;
        WSYNC();
    }
    return ok;
}
