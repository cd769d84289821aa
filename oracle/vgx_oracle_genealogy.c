/*
 * vgx_oracle_genealogy.c — TEST INFRASTRUCTURE.  Op-for-op restatement of the reference's backward pass
 * BirthDeathModel.GetGenealogy (src/_BirthDeath.pyx:743-1000) with its recorders Mutations.AddMutation /
 * Migrations.AddMigration (src/models.pxi:13-29, 44-48), on caller-owned arrays.
 *
 * Random numbers: self.seed.uniform() (pyx:801-803, 831-833, 844-849, ...) = PCG64 next_double of the wrapper
 * restated in vgx_oracle.c, and numpy's random_hypergeometric(self.seed.rng, good, bad, sample) (pyx:905, 933,
 * 950, 956; numpy/random/src/distributions/random_hypergeometric.c, logfactorial.c, distributions.c:random_interval,
 * numpy 2.2) restated below and pinned against numpy.random.Generator(PCG64).hypergeometric in tests/test_rng.py.
 * PARITY STATUS: pinned on tests/golden/genealogy_*.npz (recorded from the reference build) for everything
 * downstream of the uniform stream; the seed -> stream mapping itself is unpinned (mc_lib absent, see vgx_oracle.h).
 */
#include "vgx_oracle.h"
#include "logfact_table.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- numpy bit-generator front end on PCG64: next_double, next_uint64, buffered next_uint32 ------------------ */
static uint64_t g_next64(vgo_gen_rng *r) { return vgo_pcg64_next64(&r->g); }
static double g_double(vgo_gen_rng *r) { return (double)(g_next64(r) >> 11) * (1.0 / 9007199254740992.0); }
static uint32_t g_next32(vgo_gen_rng *r) { /* numpy pcg64.h pcg64_next32 */
    if (r->has_uint32) {
        r->has_uint32 = 0;
        return (uint32_t)r->uinteger;
    }
    uint64_t next = g_next64(r);
    r->has_uint32 = 1;
    r->uinteger = (uint32_t)(next >> 32);
    return (uint32_t)(next & 0xffffffffu);
}

static uint64_t random_interval(vgo_gen_rng *r, uint64_t max) { /* distributions.c random_interval: [0, max] */
    uint64_t mask, value;
    if (max == 0) return 0;
    mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    if (max <= 0xffffffffUL) {
        while ((value = (g_next32(r) & mask)) > max) {}
    } else {
        while ((value = (g_next64(r) & mask)) > max) {}
    }
    return value;
}

static double logfactorial(int64_t k) { /* logfactorial.c */
    const double halfln2pi = 0.9189385332046728;
    if (k < 126) return vgo_logfact[k];
    return (k + 0.5) * log((double)k) - k + (halfln2pi + (1.0 / k) * (1 / 12.0 - 1 / (360.0 * k * k)));
}

static int64_t hypergeometric_sample(vgo_gen_rng *r, int64_t good, int64_t bad, int64_t sample) {
    int64_t remaining_total, remaining_good, result, computed_sample;
    int64_t total = good + bad;
    computed_sample = (sample > total / 2) ? total - sample : sample;
    remaining_total = total;
    remaining_good = good;
    while ((computed_sample > 0) && (remaining_good > 0) && (remaining_total > remaining_good)) {
        --remaining_total;
        if ((int64_t)random_interval(r, (uint64_t)remaining_total) < remaining_good) --remaining_good;
        --computed_sample;
    }
    if (remaining_total == remaining_good) remaining_good -= computed_sample;
    result = (sample > total / 2) ? remaining_good : good - remaining_good;
    return result;
}

#define HYP_D1 1.7155277699214135
#define HYP_D2 0.8989161620588988
#define MIN_(a, b) ((a) < (b) ? (a) : (b))
#define MAX_(a, b) ((a) > (b) ? (a) : (b))

static int64_t hypergeometric_hrua(vgo_gen_rng *r, int64_t good, int64_t bad, int64_t sample) {
    int64_t mingoodbad, maxgoodbad, popsize, computed_sample, m, K;
    double p, q, mu, var, a, c, b, h, g;
    popsize = good + bad;
    computed_sample = MIN_(sample, popsize - sample);
    mingoodbad = MIN_(good, bad);
    maxgoodbad = MAX_(good, bad);
    p = ((double)mingoodbad) / popsize;
    q = ((double)maxgoodbad) / popsize;
    mu = computed_sample * p;
    a = mu + 0.5;
    var = ((double)(popsize - computed_sample) * computed_sample * p * q / (popsize - 1));
    c = sqrt(var + 0.5);
    h = HYP_D1 * c + HYP_D2;
    m = (int64_t)floor((double)(computed_sample + 1) * (mingoodbad + 1) / (popsize + 2));
    g = (logfactorial(m) + logfactorial(mingoodbad - m) + logfactorial(computed_sample - m) +
         logfactorial(maxgoodbad - computed_sample + m));
    b = MIN_(MIN_(computed_sample, mingoodbad) + 1, floor(a + 16 * c));
    while (1) {
        double U, V, X, T, gp;
        U = g_double(r);
        V = g_double(r);
        X = a + h * (V - 0.5) / U;
        if ((X < 0.0) || (X >= b)) continue;
        K = (int64_t)floor(X);
        gp = (logfactorial(K) + logfactorial(mingoodbad - K) + logfactorial(computed_sample - K) +
              logfactorial(maxgoodbad - computed_sample + K));
        T = g - gp;
        if ((U * (4.0 - U) - 3.0) <= T) break;
        if (U * (U - T) >= 1) continue;
        if (2.0 * log(U) <= T) break;
    }
    if (good > bad) K = computed_sample - K;
    if (computed_sample < sample) K = good - K;
    return K;
}

int64_t vgo_hypergeometric(vgo_gen_rng *r, int64_t good, int64_t bad, int64_t sample) { /* random_hypergeometric */
    if ((sample >= 10) && (sample <= good + bad - 10)) return hypergeometric_hrua(r, good, bad, sample);
    return hypergeometric_sample(r, good, bad, sample);
}

/* ---- growable index vectors (std::vector<Py_ssize_t> of the reference) -------------------------------------- */
typedef struct { int64_t *v; int64_t n, cap; } ivec;
static void iv_push(ivec *a, int64_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 4;
        a->v = (int64_t *)realloc(a->v, (size_t)a->cap * 8);
    }
    a->v[a->n++] = x;
}
static void iv_pop(ivec *a) { a->n -= 1; }

static int add_mutation(vgo_genealogy *G, int64_t nodeId, int64_t haplotype, int64_t newHaplotype, double time) { /* models.pxi:13-29 */
    if (G->mut_n >= G->mut_cap) return 1;
    int64_t d = llabs(newHaplotype - haplotype), site = 0;
    while (d >= 4) { d = d / 4; site += 1; }
    int64_t digit4 = 1;
    for (int64_t i = 0; i < site; i++) digit4 *= 4;
    int64_t k = G->mut_n++;
    G->mut_node[k] = nodeId;
    G->mut_DS[k] = (newHaplotype / digit4) % 4;
    G->mut_AS[k] = (haplotype / digit4) % 4;
    G->mut_site[k] = site;
    G->mut_time[k] = time;
    return 0;
}
static int add_migration(vgo_genealogy *G, int64_t nodeId, double time, int64_t oldPop, int64_t newPop) { /* models.pxi:44-48 */
    if (G->mig_n >= G->mig_cap) return 1;
    int64_t k = G->mig_n++;
    G->mig_node[k] = nodeId; G->mig_time[k] = time; G->mig_old[k] = oldPop; G->mig_new[k] = newPop;
    return 0;
}

#define LB(p, h) (&live[(p) * H + (h)])
#define NL(p, h) (&fresh[(p) * H + (h)])
#define INF(p, h) (G->infectious[(p) * H + (h)])
#define DEL(p, h) (G->infectiousDelta[(p) * H + (h)])
#define NEW_NODE(pop, t)                 \
    do {                                 \
        G->tree[ptr] = -1;               \
        G->tree_pop[ptr] = (pop);        \
        G->times[ptr] = (t);             \
        ptr += 1;                        \
    } while (0)

int vgo_get_genealogy(vgo_genealogy *G) { /* pyx:743-1000 */
    const int64_t P = G->popNum, H = G->hapNum, s = G->sCounter;
    if (s < 2) return VGO_ERR_BAD_ARG; /* pyx:762-765: the reference prints a message and exits */
    const int64_t nodes = 2 * s - 1;
    int64_t ptr = 0, overflow = 0;
    vgo_gen_rng *R = &G->rng;
    memset(G->tree, 0, (size_t)nodes * 8);
    memset(G->tree_pop, 0, (size_t)nodes * 8);
    memset(G->times, 0, (size_t)nodes * 8);
    ivec *live = (ivec *)calloc((size_t)(P * H), sizeof(ivec));
    ivec *fresh = (ivec *)calloc((size_t)(P * H), sizeof(ivec));
    for (int64_t i = 0; i < P * H; i++) G->infectiousDelta[i] = 0;
    G->mut_n = 0; G->mig_n = 0;

    for (int64_t e_id = G->ev_ptr - 1; e_id >= 0 && !overflow; e_id--) {
        const double e_time = G->ev_times[e_id];
        const int64_t e_type = G->ev_types[e_id], e_hap = G->ev_haplotypes[e_id], e_pop = G->ev_populations[e_id];
        const int64_t e_nh = G->ev_newHaplotypes[e_id], e_np = G->ev_newPopulations[e_id];
        if (e_type == VGO_BIRTH) {
            ivec *b = LB(e_pop, e_hap);
            int64_t lbs = b->n, lbs_e = INF(e_pop, e_hap);
            double p = (double)lbs * ((double)lbs - 1.0) / (double)lbs_e / ((double)lbs_e - 1.0);
            if (g_double(R) < p) {
                int64_t n1 = (int64_t)floor(lbs * g_double(R));
                int64_t n2 = (int64_t)floor((lbs - 1) * g_double(R));
                if (n2 >= n1) n2 += 1;
                int64_t id1 = b->v[n1], id2 = b->v[n2], id3 = ptr;
                b->v[n1] = id3;
                b->v[n2] = b->v[lbs - 1];
                iv_pop(b);
                G->tree[id1] = id3; G->tree[id2] = id3;
                NEW_NODE(e_pop, e_time);
            }
            INF(e_pop, e_hap) -= 1;
        } else if (e_type == VGO_DEATH) {
            INF(e_pop, e_hap) += 1;
        } else if (e_type == VGO_SAMPLING) {
            INF(e_pop, e_hap) += 1;
            iv_push(LB(e_pop, e_hap), ptr);
            NEW_NODE(e_pop, e_time);
        } else if (e_type == VGO_MUTATION) {
            ivec *b = LB(e_pop, e_nh);
            int64_t lbs = b->n;
            double p = (double)lbs / (double)INF(e_pop, e_nh);
            if (g_double(R) < p) {
                int64_t n1 = (int64_t)floor(lbs * g_double(R));
                int64_t id1 = b->v[n1];
                b->v[n1] = b->v[lbs - 1];
                iv_pop(b);
                iv_push(LB(e_pop, e_hap), id1);
                overflow |= add_mutation(G, id1, e_hap, e_nh, e_time);
            }
            INF(e_pop, e_nh) -= 1;
            INF(e_pop, e_hap) += 1;
        } else if (e_type == VGO_SUSCCHANGE) {
        } else if (e_type == VGO_MIGRATION) {
            ivec *bt = LB(e_np, e_hap);
            int64_t lbs = bt->n;
            double p = (double)lbs / (double)INF(e_np, e_hap);
            if (g_double(R) < p) {
                int64_t nt = (int64_t)floor(lbs * g_double(R));
                ivec *bs = LB(e_pop, e_hap);
                int64_t lbss = bs->n;
                double p1 = (double)lbss / (double)INF(e_pop, e_hap);
                if (g_double(R) < p1) {
                    int64_t ns = (int64_t)floor(lbss * g_double(R));
                    int64_t idt = bt->v[nt], ids = bs->v[ns], id3 = ptr;
                    bs->v[ns] = id3;
                    bt->v[nt] = bt->v[lbs - 1];
                    iv_pop(bt);
                    G->tree[idt] = id3; G->tree[ids] = id3;
                    NEW_NODE(e_pop, e_time);
                    overflow |= add_migration(G, idt, e_time, e_pop, e_np);
                } else {
                    iv_push(bs, bt->v[nt]);
                    bt->v[nt] = bt->v[lbs - 1];
                    iv_pop(bt);
                }
            }
            INF(e_np, e_hap) -= 1;
        } else if (e_type == VGO_MULTITYPE) {
            for (int64_t me = e_hap; me < e_pop && !overflow; me++) {
                const int64_t num = G->mev_num[me], mtype = G->mev_types[me], mh = G->mev_haplotypes[me];
                const int64_t mp = G->mev_populations[me], mnh = G->mev_newHaplotypes[me], mnp = G->mev_newPopulations[me];
                const double mt = G->mev_times[me];
                if (mtype == VGO_BIRTH) {
                    ivec *b = LB(mp, mh);
                    int64_t lbs = b->n, lbs_e = INF(mp, mh), k;
                    if (num == 0 || lbs == 0) k = 0;
                    else k = vgo_hypergeometric(R, (int64_t)(lbs * (lbs - 1.0) / 2.0),
                                                (int64_t)(lbs_e * (lbs_e - 1) / 2 - lbs * (lbs - 1) / 2), num);
                    for (int64_t i = 0; i < k; i++) {
                        int64_t n1 = (int64_t)floor(lbs * g_double(R));
                        int64_t n2 = (int64_t)floor((lbs - 1) * g_double(R));
                        if (n2 >= n1) n2 += 1;
                        int64_t id1 = b->v[n1], id2 = b->v[n2], id3 = ptr;
                        iv_push(NL(mp, mh), id3);
                        if (n1 == lbs - 1) {
                            iv_pop(b);
                            b->v[n2] = b->v[lbs - 2];
                            iv_pop(b);
                        } else if (n2 == lbs - 1) {
                            iv_pop(b);
                            b->v[n1] = b->v[lbs - 2];
                            iv_pop(b);
                        } else {
                            b->v[n1] = b->v[lbs - 1];
                            iv_pop(b);
                            b->v[n2] = b->v[lbs - 2];
                            iv_pop(b);
                        }
                        G->tree[id1] = id3; G->tree[id2] = id3;
                        NEW_NODE(mp, mt);
                        lbs -= 2;
                    }
                    DEL(mp, mh) -= num;
                } else if (mtype == VGO_DEATH) {
                    DEL(mp, mh) += num;
                } else if (mtype == VGO_SAMPLING) {
                    DEL(mp, mh) += num;
                    for (int64_t i = 0; i < num; i++) {
                        iv_push(NL(mp, mh), ptr);
                        NEW_NODE(mp, mt);
                    }
                } else if (mtype == VGO_MUTATION) {
                    ivec *b = LB(mp, mnh);
                    int64_t lbs = b->n, k;
                    if (num == 0 || lbs == 0) k = 0;
                    else k = vgo_hypergeometric(R, lbs, INF(mp, mnh) - lbs, num);
                    for (int64_t i = 0; i < k; i++) {
                        int64_t n1 = (int64_t)floor(lbs * g_double(R));
                        int64_t id1 = b->v[n1];
                        b->v[n1] = b->v[lbs - 1];
                        iv_pop(b);
                        iv_push(NL(mp, mh), id1);
                        overflow |= add_mutation(G, id1, mh, mnh, mt);
                        lbs -= 1;
                    }
                    DEL(mp, mnh) -= num;
                    DEL(mp, mh) += num;
                } else if (mtype == VGO_SUSCCHANGE) {
                } else if (mtype == VGO_MIGRATION) {
                    ivec *bt = LB(mnp, mh);
                    int64_t lbs = bt->n, k;
                    if (num == 0 || lbs == 0) {
                        k = 0;
                    } else {
                        k = vgo_hypergeometric(R, lbs, INF(mnp, mh) - lbs, num);
                        /* everything below sits inside the else branch upstream (pyx:951-982) */
                        ivec *bs = LB(mp, mh);
                        int64_t lbss = bs->n, k2;
                        if (k == 0 || lbss == 0) k2 = 0;
                        else k2 = vgo_hypergeometric(R, lbss, INF(mp, mh) - lbss, k);
                        for (int64_t i = 0; i < k2; i++) {
                            int64_t nt = (int64_t)floor(lbs * g_double(R));
                            int64_t ns = (int64_t)floor(lbss * g_double(R));
                            int64_t idt = bt->v[nt], ids = bs->v[ns], id3 = ptr;
                            bs->v[ns] = bs->v[bs->n - 1];
                            iv_pop(bs);
                            bt->v[nt] = bt->v[lbs - 1];
                            iv_pop(bt);
                            iv_push(NL(mp, mh), id3);
                            G->tree[idt] = id3; G->tree[ids] = id3;
                            NEW_NODE(mp, mt);
                            overflow |= add_migration(G, idt, mt, mp, mnp);
                            lbss -= 1;
                            lbs -= 1;
                        }
                        for (int64_t i = 0; i < k - k2; i++) {
                            int64_t nt = (int64_t)floor(lbs * g_double(R));
                            iv_push(NL(mp, mh), bt->v[nt]);
                            bt->v[nt] = bt->v[lbs - 1];
                            iv_pop(bt);
                            lbs -= 1;
                        }
                    }
                    DEL(mnp, mh) -= num;
                } else {
                    overflow = 2;
                }
                /* pyx:988-994: after EVERY multievent row */
                for (int64_t pi = 0; pi < P; pi++)
                    for (int64_t hi = 0; hi < H; hi++) {
                        INF(pi, hi) += DEL(pi, hi);
                        DEL(pi, hi) = 0;
                        ivec *nl = NL(pi, hi);
                        while (nl->n > 0) {
                            iv_push(LB(pi, hi), nl->v[nl->n - 1]);
                            iv_pop(nl);
                        }
                    }
            }
        } else {
            overflow = 2;
        }
    }
    G->nodes_used = ptr;
    /* pyx:998-1000 */
    for (int64_t i = 0; i < 2 * s - 2 && !overflow; i++) {
        int64_t par = G->tree[i];
        if (par < 0 || par >= nodes) { overflow = 3; break; } /* upstream would index out of bounds */
        if (G->tree_pop[par] != G->tree_pop[i]) overflow |= add_migration(G, i, G->times[i], G->tree_pop[par], G->tree_pop[i]);
    }
    for (int64_t i = 0; i < P * H; i++) { free(live[i].v); free(fresh[i].v); }
    free(live); free(fresh);
    return overflow ? VGO_ERR_BAD_ARG : VGO_OK;
}
