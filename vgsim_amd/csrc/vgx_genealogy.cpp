// vgx_genealogy.cpp — backward (coalescent) pass over the event log: replaces BirthDeathModel.GetGenealogy
// (reference src/_BirthDeath.pyx:743-1000) and its recorders Mutations/Migrations (src/models.pxi:1-48).
//
// Host code of libvgx (the pass is one sequential walk, O(events); SURVEY.md §8f rank 1: "C++ first").  Unlike the
// reference, which allocates two std::vector per (population, haplotype) up front (P x H x 48 B: 200 MB at
// 65 536 x 64, 12 GB at 2^20 x 256) and sweeps all P x H compartments after every multievent row (pyx:988-994), the
// live lineages are kept per OCCUPIED compartment in a hash map and only compartments touched by a row are swept —
// same operations on the same vectors in the same order, so trees, times and records are identical.
// Random numbers: the wrapper's uniform() = PCG64 next_double (pyx:801 ...), numpy's random_hypergeometric on the
// same bit generator for MULTITYPE thinning (pyx:905, 933, 950, 956).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vgx.h"
#include "vgx_logfact.h"
#include "vgx_rng.h"

namespace {

enum { BIRTH = 0, DEATH = 1, SAMPLING = 2, MUTATION = 3, SUSCCHANGE = 4, MIGRATION = 5, MULTITYPE = 6 };  // ev:2-8

// numpy bit-generator front end on PCG64: next_double, next_uint64, buffered next_uint32 (pcg64.h)
struct Gen {
    VgxPcg64 g;
    bool has32;
    uint32_t spare;
    uint64_t next64() { return vgx_pcg64_next(g); }
    double uniform() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t next32() {
        if (has32) { has32 = false; return spare; }
        uint64_t x = next64();
        has32 = true;
        spare = (uint32_t)(x >> 32);
        return (uint32_t)x;
    }
    uint64_t interval(uint64_t max) {  // distributions.c random_interval: uniform on [0, max]
        if (max == 0) return 0;
        uint64_t mask = max, v;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffull) { while ((v = (next32() & mask)) > max) {} }
        else { while ((v = (next64() & mask)) > max) {} }
        return v;
    }
};

// ln(k!): table below 126, Stirling above — numpy logfactorial.c
double logfact(int64_t k) {
    if (k < 126) return vgx_logfact_table[k];
    const double halfln2pi = 0.9189385332046728;
    return (k + 0.5) * std::log((double)k) - k + (halfln2pi + (1.0 / k) * (1 / 12.0 - 1 / (360.0 * k * k)));
}

int64_t hyper_small(Gen &r, int64_t good, int64_t bad, int64_t sample) {  // hypergeometric_sample
    const int64_t total = good + bad;
    int64_t left = (sample > total / 2) ? total - sample : sample;
    int64_t rem_total = total, rem_good = good;
    while (left > 0 && rem_good > 0 && rem_total > rem_good) {
        --rem_total;
        if ((int64_t)r.interval((uint64_t)rem_total) < rem_good) --rem_good;
        --left;
    }
    if (rem_total == rem_good) rem_good -= left;
    return (sample > total / 2) ? rem_good : good - rem_good;
}

int64_t hyper_hrua(Gen &r, int64_t good, int64_t bad, int64_t sample) {  // hypergeometric_hrua (Stadlober 1990)
    const double D1 = 1.7155277699214135, D2 = 0.8989161620588988;
    const int64_t popsize = good + bad;
    const int64_t cs = std::min(sample, popsize - sample);
    const int64_t mn = std::min(good, bad), mx = std::max(good, bad);
    const double p = ((double)mn) / popsize, q = ((double)mx) / popsize;
    const double mu = cs * p, a = mu + 0.5;
    const double var = ((double)(popsize - cs) * cs * p * q / (popsize - 1));
    const double c = std::sqrt(var + 0.5), h = D1 * c + D2;
    const int64_t m = (int64_t)std::floor((double)(cs + 1) * (mn + 1) / (popsize + 2));
    const double g = logfact(m) + logfact(mn - m) + logfact(cs - m) + logfact(mx - cs + m);
    const double b = std::min((double)(std::min(cs, mn) + 1), std::floor(a + 16 * c));
    int64_t K;
    while (true) {
        double U = r.uniform(), V = r.uniform();
        double X = a + h * (V - 0.5) / U;
        if (X < 0.0 || X >= b) continue;
        K = (int64_t)std::floor(X);
        double gp = logfact(K) + logfact(mn - K) + logfact(cs - K) + logfact(mx - cs + K);
        double T = g - gp;
        if ((U * (4.0 - U) - 3.0) <= T) break;
        if (U * (U - T) >= 1) continue;
        if (2.0 * std::log(U) <= T) break;
    }
    if (good > bad) K = cs - K;
    if (cs < sample) K = good - K;
    return K;
}

int64_t hypergeometric(Gen &r, int64_t good, int64_t bad, int64_t sample) {  // random_hypergeometric
    if (sample >= 10 && sample <= good + bad - 10) return hyper_hrua(r, good, bad, sample);
    return hyper_small(r, good, bad, sample);
}

struct Pass {
    vgx_genealogy_io &io;
    const int64_t H;
    Gen rng;
    int64_t ptr = 0;
    std::unordered_map<int64_t, std::vector<int64_t>> live, fresh;  // lineages per compartment; arrivals of a row
    std::unordered_map<int64_t, int64_t> delta;                      // infectiousDelta of the current row
    std::string err;

    explicit Pass(vgx_genealogy_io &x) : io(x), H(x.hapNum) {}
    int64_t key(int64_t p, int64_t h) const { return p * H + h; }
    int64_t &inf(int64_t p, int64_t h) { return io.infectious[p * H + h]; }
    void node(int64_t pop, double t) { io.tree[ptr] = -1; io.tree_pop[ptr] = pop; io.times[ptr] = t; ptr += 1; }
    bool mutation(int64_t nodeId, int64_t hap, int64_t nh, double t) {  // models.pxi:13-29
        if (io.mut_n >= io.mut_cap) { err = "mutation record capacity exceeded"; return false; }
        int64_t d = nh > hap ? nh - hap : hap - nh, site = 0, digit4 = 1;
        while (d >= 4) { d /= 4; site += 1; digit4 *= 4; }
        int64_t k = io.mut_n++;
        io.mut_node[k] = nodeId; io.mut_DS[k] = (nh / digit4) % 4; io.mut_AS[k] = (hap / digit4) % 4;
        io.mut_site[k] = site; io.mut_time[k] = t;
        return true;
    }
    bool migration(int64_t nodeId, double t, int64_t oldPop, int64_t newPop) {  // models.pxi:44-48
        if (io.mig_n >= io.mig_cap) { err = "migration record capacity exceeded"; return false; }
        int64_t k = io.mig_n++;
        io.mig_node[k] = nodeId; io.mig_time[k] = t; io.mig_old[k] = oldPop; io.mig_new[k] = newPop;
        return true;
    }
    static void swap_pop(std::vector<int64_t> &v, int64_t i) { v[(size_t)i] = v.back(); v.pop_back(); }

    bool single(int64_t e) {  // one event of the direct path (pyx:797-872)
        const double t = io.ev_times[e];
        const int64_t type = io.ev_types[e], hap = io.ev_haplotypes[e], pop = io.ev_populations[e];
        const int64_t nh = io.ev_newHaplotypes[e], np = io.ev_newPopulations[e];
        switch (type) {
        case BIRTH: {
            auto it = live.find(key(pop, hap));
            const int64_t lbs = it == live.end() ? 0 : (int64_t)it->second.size(), lbs_e = inf(pop, hap);
            const double p = (double)lbs * ((double)lbs - 1.0) / (double)lbs_e / ((double)lbs_e - 1.0);
            if (rng.uniform() < p) {
                auto &b = it->second;
                int64_t n1 = (int64_t)std::floor(lbs * rng.uniform());
                int64_t n2 = (int64_t)std::floor((lbs - 1) * rng.uniform());
                if (n2 >= n1) n2 += 1;
                const int64_t id1 = b[(size_t)n1], id2 = b[(size_t)n2], id3 = ptr;
                b[(size_t)n1] = id3;
                swap_pop(b, n2);
                io.tree[id1] = id3; io.tree[id2] = id3;
                node(pop, t);
            }
            inf(pop, hap) -= 1;
            return true;
        }
        case DEATH: inf(pop, hap) += 1; return true;
        case SAMPLING:
            inf(pop, hap) += 1;
            live[key(pop, hap)].push_back(ptr);
            node(pop, t);
            return true;
        case MUTATION: {
            auto it = live.find(key(pop, nh));
            const int64_t lbs = it == live.end() ? 0 : (int64_t)it->second.size();
            const double p = (double)lbs / (double)inf(pop, nh);
            if (rng.uniform() < p) {
                int64_t n1 = (int64_t)std::floor(lbs * rng.uniform());
                const int64_t id1 = it->second[(size_t)n1];
                swap_pop(it->second, n1);
                live[key(pop, hap)].push_back(id1);
                if (!mutation(id1, hap, nh, t)) return false;
            }
            inf(pop, nh) -= 1;
            inf(pop, hap) += 1;
            return true;
        }
        case SUSCCHANGE: return true;
        case MIGRATION: {
            auto it = live.find(key(np, hap));
            const int64_t lbs = it == live.end() ? 0 : (int64_t)it->second.size();
            const double p = (double)lbs / (double)inf(np, hap);
            if (rng.uniform() < p) {
                int64_t nt = (int64_t)std::floor(lbs * rng.uniform());
                auto &bs = live[key(pop, hap)];
                auto &bt = live[key(np, hap)];  // (a rehash by the line above would invalidate `it`)
                const int64_t lbss = (int64_t)bs.size();
                const double p1 = (double)lbss / (double)inf(pop, hap);
                if (rng.uniform() < p1) {
                    int64_t ns = (int64_t)std::floor(lbss * rng.uniform());
                    const int64_t idt = bt[(size_t)nt], ids = bs[(size_t)ns], id3 = ptr;
                    bs[(size_t)ns] = id3;
                    swap_pop(bt, nt);
                    io.tree[idt] = id3; io.tree[ids] = id3;
                    node(pop, t);
                    if (!migration(idt, t, pop, np)) return false;
                } else {
                    bs.push_back(bt[(size_t)nt]);
                    swap_pop(bt, nt);
                }
            }
            inf(np, hap) -= 1;
            return true;
        }
        case MULTITYPE:
            for (int64_t r = hap; r < pop; ++r)
                if (!row(r)) return false;
            return true;
        default: err = "unknown event type " + std::to_string(type); return false;
        }
    }

    bool row(int64_t r) {  // one multievent row (pyx:873-994)
        if (!io.mev_num) { err = "MULTITYPE event without a multievent log"; return false; }
        if (r < 0 || r >= io.mev_rows) { err = "multievent row out of range"; return false; }
        const int64_t num = io.mev_num[r], type = io.mev_types[r], hap = io.mev_haplotypes[r], pop = io.mev_populations[r];
        const int64_t nh = io.mev_newHaplotypes[r], np = io.mev_newPopulations[r];
        const double t = io.mev_times[r];
        switch (type) {
        case BIRTH: {
            auto &b = live[key(pop, hap)];
            int64_t lbs = (int64_t)b.size();
            const int64_t lbs_e = inf(pop, hap);
            int64_t k = 0;
            if (!(num == 0 || lbs == 0))
                k = hypergeometric(rng, (int64_t)(lbs * (lbs - 1.0) / 2.0), (int64_t)(lbs_e * (lbs_e - 1) / 2 - lbs * (lbs - 1) / 2), num);
            for (int64_t i = 0; i < k; ++i) {
                int64_t n1 = (int64_t)std::floor(lbs * rng.uniform());
                int64_t n2 = (int64_t)std::floor((lbs - 1) * rng.uniform());
                if (n2 >= n1) n2 += 1;
                const int64_t id1 = b[(size_t)n1], id2 = b[(size_t)n2], id3 = ptr;
                fresh[key(pop, hap)].push_back(id3);
                if (n1 == lbs - 1) { b.pop_back(); b[(size_t)n2] = b[(size_t)(lbs - 2)]; b.pop_back(); }
                else if (n2 == lbs - 1) { b.pop_back(); b[(size_t)n1] = b[(size_t)(lbs - 2)]; b.pop_back(); }
                else { b[(size_t)n1] = b[(size_t)(lbs - 1)]; b.pop_back(); b[(size_t)n2] = b[(size_t)(lbs - 2)]; b.pop_back(); }
                io.tree[id1] = id3; io.tree[id2] = id3;
                node(pop, t);
                lbs -= 2;
            }
            delta[key(pop, hap)] -= num;
            break;
        }
        case DEATH: delta[key(pop, hap)] += num; break;
        case SAMPLING:
            delta[key(pop, hap)] += num;
            for (int64_t i = 0; i < num; ++i) { fresh[key(pop, hap)].push_back(ptr); node(pop, t); }
            break;
        case MUTATION: {
            auto &b = live[key(pop, nh)];
            int64_t lbs = (int64_t)b.size(), k = 0;
            if (!(num == 0 || lbs == 0)) k = hypergeometric(rng, lbs, inf(pop, nh) - lbs, num);
            for (int64_t i = 0; i < k; ++i) {
                int64_t n1 = (int64_t)std::floor(lbs * rng.uniform());
                const int64_t id1 = b[(size_t)n1];
                b[(size_t)n1] = b[(size_t)(lbs - 1)];
                b.pop_back();
                fresh[key(pop, hap)].push_back(id1);
                if (!mutation(id1, hap, nh, t)) return false;
                lbs -= 1;
            }
            delta[key(pop, nh)] -= num;
            delta[key(pop, hap)] += num;
            break;
        }
        case SUSCCHANGE: break;
        case MIGRATION: {
            auto &bs = live[key(pop, hap)];
            auto &bt = live[key(np, hap)];
            int64_t lbs = (int64_t)bt.size();
            if (!(num == 0 || lbs == 0)) {
                const int64_t k = hypergeometric(rng, lbs, inf(np, hap) - lbs, num);
                int64_t lbss = (int64_t)bs.size(), k2 = 0;
                if (!(k == 0 || lbss == 0)) k2 = hypergeometric(rng, lbss, inf(pop, hap) - lbss, k);
                for (int64_t i = 0; i < k2; ++i) {
                    int64_t nt = (int64_t)std::floor(lbs * rng.uniform());
                    int64_t ns = (int64_t)std::floor(lbss * rng.uniform());
                    const int64_t idt = bt[(size_t)nt], ids = bs[(size_t)ns], id3 = ptr;
                    swap_pop(bs, ns);
                    bt[(size_t)nt] = bt[(size_t)(lbs - 1)];
                    bt.pop_back();
                    fresh[key(pop, hap)].push_back(id3);
                    io.tree[idt] = id3; io.tree[ids] = id3;
                    node(pop, t);
                    if (!migration(idt, t, pop, np)) return false;
                    lbss -= 1;
                    lbs -= 1;
                }
                for (int64_t i = 0; i < k - k2; ++i) {
                    int64_t nt = (int64_t)std::floor(lbs * rng.uniform());
                    fresh[key(pop, hap)].push_back(bt[(size_t)nt]);
                    bt[(size_t)nt] = bt[(size_t)(lbs - 1)];
                    bt.pop_back();
                    lbs -= 1;
                }
            }
            delta[key(np, hap)] -= num;
            break;
        }
        default: err = "unknown multievent type " + std::to_string(type); return false;
        }
        // pyx:988-994, restricted to the compartments this row touched
        for (auto &d : delta) io.infectious[d.first] += d.second;
        delta.clear();
        for (auto &f : fresh) {
            auto &dst = live[f.first];
            while (!f.second.empty()) { dst.push_back(f.second.back()); f.second.pop_back(); }
        }
        fresh.clear();
        return true;
    }
};

}  // namespace

extern "C" int vgx_get_genealogy(vgx_genealogy_io *io, char *errbuf, int64_t errcap) {
    auto fail = [&](const std::string &m) {
        if (errbuf && errcap > 0) std::snprintf(errbuf, (size_t)errcap, "%s", m.c_str());
        return VGX_ERR_ARG;
    };
    if (!io || !io->infectious || !io->tree || !io->tree_pop || !io->times) return fail("vgx_get_genealogy: null argument");
    if (io->sCounter < 2) return fail("Less than two cases were sampled...");  // pyx:762-765
    const int64_t nodes = 2 * io->sCounter - 1;
    std::memset(io->tree, 0, (size_t)nodes * 8);
    std::memset(io->tree_pop, 0, (size_t)nodes * 8);
    std::memset(io->times, 0, (size_t)nodes * 8);
    io->mut_n = 0;
    io->mig_n = 0;
    Pass ps(*io);
    ps.rng.g.sh = io->rng_state[0]; ps.rng.g.sl = io->rng_state[1]; ps.rng.g.ih = io->rng_state[2]; ps.rng.g.il = io->rng_state[3];
    ps.rng.has32 = io->rng_has_uint32 != 0;
    ps.rng.spare = (uint32_t)io->rng_uinteger;
    for (int64_t e = io->ev_ptr - 1; e >= 0; --e) {
        if (ps.ptr >= nodes && (io->ev_types[e] == SAMPLING)) return fail("vgx_get_genealogy: more sampling events than sCounter");
        if (!ps.single(e)) return fail("vgx_get_genealogy: " + ps.err);
        if (ps.ptr > nodes) return fail("vgx_get_genealogy: tree overflow (event log and sCounter disagree)");
    }
    io->nodes_used = ps.ptr;
    for (int64_t i = 0; i < 2 * io->sCounter - 2; ++i) {  // pyx:998-1000
        const int64_t par = io->tree[i];
        if (par < 0 || par >= nodes) return fail("vgx_get_genealogy: lineage " + std::to_string(i) + " never coalesced (several roots)");
        if (io->tree_pop[par] != io->tree_pop[i] && !ps.migration(i, io->times[i], io->tree_pop[par], io->tree_pop[i]))
            return fail("vgx_get_genealogy: " + ps.err);
    }
    io->rng_state[0] = ps.rng.g.sh; io->rng_state[1] = ps.rng.g.sl; io->rng_state[2] = ps.rng.g.ih; io->rng_state[3] = ps.rng.g.il;
    io->rng_has_uint32 = ps.rng.has32 ? 1 : 0;
    io->rng_uinteger = ps.rng.spare;
    return VGX_OK;
}

// PCG64 (state, inc) of numpy's PCG64(SeedSequence(seed, spawn_key=(attempt,))) advanced by `draws` outputs: the
// position of the reference's self.seed after a simulate call that drew `draws` uniforms in its last attempt.
extern "C" void vgx_rng_position(int64_t seed, int64_t attempt, int64_t draws, uint64_t out[4]) {
    VgxPcg64 g;
    vgx_pcg64_seed(g, (uint64_t)seed, (uint32_t)attempt);
    // jump ahead: state' = A^n state + (A^n - 1)/(A - 1) inc, by repeated squaring (O(log n))
    uint64_t ah = 0x2360ED051FC65DA4ull, al = 0x4385DF649FCCF645ull;   // multiplier
    uint64_t ch = g.ih, cl = g.il;                                      // increment
    uint64_t acc_mh = 0, acc_ml = 1, acc_ph = 0, acc_pl = 0;
    uint64_t n = (uint64_t)draws;
    while (n > 0) {
        if (n & 1) {
            uint64_t th, tl;
            vgx_mul128(acc_mh, acc_ml, ah, al, th, tl); acc_mh = th; acc_ml = tl;
            vgx_mul128(acc_ph, acc_pl, ah, al, th, tl); acc_ph = th; acc_pl = tl;
            vgx_add128(acc_ph, acc_pl, ch, cl);
        }
        uint64_t th, tl, a1h = ah, a1l = al;
        vgx_add128(a1h, a1l, 0, 1);
        vgx_mul128(a1h, a1l, ch, cl, th, tl); ch = th; cl = tl;
        vgx_mul128(ah, al, ah, al, th, tl); ah = th; al = tl;
        n >>= 1;
    }
    uint64_t sh, sl;
    vgx_mul128(acc_mh, acc_ml, g.sh, g.sl, sh, sl);
    vgx_add128(sh, sl, acc_ph, acc_pl);
    out[0] = sh; out[1] = sl; out[2] = g.ih; out[3] = g.il;
}
