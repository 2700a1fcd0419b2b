// Host-side plan of the LDS-staged column sweep (include/sgcn.h, sgcn_ldsplan_t): virtual rows -> tiles of
// NW x RW rows inside row groups -> per tile: nonzeros sorted by sweep position of their column, columns referenced
// at least `min_reuse` times get a ring slot (chunks of S slots), everything else goes to the residual CSR.
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <numeric>
#include <queue>
#include <thread>
#include <vector>

struct sgcn_ldsplan_host {
    int32_t VW, NW, RW, S, U, M, K, nparts;
    std::vector<int32_t> tile_chunk_ptr, chunk_cols, chunk_hdr, tile_rows, tile_slots;
    std::vector<int64_t> ent_ptr;
    std::vector<uint32_t> words;
    std::vector<float> vals, row_fold;
    int32_t unit = 0;
    int32_t xcd_tile_ptr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<sgcn_fix_t> fix;
    int64_t nslots = 0, nent = 0, staged = 0;
    std::vector<int32_t> res_rowptr, res_col;
    std::vector<float> res_val;
};

namespace {

struct VRow { int32_t row, piece, npieces, nnz; };

// Longest-processing-time dealing with a capacity: the next heaviest item goes to the lightest bin that still has a
// free place.  Returns bin * cap + place -> item (or -1).
// `speed` (nullable, per bin, per cent): a bin's load counts as load * 100 / speed -- bins that work faster get more.
std::vector<int64_t> deal(const std::vector<int64_t>& weight, int64_t nbins, int32_t cap, const int32_t* speed = nullptr) {
    std::vector<int64_t> assign((size_t)nbins * cap, -1);
    std::vector<int32_t> fill((size_t)nbins, 0);
    std::vector<int64_t> load((size_t)nbins, 0);
    typedef std::pair<int64_t, int64_t> WB;
    std::priority_queue<WB, std::vector<WB>, std::greater<WB>> heap;
    for (int64_t b = 0; b < nbins; b++) heap.push({0, b});
    for (size_t i = 0; i < weight.size(); i++) {          // `weight` is sorted heaviest first by the caller
        WB top = heap.top();
        heap.pop();
        const int64_t b = top.second;
        assign[(size_t)b * cap + fill[b]++] = (int64_t)i;
        load[(size_t)b] += weight[i];
        if (fill[b] < cap) heap.push({speed ? load[(size_t)b] * 10000 / speed[b] : load[(size_t)b], b});
    }
    return assign;
}

struct Edge { int32_t pos, col, w, lr; float val; int32_t row; };

struct TileOut {
    std::vector<int32_t> chunk_cols;                // nchunks * S
    std::vector<std::vector<uint32_t>> wave_word;   // per wave: entry words, chunk after chunk
    std::vector<std::vector<float>> wave_val;       // ... and values
    std::vector<std::vector<int64_t>> wave_cnt;     // per wave: entry words per chunk (padded)
    std::vector<std::vector<int32_t>> wave_pairs;   // per wave: how many of them (the first ones) are PAIR words, per chunk
    std::vector<Edge> residual;
    int64_t staged = 0;
};

}  // namespace

extern "C" {

int sgcn_ldsplan_create(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t K,
                        const int32_t* col_pos, const int32_t* row_group, int32_t VW, int32_t T, int32_t min_reuse,
                        int32_t mode, int32_t ring_slots, sgcn_ldsplan_host_t** out) {
    if (!out || M < 0 || K < 0 || (M > 0 && (!rowptr || !col || !val)))
        return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: bad argument");
    if (VW != 2) return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: VW must be 2 (128-column slabs)");
    if (min_reuse < 1) min_reuse = 1;
    if (T <= 0) T = 2048;
    if (ring_slots <= 0) ring_slots = 128;
    if (ring_slots != 80 && ring_slots != 128)
        return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: ring_slots is 80 (three ring parts) or 128 (two)");
    const int32_t NW = 8, RW = 192 / VW, S = ring_slots, GE = 8, NPART = ring_slots == 80 ? 3 : 2;
    const int32_t R = NW * RW;
    const uint32_t piece = 256u * (uint32_t)VW;
    const uint32_t zero_addr = (uint32_t)NPART * S * piece;
    for (int32_t r = 0; r < M; r++)
        if (rowptr[r + 1] < rowptr[r]) return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: rowptr not monotone at %d", r);
    if (row_group)
        for (int32_t r = 0; r < M; r++)
            if (row_group[r] < 0) return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: negative group label at row %d", r);
    const int64_t nnz = M > 0 ? (int64_t)rowptr[M] - rowptr[0] : 0;
    for (int64_t p = 0; p < nnz; p++)
        if (col[rowptr[0] + p] < 0 || col[rowptr[0] + p] >= K)
            return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: column out of range");
    if (col_pos) {
        std::vector<uint8_t> seen((size_t)K, 0);
        for (int32_t c = 0; c < K; c++) {
            if (col_pos[c] < 0 || col_pos[c] >= K || seen[(size_t)col_pos[c]])
                return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: col_pos is not a permutation");
            seen[(size_t)col_pos[c]] = 1;
        }
    }
    auto* h = new sgcn_ldsplan_host();
    h->VW = VW; h->NW = NW; h->RW = RW; h->S = S; h->U = GE; h->M = M; h->K = K; h->nparts = NPART;

    // unit plan: every nonzero of a row carries the same value (bit for bit) -> kept once per row, beside the row scale
    h->row_fold.assign((size_t)M, 1.0f);
    h->unit = mode == 0 ? 1 : 0;
    for (int32_t r = 0; r < M && h->unit; r++) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        if (e > b) h->row_fold[(size_t)r] = val[b];
        for (int32_t p = b + 1; p < e; p++)
            if (std::memcmp(&val[p], &val[b], 4) != 0) { h->unit = 0; break; }
    }
    if (!h->unit) std::fill(h->row_fold.begin(), h->row_fold.end(), 1.0f);
    // entries per wave and chunk the kernel's entry ring holds: 4 registers of 64 words (unit), 2 of words + 2 of values
    const int32_t CAP = h->unit ? 256 : 128;

    // split rows -> workspace slots, consecutive per row, in row order (the fix-up adds them in this order)
    std::vector<int32_t> first_slot((size_t)M, -1);
    int32_t slot = 0;
    for (int32_t r = 0; r < M; r++) {
        const int32_t n = rowptr[r + 1] - rowptr[r];
        if (n <= T) continue;
        const int32_t c = (n + T - 1) / T;
        first_slot[r] = slot;
        h->fix.push_back(sgcn_fix_t{r, slot, c});
        slot += c;
    }
    h->nslots = slot;

    // rows bucketed by group (stable); tiles never straddle groups
    std::vector<int32_t> order((size_t)M);
    std::vector<int64_t> gptr;
    if (!row_group) {
        std::iota(order.begin(), order.end(), 0);
        gptr = {0, (int64_t)M};
    } else {
        int32_t ng = 0;
        for (int32_t r = 0; r < M; r++) ng = std::max(ng, row_group[r] + 1);
        gptr.assign((size_t)ng + 1, 0);
        for (int32_t r = 0; r < M; r++) gptr[(size_t)row_group[r] + 1]++;
        for (int32_t g = 0; g < ng; g++) gptr[(size_t)g + 1] += gptr[g];
        std::vector<int64_t> fillp(gptr.begin(), gptr.end() - 1);
        for (int32_t r = 0; r < M; r++) order[(size_t)fillp[row_group[r]]++] = r;
    }

    // virtual rows -> tiles (LPT inside the group) -> waves (LPT inside the tile)
    struct TileRows { std::vector<VRow> vr; std::vector<int32_t> wave, lr; };   // per tile: its virtual rows and places
    std::vector<TileRows> tiles;
    for (size_t g = 0; g + 1 < gptr.size(); g++) {
        std::vector<VRow> v;
        for (int64_t i = gptr[g]; i < gptr[g + 1]; i++) {
            const int32_t r = order[(size_t)i];
            const int32_t n = rowptr[r + 1] - rowptr[r];
            if (n <= T) { v.push_back({r, 0, 1, n}); continue; }
            const int32_t c = (n + T - 1) / T;
            for (int32_t q = 0; q < c; q++) v.push_back({r, q, c, (n - q + c - 1) / c});
        }
        if (v.empty()) continue;
        std::stable_sort(v.begin(), v.end(), [](const VRow& x, const VRow& y) { return x.nnz > y.nnz; });
        const int64_t nt = ((int64_t)v.size() + R - 1) / R;
        std::vector<int64_t> w(v.size());
        for (size_t i = 0; i < v.size(); i++) w[i] = v[i].nnz;
        const std::vector<int64_t> ta = deal(w, nt, R);
        for (int64_t t = 0; t < nt; t++) {
            TileRows tr;
            for (int32_t q = 0; q < R; q++) {
                const int64_t i = ta[(size_t)t * R + q];
                if (i >= 0) tr.vr.push_back(v[(size_t)i]);              // heaviest first (deal keeps the order)
            }
            std::vector<int64_t> ww(tr.vr.size());
            for (size_t i = 0; i < tr.vr.size(); i++) ww[i] = tr.vr[i].nnz;
            // (knob lds_wave_bias, per cent, default 100 = even shares: waves 0-3 of a workgroup are the OLDER wave of their SIMD
            // and issue first -- without the s_setprio around the kernel's update chain they finished a chunk 24 % ahead of
            // waves 4-7 and waited at its barrier; with it the two are level and the bias buys nothing)
            int32_t speed[8];
            for (int32_t wv = 0; wv < NW; wv++) speed[wv] = wv < NW / 2 ? (int32_t)sgcn_tune_get("lds_wave_bias") : 100;
            const std::vector<int64_t> wa = deal(ww, NW, RW, speed);
            tr.wave.assign(tr.vr.size(), 0);
            tr.lr.assign(tr.vr.size(), 0);
            for (int32_t wv = 0; wv < NW; wv++)
                for (int32_t q = 0; q < RW; q++) {
                    const int64_t i = wa[(size_t)wv * RW + q];
                    if (i >= 0) { tr.wave[(size_t)i] = wv; tr.lr[(size_t)i] = q; }
                }
            tiles.push_back(std::move(tr));
        }
    }
    const int64_t ntiles = (int64_t)tiles.size();
    h->tile_rows.assign((size_t)ntiles * R, -1);
    h->tile_slots.assign((size_t)ntiles * R, -1);
    for (int64_t t = 0; t < ntiles; t++) {
        const TileRows& tr = tiles[(size_t)t];
        for (size_t i = 0; i < tr.vr.size(); i++) {
            const size_t place = ((size_t)t * NW + tr.wave[i]) * RW + tr.lr[i];
            h->tile_rows[place] = tr.vr[i].row;
            h->tile_slots[place] = tr.vr[i].npieces > 1 ? first_slot[tr.vr[i].row] + tr.vr[i].piece : -1;
        }
    }

    // per tile (independent: a few host threads): sort by sweep position, count references per column, chunk
    std::vector<TileOut> outs((size_t)ntiles);
    std::atomic<int> overflow{0};
    auto build_tile = [&](int64_t t) {
        const TileRows& tr = tiles[(size_t)t];
        TileOut& o = outs[(size_t)t];
        std::vector<Edge> e;
        for (size_t i = 0; i < tr.vr.size(); i++) {
            const VRow& vr = tr.vr[i];
            const int32_t base = rowptr[vr.row], n = rowptr[vr.row + 1] - base;
            for (int32_t j = vr.piece; j < n; j += vr.npieces) {          // strided pieces: each spans the whole sweep
                const int32_t c = col[base + j];
                e.push_back(Edge{col_pos ? col_pos[c] : c, c, tr.wave[i], tr.lr[i], val[base + j], vr.row});
            }
        }
        // by sweep position; ties (the same column): by wave, then in row-stream order (stable) -- deterministic
        std::stable_sort(e.begin(), e.end(), [](const Edge& x, const Edge& y) { return x.pos < y.pos; });
        o.wave_word.assign((size_t)NW, {});
        o.wave_val.assign((size_t)NW, {});
        o.wave_cnt.assign((size_t)NW, {});
        o.wave_pairs.assign((size_t)NW, {});
        // the open chunk, per wave: PAIR words (unit plans: two rows of the wave on one staged piece -- one LDS read, two
        // updates; row offsets in bits 0-7 and 24-31) and single words (+ values); a chunk's words = pairs, padded to whole
        // groups, then singles, padded
        std::vector<std::vector<uint32_t>> pw((size_t)NW), sw((size_t)NW);
        std::vector<std::vector<float>> sv((size_t)NW);
        std::vector<int64_t> add((size_t)NW, 0);
        const bool pairs = h->unit != 0;
        const bool mix = sgcn_tune_get("lds_mix") != 0;
        constexpr int64_t kMaxPairWords = 128;             // the kernel's pair path covers the first 16 groups of a chunk
        auto padded = [&](size_t n) { return (int64_t)((n + GE - 1) / GE * GE); };
        int32_t nslot = 0;                   // slots used in the open chunk
        int64_t nchunks = 0;
        auto close_chunk = [&]() {
            if (nslot == 0) return;
            const int32_t last = o.chunk_cols.back();
            for (int32_t s = nslot; s < S; s++) o.chunk_cols.push_back(last);      // unused slots fetch a valid column
            for (int32_t wv = 0; wv < NW; wv++) {
                std::vector<uint32_t>& W = o.wave_word[(size_t)wv];
                std::vector<float>& V = o.wave_val[(size_t)wv];
                int64_t c = 0;
                for (uint32_t x : pw[(size_t)wv]) { W.push_back(x); V.push_back(0.f); c++; }
                while (c % GE) { W.push_back(zero_addr); V.push_back(0.f); c++; }   // pads: the zero piece (onto row 0)
                o.wave_pairs[(size_t)wv].push_back((int32_t)c);
                for (size_t q = 0; q < sw[(size_t)wv].size(); q++) { W.push_back(sw[(size_t)wv][q]); V.push_back(sv[(size_t)wv][q]); c++; }
                while (c % GE) { W.push_back(zero_addr); V.push_back(0.f); c++; }
                o.wave_cnt[(size_t)wv].push_back(c);
                pw[(size_t)wv].clear(); sw[(size_t)wv].clear(); sv[(size_t)wv].clear();
            }
            nslot = 0;
            nchunks++;
        };
        // Staging order.  Columns the tile uses three times or more ("hot": its own community's) and the ones it uses once
        // or twice ("cold": staged only by plans with min_reuse < 3, i.e. when there is no residual sweep) are DEALT into
        // the chunks in proportion, each kind in sweep order: a chunk of cold columns alone has ~16 entries per wave and
        // waits for its fabric-bound pieces (fill wait 9-17 % of the sweep of an all-staged plan), a chunk of hot ones has
        // ~160 and hides them -- mixed, every chunk has the arithmetic to cover its requests.  (Which chunk a column is staged
        // in changes the order of a row's additions, nothing else.)
        std::vector<std::pair<size_t, size_t>> hot, cold, all, order;
        for (size_t i = 0; i < e.size();) {
            size_t j = i;
            while (j < e.size() && e[j].pos == e[i].pos) j++;
            if ((int64_t)(j - i) < min_reuse) {
                for (size_t q = i; q < j; q++) o.residual.push_back(e[q]);
            } else {
                ((j - i) >= 3 || !mix ? hot : cold).push_back({i, j});
                all.push_back({i, j});
            }
            i = j;
        }
        {
            // Only a tile with no more cold columns than hot ones is mixed (knob lds_mix, 0 = never); any other keeps the plain
            // sweep order.  Measured on S-Reddit-SBM, everything staged (profiles/r37_lds_mix.txt): 0.7 cold per hot (p_in
            // 0.95) 1.74 -> 1.60 ms; at 1.25 per hot (p_in 0.9) and 2.45 (p_in 0.8) every form of mixing tried -- in
            // proportion, capped with the rest behind or left in place -- loses 2-4 %.
            if (!mix || cold.empty() || hot.empty() || cold.size() > hot.size()) {
                order = all;
            } else {
                size_t ih = 0, ic = 0;
                const size_t tot = hot.size() + cold.size();
                for (size_t t = 0; t < tot; t++) {
                    const bool take_cold = ic < cold.size() && (ih >= hot.size() || ic * tot < (t + 1) * cold.size());
                    order.push_back(take_cold ? cold[ic++] : hot[ih++]);
                }
            }
        }
        for (const auto& grp : order) {
            const size_t i = grp.first, j = grp.second;
            {
                // a chunk holds at most S columns and at most CAP entry words per wave (what the kernel loads per chunk),
                // at most kMaxPairWords of them pairs
                std::fill(add.begin(), add.end(), 0);
                for (size_t q = i; q < j; q++) add[(size_t)e[q].w]++;
                bool over = nslot == S;
                for (int32_t wv = 0; wv < NW; wv++) {
                    const int64_t np2 = pairs ? add[(size_t)wv] / 2 : 0, ns2 = add[(size_t)wv] - 2 * np2;
                    over = over || padded(pw[(size_t)wv].size() + (size_t)np2) + padded(sw[(size_t)wv].size() + (size_t)ns2) > CAP ||
                           (int64_t)pw[(size_t)wv].size() + np2 > kMaxPairWords;
                    if (np2 > kMaxPairWords || padded((size_t)np2) + padded((size_t)ns2) > CAP) overflow = 1;   // (one column, more entries of one wave than the ring holds: duplicates)
                }
                if (over) close_chunk();
                const uint32_t addr = (uint32_t)(nchunks % NPART) * S * piece + (uint32_t)nslot * piece;
                o.chunk_cols.push_back(e[i].col);
                nslot++;
                o.staged++;
                // the column's entries wave by wave (they arrive sorted by position only): two at a time into a pair word
                for (int32_t wv = 0; wv < NW; wv++) {
                    if (!add[(size_t)wv]) continue;
                    int32_t held = -1;
                    for (size_t q = i; q < j; q++) {
                        if (e[q].w != wv) continue;
                        if (!pairs) {
                            sw[(size_t)wv].push_back(addr | (uint32_t)(e[q].lr * VW));
                            sv[(size_t)wv].push_back(e[q].val);
                        } else if (held < 0) {
                            held = e[q].lr;
                        } else {
                            pw[(size_t)wv].push_back(addr | (uint32_t)(held * VW) | ((uint32_t)(e[q].lr * VW) << 24));
                            held = -1;
                        }
                    }
                    if (held >= 0) { sw[(size_t)wv].push_back(addr | (uint32_t)(held * VW)); sv[(size_t)wv].push_back(0.f); }
                }
            }
        }
        close_chunk();
    };
    {
        const unsigned nthr = (unsigned)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, ntiles}));
        std::atomic<int64_t> next{0};
        auto work = [&]() { for (int64_t t; (t = next.fetch_add(1)) < ntiles;) build_tile(t); };
        std::vector<std::thread> pool;
        for (unsigned q = 1; q < nthr; q++) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }

    if (overflow) {
        delete h;
        return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: a column holds more entries of one wave than a chunk may (duplicate nonzeros?)");
    }
    // assemble: chunks tile after tile; entries in (tile, wave, chunk) order
    h->tile_chunk_ptr.assign((size_t)ntiles + 1, 0);
    int64_t nchunks = 0, nent = 0;
    for (int64_t t = 0; t < ntiles; t++) {
        nchunks += (int64_t)outs[(size_t)t].chunk_cols.size() / S;
        h->tile_chunk_ptr[(size_t)t + 1] = (int32_t)nchunks;
        for (int32_t wv = 0; wv < NW; wv++) nent += (int64_t)outs[(size_t)t].wave_word[(size_t)wv].size();
        h->staged += outs[(size_t)t].staged;
    }
    if (nchunks >= (1ll << 31) / S) {
        delete h;
        return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_create: plan too large");
    }
    h->nent = nent;
    h->chunk_cols.reserve((size_t)nchunks * S);
    h->ent_ptr.assign((size_t)nchunks * NW + 1, 0);
    h->words.reserve((size_t)(nent + 256));
    h->vals.reserve((size_t)(nent + 256));
    int64_t e0 = 0;
    for (int64_t t = 0; t < ntiles; t++) {
        const TileOut& o = outs[(size_t)t];
        h->chunk_cols.insert(h->chunk_cols.end(), o.chunk_cols.begin(), o.chunk_cols.end());
        const int64_t cb = h->tile_chunk_ptr[(size_t)t], nc = h->tile_chunk_ptr[(size_t)t + 1] - cb;
        for (int32_t wv = 0; wv < NW; wv++) {
            for (int64_t k = 0; k < nc; k++) {
                h->ent_ptr[(size_t)(cb * NW + wv * nc + k)] = e0;
                e0 += o.wave_cnt[(size_t)wv][(size_t)k];
            }
            h->words.insert(h->words.end(), o.wave_word[(size_t)wv].begin(), o.wave_word[(size_t)wv].end());
            h->vals.insert(h->vals.end(), o.wave_val[(size_t)wv].begin(), o.wave_val[(size_t)wv].end());
        }
    }
    h->ent_ptr[(size_t)nchunks * NW] = e0;
    // contiguous tile ranges for the 8 XCDs, balanced by estimated TIME, not by count: a tile costs its entries (per wave)
    // plus a fixed amount per chunk (measured: ~39 cycles per entry of a wave, ~1800 per chunk for barrier, fill issue and
    // pipeline start), and tiles of large sparse communities have four times the chunks of small dense ones
    {
        std::vector<double> w((size_t)ntiles);
        double tot = 0;
        for (int64_t t = 0; t < ntiles; t++) {
            int64_t ne = 0;
            for (int32_t wv = 0; wv < NW; wv++) ne += (int64_t)outs[(size_t)t].wave_word[(size_t)wv].size();
            w[(size_t)t] = 39.0 * (double)ne / NW + 1800.0 * (double)(h->tile_chunk_ptr[(size_t)t + 1] - h->tile_chunk_ptr[(size_t)t]) + 20000.0;
            tot += w[(size_t)t];
        }
        double acc = 0;
        int64_t t = 0;
        for (int x = 1; x < 8; x++) {
            while (t < ntiles && acc + 0.5 * w[(size_t)t] < tot * x / 8.0) acc += w[(size_t)t++];
            h->xcd_tile_ptr[x] = (int32_t)t;
        }
        h->xcd_tile_ptr[0] = 0;
        h->xcd_tile_ptr[8] = (int32_t)ntiles;
    }
    // per (chunk, wave) header, 32 ints: what a wave needs to request a chunk, in ONE 128-byte load -- the column ids of
    // its S / NW ring slots (words 0-15), its entry count in groups (16), the position of its entries (17, 18) and how many of
    // the groups hold pair words (19)
    {
        const int32_t per = S / NW;                          // slots a wave fetches per chunk
        h->chunk_hdr.assign((size_t)nchunks * NW * 32, 0);
        for (int64_t t = 0; t < ntiles; t++) {
            const int64_t cb = h->tile_chunk_ptr[(size_t)t], nc = h->tile_chunk_ptr[(size_t)t + 1] - cb;
            for (int64_t k = 0; k < nc; k++)
                for (int32_t wv = 0; wv < NW; wv++) {
                    int32_t* hd = &h->chunk_hdr[(size_t)((cb + k) * NW + wv) * 32];
                    for (int32_t q = 0; q < per; q++) hd[q] = h->chunk_cols[(size_t)(cb + k) * S + (size_t)wv * per + q];
                    const int64_t a = h->ent_ptr[(size_t)(cb * NW + wv * nc + k)], b = h->ent_ptr[(size_t)(cb * NW + wv * nc + k) + 1];
                    hd[16] = (int32_t)((b - a) / GE);
                    hd[19] = outs[(size_t)t].wave_pairs[(size_t)wv][(size_t)k] / GE;       // the first hd[19] groups are pair words
                    hd[17] = (int32_t)(uint32_t)(a & 0xffffffffll);
                    hd[18] = (int32_t)(a >> 32);
                }
        }
    }
    for (int32_t q = 0; q < 256; q++) { h->words.push_back(zero_addr); h->vals.push_back(0.f); }   // the kernel's over-read

    // residual CSR: rows in order, a row's nonzeros by column
    h->res_rowptr.assign((size_t)M + 1, 0);
    int64_t rn = 0;
    for (const TileOut& o : outs) {
        rn += (int64_t)o.residual.size();
        for (const Edge& x : o.residual) h->res_rowptr[(size_t)x.row + 1]++;
    }
    for (int32_t r = 0; r < M; r++) h->res_rowptr[(size_t)r + 1] += h->res_rowptr[(size_t)r];
    h->res_col.resize((size_t)rn);
    h->res_val.resize((size_t)rn);
    {
        std::vector<int32_t> fillp(h->res_rowptr.begin(), h->res_rowptr.end() - 1);
        for (const TileOut& o : outs)
            for (const Edge& x : o.residual) {
                const int32_t p = fillp[(size_t)x.row]++;
                h->res_col[(size_t)p] = x.col;
                h->res_val[(size_t)p] = x.val;
            }
        std::vector<std::pair<int32_t, float>> buf;
        for (int32_t r = 0; r < M; r++) {
            const int32_t a = h->res_rowptr[(size_t)r], b = h->res_rowptr[(size_t)r + 1];
            if (b - a < 2) continue;
            buf.clear();
            for (int32_t p = a; p < b; p++) buf.push_back({h->res_col[(size_t)p], h->res_val[(size_t)p]});
            std::stable_sort(buf.begin(), buf.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            for (int32_t p = a; p < b; p++) { h->res_col[(size_t)p] = buf[(size_t)(p - a)].first; h->res_val[(size_t)p] = buf[(size_t)(p - a)].second; }
        }
    }
    *out = h;
    return SGCN_OK;
}

int sgcn_ldsplan_sizes(const sgcn_ldsplan_host_t* h, int64_t* s) {
    if (!h || !s) return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_sizes: null argument");
    s[0] = (int64_t)h->tile_chunk_ptr.size() - 1;
    s[1] = h->tile_chunk_ptr.back();
    s[2] = h->nent;
    s[3] = (int64_t)h->fix.size();
    s[4] = h->nslots;
    s[5] = (int64_t)h->res_col.size();
    s[6] = h->staged;
    s[7] = h->unit;
    for (int x = 0; x < 9; x++) s[8 + x] = h->xcd_tile_ptr[x];
    s[17] = h->S;
    s[18] = h->nparts;
    return SGCN_OK;
}

int sgcn_ldsplan_export(const sgcn_ldsplan_host_t* h, int32_t* tile_chunk_ptr, int32_t* chunk_cols, int32_t* chunk_hdr,
                        int64_t* ent_ptr, uint32_t* words, float* vals, float* row_fold, int32_t* tile_rows,
                        int32_t* tile_slots, sgcn_fix_t* fix, int32_t* res_rowptr, int32_t* res_col, float* res_val) {
    if (!h) return sgcn::fail(SGCN_ERR_INVALID, "ldsplan_export: null plan");
    auto cp = [](void* dst, const void* src, size_t bytes) { if (dst && bytes) std::memcpy(dst, src, bytes); };
    cp(tile_chunk_ptr, h->tile_chunk_ptr.data(), h->tile_chunk_ptr.size() * 4);
    cp(chunk_cols, h->chunk_cols.data(), h->chunk_cols.size() * 4);
    cp(chunk_hdr, h->chunk_hdr.data(), h->chunk_hdr.size() * 4);
    cp(ent_ptr, h->ent_ptr.data(), h->ent_ptr.size() * 8);
    cp(words, h->words.data(), h->words.size() * 4);
    cp(vals, h->vals.data(), h->vals.size() * 4);
    cp(row_fold, h->row_fold.data(), h->row_fold.size() * 4);
    cp(tile_rows, h->tile_rows.data(), h->tile_rows.size() * 4);
    cp(tile_slots, h->tile_slots.data(), h->tile_slots.size() * 4);
    cp(fix, h->fix.data(), h->fix.size() * sizeof(sgcn_fix_t));
    cp(res_rowptr, h->res_rowptr.data(), h->res_rowptr.size() * 4);
    cp(res_col, h->res_col.data(), h->res_col.size() * 4);
    cp(res_val, h->res_val.data(), h->res_val.size() * 4);
    return SGCN_OK;
}

void sgcn_ldsplan_destroy(sgcn_ldsplan_host_t* h) { delete h; }

}  // extern "C"
