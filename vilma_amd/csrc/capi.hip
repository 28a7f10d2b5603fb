// C-ABI of libvilma_hip.so (see include/vilma_hip.h): context, LD store, state and the
// evaluation entry points.  Device memory is owned by the context (hipMalloc); kernels run on
// the stream the caller passes.
#include "ctx.h"
#include "detmath.h"

namespace vilma_detail { std::string g_create_error; }

namespace {

void free_items(ItemSet &it) {
    dev_free(it.sym); dev_free(it.comb); dev_free(it.tile); dev_free(it.tcomb); dev_free(it.a); dev_free(it.row); dev_free(it.rcomb);
    for (int k = 0; k < 5; ++k) dev_free(it.eig[k]);
    dev_free(it.eig_all);
    dev_free(it.fcomb);
    it = ItemSet();
}

void free_ready(vilma_ctx *c) {
    for (int s = 0; s < 3; ++s) { dev_free(c->pool[s]); c->pool[s] = nullptr; }
    dev_free(c->dot_partials); c->dot_partials = nullptr;
    dev_free(c->sym_scratch); c->sym_scratch = nullptr;
    free_items(c->all);
    for (auto &it : c->solo) free_items(it);
    c->solo.clear();
    c->ready = false;
    c->have_moments = false;
}

struct HostItems {
    std::vector<SymItem> sym;
    std::vector<SymCombItem> comb;
    std::vector<SymTile> tile;
    std::vector<TileCombItem> tcomb;
    std::vector<LdItem> a;
    std::vector<RowItem> row;
    std::vector<RowCombItem> rcomb;
    std::vector<EigenGroup> groups;
    int64_t group_bytes = 0;          // U bytes in the group being filled
    std::vector<EigItem> eig[5];
    std::vector<RowCombItem> fcomb;
};

void make_items(const vilma_ctx *c, int p, const CohortLd &co, int32_t t_base, int32_t s_base,
                int32_t &slot, HostItems &H) {
    const int32_t PN = (int32_t)(c->P * c->N), pN = (int32_t)(p * c->N);
    int32_t s_off = s_base;
    for (const BlockRec &b : co.blocks) {
        if (b.form == 0 && c->tile_rows > 0) {
            // tiled product: row strip I x column strip K (lower triangle of the strip grid)
            const int TR = c->tile_rows, TS = c->tile_slabs, CW = 128 * TS, q = TR / CW;
            const int G = (b.n + TR - 1) / TR, ns = n_slabs(b.n), sn = pad2(b.n);
            if (b.n <= CW) {             // (CW <= TR) one item takes the block: y written directly
                SymTile it;
                it.a = co.store + b.off_a; it.n = b.n; it.x_off = pN + b.start;
                it.R0 = 0; it.R1 = b.n; it.J0 = 0; it.nJ = ns; it.out0 = 0;
                it.row_off = PN + pN + b.start; it.col_off = -1; it.direct = 1 + slot++;
                H.tile.push_back(it);
                continue;
            }
            for (int I = 0; I < G; ++I) {
                const int R0 = I * TR, R1 = std::min(b.n, R0 + TR);
                for (int K = 0; K * CW < R1; ++K) {
                    const int c0 = K * CW;
                    const bool diag = c0 >= R0;
                    SymTile it;
                    it.a = co.store + b.off_a; it.n = b.n; it.x_off = pN + b.start;
                    it.R0 = R0; it.R1 = R1; it.J0 = K * TS;
                    // slabs that begin below the item's last row have nothing in it
                    it.nJ = std::min(std::min(TS, ns - it.J0), (R1 - c0 + 127) / 128);
                    it.out0 = std::max(R0, c0);
                    it.row_off = s_off + K * sn + it.out0;
                    it.col_off = diag ? -1 : s_off + (K + I - K / q) * sn + c0;
                    it.direct = 0;
                    H.tile.push_back(it);
                }
            }
            for (int j0 = 0; j0 < b.n; j0 += 256) {
                TileCombItem cb;
                cb.n = b.n; cb.s_base = s_off; cb.y_off = PN + pN + b.start;
                cb.dot_off = pN + b.start; cb.dot_slot = slot++; cb.j0 = j0;
                cb.cw = CW; cb.tr = TR; cb.G = G;
                H.tcomb.push_back(cb);
            }
            s_off += tile_scratch_elems(b.n, TR, CW);
        } else if (b.form == 0) {
            int64_t off = b.off_a;
            const int ns = n_slabs(b.n);
            const int CH = c->chunk_rows;
            const int nch_max = (b.n + CH - 1) / CH;
            const int32_t c_base = s_off + ns * pad2(b.n);   // C[slab][chunk][128] behind S[slab][pad2(n)]
            for (int J = 0; J < ns; ++J) {
                const int rows = b.n - 128 * J, wJ = slab_width(b.n, J), ldJ = pad_ld(wJ);
                for (int r0 = 0, ch = 0; r0 < rows; r0 += CH, ++ch) {
                    SymItem it;
                    it.a = co.store + off + (int64_t)r0 * ldJ;
                    it.rows = std::min(CH, rows - r0); it.w = wJ; it.ld = ldJ; it.j0 = 128 * J;
                    it.x_off = pN + b.start; it.s_off = s_off + J * pad2(b.n); it.r0 = r0;
                    it.c_off = c_base + (J * nch_max + ch) * 128;
                    H.sym.push_back(it);
                }
                off += sym_panel_elems(b.n, J);
            }
            for (int j0 = 0; j0 < b.n; j0 += 256) {         // one workgroup per 256 columns
                SymCombItem cb;
                cb.n = b.n; cb.s_base = s_off; cb.y_off = PN + pN + b.start;
                cb.dot_off = pN + b.start; cb.dot_slot = slot++; cb.j0 = j0;
                cb.c_base = c_base; cb.nch_max = nch_max; cb.chunk_rows = CH;
                H.comb.push_back(cb);
            }
            s_off += sym_scratch_elems(b.n, CH);
        } else if (b.w > 0) {
            // eigen form, fused product: column-major U, one item per slab of columns
            const int R = b.w, cols = eig_slab_cols(b.n, b.r, R);
            const int ns = (b.r + cols - 1) / cols, ldc = pad2(b.n);
            const double *U = co.store + b.off_a, *sv = co.store + b.off_v;
            if (ns == 1) {               // one slab: the item writes y and the block's y.z partial
                EigItem it;
                it.a = U; it.scale = sv; it.n = b.n; it.ncols = b.r; it.ldc = ldc;
                it.x_off = pN + b.start; it.s_off = PN + pN + b.start; it.direct = 1 + slot++;
                H.eig[eig_class(R)].push_back(it);
                continue;
            }
            for (int J = 0; J < ns; ++J) {
                EigItem it;
                const int c0 = J * cols;
                it.a = U + (int64_t)c0 * ldc;
                it.scale = sv + c0;
                it.n = b.n; it.ncols = std::min(cols, b.r - c0); it.ldc = ldc;
                it.x_off = pN + b.start; it.s_off = s_off + J * pad2(b.n); it.direct = 0;
                H.eig[eig_class(R)].push_back(it);
            }
            for (int i0 = 0; i0 < b.n; i0 += 256) {
                RowCombItem cb;
                cb.n = b.n; cb.ns = ns; cb.s_base = s_off; cb.y_off = PN + pN + b.start;
                cb.dot_off = pN + b.start; cb.dot_slot = slot++; cb.i0 = i0; cb.pad = 0;
                H.fcomb.push_back(cb);
            }
            s_off += ns * pad2(b.n);
        } else {
            // eigen form: U [n x ld(r)] then s [ld(r)].  New group when this block would push the
            // group's U past the budget (a block larger than the budget gets a group of its own)
            const int ldr = pad_ld(b.r), ns = (b.r + 127) / 128, CH = c->chunk_rows;
            const int64_t ubytes = (int64_t)b.n * ldr * 8;
            if (H.groups.empty() || H.group_bytes + ubytes > c->eigen_group_bytes) {
                H.groups.push_back({(int)H.a.size(), 0, (int)H.row.size(), 0, (int)H.rcomb.size(), 0});
                H.group_bytes = 0;
            }
            H.group_bytes += ubytes;
            EigenGroup &g = H.groups.back();
            const double *U = co.store + b.off_a, *sv = co.store + b.off_v;
            const int32_t t_pool = 2 * PN + t_base + b.t_off;
            for (int c0 = 0; c0 < b.r; c0 += 128) {        // t' = s * (U^T x)
                LdItem it;
                it.a = U; it.rows = b.n; it.ld = ldr; it.col0 = c0;
                it.ncols = b.r; it.x_off = pN + b.start; it.y_off = t_pool;
                it.dot_off = -1; it.dot_slot = -1; it.scale = sv;
                H.a.push_back(it);
                ++g.na;
            }
            for (int J = 0; J < ns; ++J)                   // S[J][i] = sum_{c in slab J} U[i][c] t'[c]
                for (int r0 = 0; r0 < b.n; r0 += CH) {
                    RowItem it;
                    it.a = U + (int64_t)r0 * ldr + 128 * J; it.rows = std::min(CH, b.n - r0);
                    it.ld = ldr; it.w = std::min(128, b.r - 128 * J); it.t_off = t_pool + 128 * J;
                    it.s_off = s_off + J * pad2(b.n) + r0; it.pad = 0;
                    H.row.push_back(it);
                    ++g.nr;
                }
            for (int i0 = 0; i0 < b.n; i0 += 256) {        // y = sum_J S[J], and y.z
                RowCombItem cb;
                cb.n = b.n; cb.ns = ns; cb.s_base = s_off; cb.y_off = PN + pN + b.start;
                cb.dot_off = pN + b.start; cb.dot_slot = slot++; cb.i0 = i0; cb.pad = 0;
                H.rcomb.push_back(cb);
                ++g.nc;
            }
            s_off += ns * pad2(b.n);
        }
    }
}

// order of the symmetric product's work items (vilma_ctx::ld_order): 0 = longest first (default);
// 1 = the order the panels lie in the store; 2 = store order dealt out so that the workgroups one
// XCD receives under round-robin dispatch (b, b + 8, ...) walk ONE contiguous eighth of the list
void sort_items(HostItems &H, int order) {
    // longest first: workgroup run time ~ rows streamed (within each eigen group for its lists)
    for (const EigenGroup &g : H.groups) {
        std::stable_sort(H.a.begin() + g.a0, H.a.begin() + g.a0 + g.na,
                         [](const LdItem &x, const LdItem &y) {
            return (int64_t)x.rows * std::min(128, x.ncols - x.col0) >
                   (int64_t)y.rows * std::min(128, y.ncols - y.col0);
        });
        std::stable_sort(H.row.begin() + g.r0, H.row.begin() + g.r0 + g.nr,
                         [](const RowItem &x, const RowItem &y) {
            return (int64_t)x.rows * x.w > (int64_t)y.rows * y.w;
        });
    }
    // (tiled product: elements an item streams, longest first)
    auto tile_elems = [](const SymTile &t) {
        int64_t e = 0;
        for (int J = t.J0; J < t.J0 + t.nJ; ++J)
            e += (int64_t)(t.R1 - std::max(t.R0, 128 * J)) * pad_ld(slab_width(t.n, J));
        return e;
    };
    if (order == 0)
        std::stable_sort(H.tile.begin(), H.tile.end(), [&](const SymTile &x, const SymTile &y) {
            return tile_elems(x) > tile_elems(y);
        });
    if (order == 0) {
        std::stable_sort(H.sym.begin(), H.sym.end(), [](const SymItem &x, const SymItem &y) {
            return (int64_t)x.rows * x.ld > (int64_t)y.rows * y.ld;
        });
    } else if (order == 2 && H.sym.size() > 8) {
        const size_t n = H.sym.size(), per = (n + 7) / 8;
        std::vector<SymItem> out;
        out.reserve(n);
        for (size_t q = 0; q < 8 * per; ++q) {
            const size_t src = (q % 8) * per + q / 8;
            if (src < n) out.push_back(H.sym[src]);
        }
        H.sym.swap(out);
    }
    for (auto &v : H.eig)
        std::stable_sort(v.begin(), v.end(), [](const EigItem &x, const EigItem &y) {
            return (int64_t)x.n * x.ncols > (int64_t)y.n * y.ncols;
        });
}

template <typename T>
int upload_vec(vilma_ctx *c, const std::vector<T> &v, T **dev, int *count) {
    *dev = nullptr;
    *count = (int)v.size();
    if (v.empty()) return 0;
    HIPCHK(c, hipMalloc((void **)dev, v.size() * sizeof(T)));
    HIPCHK(c, hipMemcpy(*dev, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int upload_items(vilma_ctx *c, HostItems &H, ItemSet &out) {
    sort_items(H, c->ld_order);
    out.groups = H.groups;
    std::vector<EigItem> eig_all;
    for (int k = 0; k < 5; ++k) {
        if (upload_vec(c, H.eig[k], &out.eig[k], &out.n_eig[k])) return 1;
        // (the tall class needs workgroups of 512 threads: always a launch of its own)
        if (k < 4) eig_all.insert(eig_all.end(), H.eig[k].begin(), H.eig[k].end());
    }
    std::stable_sort(eig_all.begin(), eig_all.end(), [](const EigItem &x, const EigItem &y) {
        return (int64_t)x.n * x.ncols > (int64_t)y.n * y.ncols;
    });
    if (upload_vec(c, eig_all, &out.eig_all, &out.n_eig_all)) return 1;
    if (upload_vec(c, H.fcomb, &out.fcomb, &out.n_fcomb)) return 1;
    if (upload_vec(c, H.tile, &out.tile, &out.n_tile) || upload_vec(c, H.tcomb, &out.tcomb, &out.n_tcomb))
        return 1;
    return upload_vec(c, H.sym, &out.sym, &out.n_sym) || upload_vec(c, H.comb, &out.comb, &out.n_comb) ||
           upload_vec(c, H.a, &out.a, &out.n_a) || upload_vec(c, H.row, &out.row, &out.n_row) ||
           upload_vec(c, H.rcomb, &out.rcomb, &out.n_rcomb);
}

// scratch entries one dense block needs per right-hand side
int64_t dense_scratch_elems(const vilma_ctx *c, int n) {
    if (c->tile_rows <= 0) return sym_scratch_elems(n, c->chunk_rows);
    const int cw = 128 * c->tile_slabs;
    return n <= cw ? 0 : tile_scratch_elems(n, c->tile_rows, cw);
}

// Width of the tiled product's column strips for THIS shard.  The wider an item, the fewer
// workgroups store partial sums (what holds the kernel under the bare read of the store,
// profiles/r05i_ld_tiles.txt) -- but a launch wants a few items per workgroup slot to balance, and a
// shard with fewer items than slots runs for as long as its longest item streams.  Measured on one
// box (full C3 / its 2-, 4-, 8-way shards / C2): 4 slabs win from about 2 items per slot, 2 slabs
// between, 1 slab (the decomposition of ld_sym_kernel, merged stores) on the small ones.
void choose_tile(vilma_ctx *c) {
    if (!c->tile_auto || c->tile_rows <= 0) return;
    const int64_t slots = 4 * (int64_t)c->n_cu;            // four workgroups of four waves per CU
    for (int ts : {4, 2}) {
        const int tr = 512, cw = 128 * ts;
        int64_t items = 0;
        for (const CohortLd &co : c->ld)
            for (const BlockRec &b : co.blocks) {
                if (b.form != 0) continue;
                if (b.n <= cw) { ++items; continue; }
                for (int R0 = 0; R0 < b.n; R0 += tr)
                    items += (std::min(b.n, R0 + tr) + cw - 1) / cw;   // column strips left of the strip's end
            }
        if (items >= 2 * slots) { c->tile_rows = tr; c->tile_slabs = ts; return; }
    }
    c->tile_rows = 512;
    c->tile_slabs = 1;
}

// the device-resident work lists of every cohort's product (and of all cohorts together)
int build_item_lists(vilma_ctx *c, std::vector<int32_t> &dstart, int32_t &slot, int32_t &t_base,
                     int32_t &s_base) {
    HostItems all;
    dstart.assign(c->P + 1, 0);
    slot = 0; t_base = 0; s_base = 0;
    free_items(c->all);
    for (auto &it : c->solo) free_items(it);
    c->solo.clear();
    c->solo.resize(c->P);
    for (int p = 0; p < c->P; ++p) {
        dstart[p] = slot;
        HostItems one;
        int32_t s0 = slot;
        make_items(c, p, c->ld[p], t_base, s_base, s0, one);
        make_items(c, p, c->ld[p], t_base, s_base, slot, all);     // same items, same slots
        if (upload_items(c, one, c->solo[p])) return 1;
        t_base += c->ld[p].t_used;
        s_base += c->ld[p].s_used;
    }
    dstart[c->P] = slot;
    return upload_items(c, all, c->all);
}

int ensure_ready(vilma_ctx *c) {
    if (c->ready) return 0;
    for (int p = 0; p < c->P; ++p)
        if (!c->ld[p].ended) return fail(c, "LD for cohort " + std::to_string(p) + " not loaded");
    free_ready(c);
    dev_free(c->repack_tmp);             // load-time staging of the eigen-form repack: done with
    c->repack_tmp = nullptr;
    c->repack_elems = 0;
    choose_tile(c);
    for (CohortLd &co : c->ld) {
        co.s_used = co.s_eig_used;
        for (const BlockRec &b : co.blocks)
            if (b.form == 0) co.s_used += dense_scratch_elems(c, b.n);
    }
    // work items carry 32-bit offsets into the vector pool and the scratch: check the totals
    // BEFORE any offset is formed
    {
        int64_t t_total = 0, s_total = 0;
        for (int p = 0; p < c->P; ++p) { t_total += c->ld[p].t_used; s_total += c->ld[p].s_used; }
        const int64_t pool_total = 2 * (int64_t)c->P * c->N + t_total + 2;
        if (pool_total >= (int64_t)1 << 31)
            return fail(c, "shard too large: the vector pool (2 P N + eigen-form scratch = " +
                           std::to_string(pool_total) + " doubles) must stay below 2^31 because "
                           "work items carry 32-bit offsets; shard the SNPs over more GPUs");
        if (s_total >= (int64_t)1 << 31)
            return fail(c, "shard too large: the scratch of the symmetric LD product (" +
                           std::to_string(s_total) + " doubles) must stay below 2^31; shard the "
                           "SNPs over more GPUs");
    }
    std::vector<int32_t> dstart;
    int32_t slot = 0, t_base = 0, s_base = 0;
    if (build_item_lists(c, dstart, slot, t_base, s_base)) return 1;
    const int64_t pool_elems = 2 * (int64_t)c->P * c->N + t_base + 2;
    c->pool_elems = pool_elems;
    for (int s = 0; s < 3; ++s)
        if (dev_alloc(c, &c->pool[s], pool_elems)) return 1;
    // y.z partials and product scratch for two right-hand sides (a two-step beta trial)
    c->dot_stride = slot;
    c->s_stride = s_base;
    if (dev_alloc(c, &c->dot_partials, 2 * (int64_t)slot)) return 1;
    if (dev_alloc(c, &c->sym_scratch, 2 * (int64_t)s_base)) return 1;
    c->dot_start = dstart;
    c->ready = true;
    return 0;
}

// Profiling events come from a recycled pool (creating events per launch costs microseconds of
// host time inside the region being measured).
hipEvent_t prof_event(vilma_ctx *c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        c->prof = 0;                // stop bracketing rather than record into a null event
        c->prof_now = false;
        return nullptr;
    }
    return e;
}
void prof_begin(vilma_ctx *c, hipStream_t s, hipEvent_t &e0) {
    e0 = nullptr;
    if (!c->prof_now) return;
    e0 = prof_event(c);
    if (e0) (void)hipEventRecord(e0, s);
}
void prof_end(vilma_ctx *c, hipStream_t s, hipEvent_t e0, int kind) {
    if (!c->prof_now || !e0) return;
    hipEvent_t e1 = prof_event(c);
    if (!e1) { c->event_pool.push_back(e0); return; }
    (void)hipEventRecord(e1, s);
    c->pending.push_back({e0, e1, kind, c->prof_tag});
}
void prof_resolve(vilma_ctx *c) {
    for (auto &pr : c->pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(pr.e1);
        if (hipEventElapsedTime(&ms, pr.e0, pr.e1) == hipSuccess) {
            c->prof_ms[pr.kind] += ms;
            c->prof_launches[pr.kind] += 1;
        }
        c->event_pool.push_back(pr.e0);
        c->event_pool.push_back(pr.e1);
    }
    c->pending.clear();
}

// the LD product on pool `pl` (x_ld section -> y_ld section), all cohorts or one
// the LD product on pool `pl` (x_ld section -> y_ld section), all cohorts or one; with pl2 the
// same product for a second right-hand side in the same pass over the LD store (every element is
// loaded once for both)
void run_ld(vilma_ctx *c, hipStream_t s, double *pl, double *pl2, int cohort) {
    const ItemSet &it = cohort < 0 ? c->all : c->solo[cohort];
    hipEvent_t e0;
    c->prof_now = c->prof > 0 && (c->prof_tick++ % c->prof) == 0;
    if (it.n_sym > 0) {
        prof_begin(c, s, e0);
        launch_ld_sym(it.sym, it.n_sym, pl, pl2, c->sym_scratch, c->s_stride, s);
        prof_end(c, s, e0, pl2 ? VILMA_PROF_LD_SYM2 : VILMA_PROF_LD_SYM);
    }
    if (it.n_tile > 0) {
        prof_begin(c, s, e0);
        launch_ld_tile(it.tile, it.n_tile, c->tile_slabs, pl, pl2, c->sym_scratch, c->s_stride,
                       c->dot_partials, c->dot_stride, s);
        prof_end(c, s, e0, pl2 ? VILMA_PROF_LD_SYM2 : VILMA_PROF_LD_SYM);
    }
    // eigen-form blocks, group by group (one group unless VILMA_EIGEN_GROUP_MB says otherwise):
    // both passes over the group's U back to back; one bracket around all = one product
    if (!it.groups.empty() || it.n_fcomb > 0 || it.n_eig_all > 0 || it.n_eig[4] > 0) {
        prof_begin(c, s, e0);
        // fused product (U read once): one launch per block-height class present -- or, on a small
        // shard, all classes in one launch -- then the combine
        if (it.n_eig_all < c->eig_merge_below)
            launch_ld_eig_fused_all(it.eig_all, it.n_eig_all, pl, pl2, c->sym_scratch, c->s_stride,
                                    c->dot_partials, c->dot_stride, s);
        else
            for (int k = 0; k < 4; ++k) {
                if (k == 0 && c->eig_wave)
                    launch_ld_eig_wave(it.eig[0], it.n_eig[0], pl, pl2, c->sym_scratch, c->s_stride,
                                       c->dot_partials, c->dot_stride, s);
                else
                    launch_ld_eig_fused(it.eig[k], it.n_eig[k], eig_class_rows(k), pl, pl2,
                                        c->sym_scratch, c->s_stride, c->dot_partials, c->dot_stride, s);
            }
        launch_ld_eig_tall(it.eig[4], it.n_eig[4], pl, pl2, c->sym_scratch, c->s_stride,
                           c->dot_partials, c->dot_stride, s);
        launch_ld_rowsum_combine(it.fcomb, it.n_fcomb, pl, pl2, c->sym_scratch, c->s_stride,
                                 c->dot_partials, c->dot_stride, s);
        for (const EigenGroup &g : it.groups) {
            launch_ld_colsum(it.a + g.a0, g.na, pl, pl2, /*keep=*/it.groups.size() > 1, s);
            launch_ld_rowsum(it.row + g.r0, g.nr, pl, pl2, c->sym_scratch, c->s_stride, s);
            launch_ld_rowsum_combine(it.rcomb + g.c0, g.nc, pl, pl2, c->sym_scratch, c->s_stride,
                                     c->dot_partials, c->dot_stride, s);
        }
        prof_end(c, s, e0, VILMA_PROF_LD_EIG);
    }
    if (it.n_tcomb > 0)
        launch_ld_tile_combine(it.tcomb, it.n_tcomb, pl, pl2, c->sym_scratch, c->s_stride,
                               c->dot_partials, c->dot_stride, s);
    if (it.n_comb > 0)      // not bracketed: tiny, and every event pair costs host time
        launch_ld_sym_combine(it.comb, it.n_comb, pl, pl2, c->sym_scratch, c->s_stride,
                              c->dot_partials, c->dot_stride, s);
    if (c->pending.size() > 8192) prof_resolve(c);
}

void fill_snp_args(vilma_ctx *c, SnpKernelArgs &a, double step) {
    const int cur = c->mom_cur, tr = c->mom_ta, tb = c->mom_tb;
    a.N = (int32_t)c->N; a.M = c->M; a.A = c->A; a.P = c->P;
    a.mu_in = c->mu[c->mu_cur];
    a.mu_out = c->mu[c->mu_ta];
    a.mu_out2 = c->mu[c->mu_tb];
    a.pool_out2 = c->pool[tb]; a.m_out2 = c->m[tb]; a.v_out2 = c->v[tb]; a.lse_out2 = c->lse[tb];
    a.step2 = 0.0;
    a.adj = c->adj; a.se = c->se; a.sld = c->sld; a.annot = c->annot; a.invperm = c->invperm;
    a.prec = c->prec; a.log_det = c->log_det; a.lh = c->lh;
    a.pool_cur = c->pool[cur]; a.m_cur = c->m[cur];
    a.pool_out = c->pool[tr]; a.m_out = c->m[tr]; a.v_out = c->v[tr]; a.lse_out = c->lse[tr];
    a.partials = c->snp_partials;
    a.pp = nullptr;
    a.no_store = 0;
    a.lse_ref = c->have_moments ? c->lse[cur] : nullptr;
    a.sum_partials = nullptr;
    a.scal = c->scal; a.snapshot = c->snap[c->snap_cur]; a.snapshot_out = c->snap[c->snap_cur];
    a.diff = 0;
    a.step = step;
    for (int p = 0; p < VILMA_MAX_P; ++p) a.tau.v[p] = p < c->P ? c->tau[p] : 1.0;
}

// trial_mu / trial_mom: 0 = current, 1 = candidate A, 2 = candidate B
void fill_delta_args(vilma_ctx *c, DeltaArgs &a, double *out, int trial_mu = 0, int trial_mom = 0) {
    a.N = (int32_t)c->N; a.M = c->M; a.A = c->A; a.P = c->P;
    a.pp = nullptr;
    a.mat = 0; a.mu_mat = nullptr; a.cvec = nullptr; a.acoef = 1.0;
    a.mu = c->mu[trial_mu == 2 ? c->mu_tb : trial_mu == 1 ? c->mu_ta : c->mu_cur];
    a.sld = c->sld; a.annot = c->annot;
    a.prec = c->prec; a.log_det = c->log_det; a.lh = c->lh;
    a.lse = c->lse[trial_mom == 2 ? c->mom_tb : trial_mom == 1 ? c->mom_ta : c->mom_cur];
    a.out = out;
    for (int p = 0; p < VILMA_MAX_P; ++p) a.tau.v[p] = p < c->P ? c->tau[p] : 1.0;
}

// Stream for work that only needs the latest per-SNP pass: the side stream once that pass has
// been marked on the caller's stream, else the caller's stream itself.
hipStream_t side_begin(vilma_ctx *c, hipStream_t s) {
    if (!c->overlap || !c->snp_marked || !c->side) return s;
    if (hipStreamWaitEvent(c->side, c->ev_snp, 0) != hipSuccess) return s;
    return c->side;
}
// Everything queued on `s` from here on is ordered after the side work.
void side_end(vilma_ctx *c, hipStream_t s, hipStream_t used) {
    if (used == s) return;
    (void)hipEventRecord(c->ev_side, used);
    (void)hipStreamWaitEvent(s, c->ev_side, 0);
}

// blend: a beta trial at `step`; with totals2_dev also at `step2` in the same pass (candidate B)
// phase >= 0: a phase of a sweep queued ahead (device-resident roles, sweep.hip): the kernels take
// their buffers and step sizes from the context's control block, the host-side roles stay as they are
int evaluate(vilma_ctx *c, hipStream_t s, bool blend, double step, double *totals_dev,
             double *dsum_dev = nullptr, double *dmax_dev = nullptr, double step2 = 0.0,
             double *totals2_dev = nullptr, int phase = -1, double *sums_a_dev = nullptr,
             double *sums_b_dev = nullptr) {
    if (ensure_ready(c)) return 1;
    const bool queued = phase >= 0;
    if (blend && !queued && !c->have_moments)
        return fail(c, "vilma_trial_beta needs an accepted evaluation of the current state");
    const bool two = blend && totals2_dev != nullptr;
    if (two && c->P > 4)
        return fail(c, "two candidates per beta trial are built for up to four cohorts (the kernels of "
                       "five to eight evaluate one)");
    SnpKernelArgs a;
    fill_snp_args(c, a, step);
    a.step2 = step2;
    a.diff = (!blend && dsum_dev && dmax_dev) ? 1 : 0;
    // a beta trial also leaves the per-tile responsibility sums of its candidates (vilma_trial_sums)
    const int ns = two ? 2 : 1;
    a.no_store = (queued && (blend ? c->lazy_trial : c->lazy_persist)) ? (c->lazy_nobase ? 2 : 1) : 0;
    // (lazy trials beyond the stash have none; lazy trials of a mixture that fits keep it: lazy_stash)
    const bool stash = blend && (!a.no_store || c->lazy_stash) && c->sum_partials != nullptr &&
                       snp_pass_can_stash(c->M, c->P, ns);
    if (stash) a.sum_partials = c->sum_partials;
    if (sums_a_dev && !stash) return fail(c, "this trial cannot deliver the responsibility sums");
    if (!queued) c->tile_sums_ns = stash ? ns : 0;
    if (queued) set_launch_phase(&c->ctl->phase[phase]);
    if (c->poison && blend && !a.no_store) {
        const int64_t nmu = mu_buffer_elems(c->N, c->M, c->P);
        launch_poison(a.mu_out, nmu, queued ? 1 : 0, s);
        if (two) launch_poison(a.mu_out2, nmu, queued ? 2 : 0, s);
    }
    {
        // bracketed with the LD product that follows (same sampling tick)
        hipEvent_t e0;
        c->prof_now = c->prof > 0 && (c->prof_tick % c->prof) == 0;
        prof_begin(c, s, e0);
        launch_snp_pass(a, blend, ns, s);
        prof_end(c, s, e0, !blend ? VILMA_PROF_SNP_EVAL
                           : a.no_store ? (two ? VILMA_PROF_SNP_TRIAL2_LAZY : VILMA_PROF_SNP_TRIAL_LAZY)
                           : two ? VILMA_PROF_SNP_TRIAL2 : VILMA_PROF_SNP_TRIAL);
    }
    // (a queued sweep puts nothing on the side stream: its sums are part of the finalize launch,
    // and an event between the pass and the LD product costs microseconds of stream time)
    if (!queued && c->overlap && c->ev_snp) {
        (void)hipEventRecord(c->ev_snp, s);
        c->snp_marked = true;
    }
    run_ld(c, s, c->pool[c->mom_ta], two ? c->pool[c->mom_tb] : nullptr, -1);
    if (queued) set_launch_phase(nullptr);
    // every candidate's totals (and, when asked for, responsibility sums) in one launch
    launch_finalize(c->snp_partials, snp_tile_grid(c->N), c->P, c->dot_partials, c->dot_stride,
                    c->dot_start.data(), ns, totals_dev, totals2_dev, a.diff ? dsum_dev : nullptr,
                    a.diff ? dmax_dev : nullptr, sums_a_dev ? c->sum_partials : nullptr,
                    snp_sum_rows(c->N, c->A), c->A * c->M, sums_a_dev, sums_b_dev, s);
    if (!queued) c->have_b = two;
    HIPCHK(c, hipGetLastError());
    return 0;
}

// the per-tile responsibility sums of the last trial's candidates -> sums_a_dev (, sums_b_dev),
// beside the trial's LD product when a side stream is available
void reduce_tile_sums(vilma_ctx *c, hipStream_t s, double *sums_a_dev, double *sums_b_dev) {
    hipStream_t q = side_begin(c, s);
    const int64_t rows_elems = (int64_t)snp_sum_rows(c->N, c->A) * c->A * c->M;
    if (sums_b_dev && sums_b_dev > sums_a_dev) {
        // both candidates in one launch
        launch_tile_sums(c->sum_partials, c->N, c->A, c->M, 2, sums_a_dev, sums_b_dev - sums_a_dev, q);
    } else {
        launch_tile_sums(c->sum_partials, c->N, c->A, c->M, 1, sums_a_dev, 0, q);
        if (sums_b_dev)
            launch_tile_sums(c->sum_partials + rows_elems, c->N, c->A, c->M, 1, sums_b_dev, 0, q);
    }
    side_end(c, s, q);
}

}  // namespace

// the pieces of a sweep queued ahead (sweep.hip); kernels launched by these honour the launch
// predicate of the calling thread
int vilma_detail::queue_trial_phase(vilma_ctx *c, hipStream_t s, bool two, double *totals_a,
                                    double *totals_b, double *sums_a, double *sums_b) {
    return evaluate(c, s, true, 0.0, totals_a, nullptr, nullptr, 0.0, two ? totals_b : nullptr,
                    VILMA_PHASE_TRIAL, sums_a, two ? sums_b : nullptr);
}
int vilma_detail::queue_eval_phase(vilma_ctx *c, hipStream_t s, double *totals, double *dsum,
                                   double *dmax) {
    return evaluate(c, s, false, 0.0, totals, dsum, dmax, 0.0, nullptr, VILMA_PHASE_EVAL);
}
// writes_state: the pass is queued to WRITE the state out (a tau update under a persistent lazy state,
// sweep.hip), whatever it does with the sums
int vilma_detail::queue_sums_phase(vilma_ctx *c, hipStream_t s, double *sums_dev, bool writes_state) {
    DeltaArgs a;
    fill_delta_args(c, a, c->delta_partials);
    a.mat = c->lazy_trial ? 1 : 0;       // behind a lazy trial the pass also stores the candidate
    set_launch_phase(&c->ctl->phase[VILMA_PHASE_SUMS]);
    {
        // bracketed on the tick of the LD product that follows, like the per-SNP pass
        hipEvent_t e0;
        c->prof_now = c->prof > 0 && (c->prof_tick % c->prof) == 0;
        prof_begin(c, s, e0);
        launch_delta_sums(a, sums_dev, s);
        // (a persistent lazy state: the pass derives the state and stores nothing)
        prof_end(c, s, e0, (a.mat && (!c->lazy_persist || writes_state)) ? VILMA_PROF_SUMS_MAT : VILMA_PROF_SUMS);
    }
    set_launch_phase(nullptr);
    HIPCHK(c, hipGetLastError());
    return 0;
}
// The state lazy trials reached, a * mu[mu_from] + Sig cvec[mom] (PhasePtrs), written out into
// mu[mu_to] by the host's own launch (a device-resident sweep handed back in the middle of a beta
// loop: the host's line search works on stored vi_mu).  Its responsibility sums go to sums_dev.
int vilma_detail::materialise_deferred(vilma_ctx *c, hipStream_t s, int mu_from, int mu_to, int c_buf,
                                       int lse_buf, double a_def, const double *tau, double *sums_dev) {
    DeltaArgs a;
    fill_delta_args(c, a, c->delta_partials);
    a.mat = 1;
    a.mu = c->mu[mu_from];
    a.mu_mat = c->mu[mu_to];
    a.cvec = c->cvec[c_buf];
    a.acoef = a_def;
    a.lse = c->lse[lse_buf];
    for (int p = 0; p < VILMA_MAX_P; ++p) a.tau.v[p] = p < c->P ? tau[p] : 1.0;
    launch_delta_sums(a, sums_dev, s);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int vilma_detail::queue_mstep(vilma_ctx *c, hipStream_t s, const double *sums_dev, double *hyper_dev) {
    launch_mstep(sums_dev, c->counts, c->log_det, c->A, c->M, hyper_dev, c->lh, s);
    HIPCHK(c, hipGetLastError());
    return 0;
}
static void prof_drop_if(vilma_ctx *c, int64_t lo, int64_t hi) {
    size_t keep = 0;
    for (size_t i = 0; i < c->pending.size(); ++i) {
        const auto &pr = c->pending[i];
        if (pr.tag >= lo && pr.tag <= hi && pr.tag != 0) {
            c->event_pool.push_back(pr.e0);
            c->event_pool.push_back(pr.e1);
        } else {
            c->pending[keep++] = pr;
        }
    }
    c->pending.resize(keep);
}
void vilma_detail::prof_drop_tag(vilma_ctx *c, int64_t tag) { prof_drop_if(c, tag, tag); }
void vilma_detail::prof_drop_tags(vilma_ctx *c, int64_t from_tag) { prof_drop_if(c, from_tag, INT64_MAX); }

// A bare read of the LD store: what this placement of the store in HBM streams at, with nothing
// else in the way (16 B per lane, non-temporal, 8 loads in flight per thread, 32 KB per workgroup
// step).  The yardstick vilma_prof_stream_store reports beside the LD kernels' own times.
typedef double probe_v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void store_stream_kernel(const probe_v2d *__restrict__ p,
                                                           int64_t n_chunks, double *sink) {
    double acc = 0.0;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const probe_v2d *q = p + ch * 2048 + threadIdx.x;
        probe_v2d t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(q + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u].x + t[u].y;
    }
    if (acc == 1.2345e300) sink[0] = acc;      // never true for LD data: keeps the loads alive
}

// The same bare read with the store cut into chunks of `steps` x 32 KB, each read front to back by
// one workgroup; chunk c of the grid-stride loop is chunk (c * mult) mod n_chunks of the store
// (mult = 1: store order; an odd multiplier coprime to n_chunks: scattered).  What an access
// pattern alone costs: the LD kernels' workgroups each stream their own panel chunk (up to 512 KB)
// from wherever it lies, some two thousand of them at a time.
// writes != 0: a thin stream of stores beside the read stream, 1/128 of the bytes read unless
// stated (ld_sym_kernel's partial sums are that much), into `wbuf`:
//   1  every wave stores 8 doubles (64 B, lanes 0, 8, ..., 56: the shape of ld_sym_kernel's row-sum
//      partials) per 8 KiB it reads; 2 = the same, non-temporal
//   3  every wave stores one whole 128-B line (16 lanes) per 16 KiB it reads
//   4  as 1 but every other step only: half the bytes
//   5  the workgroup stores 4 KiB contiguous (256 threads x 16 B) per 512 KB it reads
//   6  nothing while it streams; at the end of its loop the workgroup stores everything it would
//      have stored, contiguously
//   7  the workgroup stores 4 KiB per 512 KB it reads (as 5), at an address that follows the ORDER
//      IN TIME of the chunks (chunk c of the grid-stride loop -> record c) instead of where the
//      chunk lies in the store: workgroups that run side by side write side by side
//   8  as 1 (64 B per wave per 8 KiB), addressed as 7
//   9  as 1 with write-through stores (sc1: the line does not stay dirty in L2)
//  10  as 5 (4 KiB per workgroup per 512 KB) with write-through stores
__global__ __launch_bounds__(256) void store_pattern_kernel(const probe_v2d *__restrict__ p,
                                                            int64_t n_chunks, int steps,
                                                            int64_t mult, double *sink, int writes,
                                                            double *__restrict__ wbuf,
                                                            int64_t wbuf_per_wg) {
    double acc = 0.0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t done = 0;                   // 32-KB steps this workgroup has read
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int64_t src = mult == 1 ? ch : (int64_t)(((unsigned __int128)ch * (uint64_t)mult) % (uint64_t)n_chunks);
        const probe_v2d *q = p + src * steps * 2048 + threadIdx.x;
        for (int st = 0; st < steps; ++st, q += 2048, ++done) {
            probe_v2d t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(q + u * 256);
            double part = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) part += t[u].x + t[u].y;
            acc += part;
            const int64_t gstep = src * steps + st;         // this 32-KB step of the store
            if ((writes == 1 || writes == 2 || writes == 9 || (writes == 4 && (st & 1) == 0)) && (lane & 7) == 0) {
                double *dst = wbuf + (gstep * 4 + w) * 8 + (lane >> 3);     // [step][wave][8]
                if (writes == 2) __builtin_nontemporal_store(part, dst);
                else if (writes == 9) __hip_atomic_store(dst, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = part;
            } else if (writes == 3 && (st & 1) == 0 && lane < 16) {
                wbuf[(gstep * 4 + w) * 8 + lane] = part;                    // 16 doubles = a line
            } else if (writes == 5 && (done & 15) == 15) {
                probe_v2d *dst = (probe_v2d *)(wbuf + (gstep - 15) * 32) + threadIdx.x;     // 4 KiB
                *dst = probe_v2d{part, acc};
            } else if (writes == 10 && (done & 15) == 15) {
                double *dst = wbuf + (gstep - 15) * 32 + 2 * threadIdx.x;
                __hip_atomic_store(dst, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (writes == 7 && (done & 15) == 15) {
                probe_v2d *dst = (probe_v2d *)(wbuf + ((ch * steps + st) - 15) * 32) + threadIdx.x;
                *dst = probe_v2d{part, acc};
            } else if (writes == 8 && (lane & 7) == 0) {
                wbuf[((ch * steps + st) * 4 + w) * 8 + (lane >> 3)] = part;
            }
        }
    }
    if (writes == 6) {
        double *dst = wbuf + (int64_t)blockIdx.x * wbuf_per_wg;
        const int64_t n = done * 32 < wbuf_per_wg ? done * 32 : wbuf_per_wg;
        for (int64_t i = 2 * threadIdx.x; i + 1 < n; i += 512) *(probe_v2d *)(dst + i) = probe_v2d{acc, acc};
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

extern "C" {

const char *vilma_version(void) { return "vilma_hip 0.1 (gfx950)"; }

const char *vilma_last_error(const vilma_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int vilma_create(int P, int64_t N, int M, int A, vilma_ctx **out) {
    if (!out) return fail(nullptr, "out is NULL");
    *out = nullptr;
    if (P < 1 || P > VILMA_MAX_P) return fail(nullptr, "P must be in 1..8 (cohorts)");
    if (N < 1 || N >= ((int64_t)1 << 30)) return fail(nullptr, "N out of range");
    if (M < 2) return fail(nullptr, "M must be >= 2 (mixture components)");
    if (A < 1) return fail(nullptr, "A must be >= 1 (annotations)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, "no HIP device available");
    vilma_ctx *c = new vilma_ctx();
    c->P = P; c->N = N; c->M = M; c->A = A;
    (void)hipGetDevice(&c->device);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0)
            c->n_cu = cus;
    }
    c->ld.resize(P);
    for (int p = 0; p < VILMA_MAX_P; ++p) c->tau[p] = 1.0;
    const int64_t PN = (int64_t)P * N;
    int rc = 0;
    rc |= dev_alloc(c, &c->adj, PN); rc |= dev_alloc(c, &c->se, PN);
    rc |= dev_alloc(c, &c->sld, PN); rc |= dev_alloc(c, &c->scal, PN);
    rc |= dev_alloc(c, &c->annot, N); rc |= dev_alloc(c, &c->invperm, PN);
    rc |= dev_alloc(c, &c->prec, (int64_t)M * P * P); rc |= dev_alloc(c, &c->log_det, M);
    rc |= dev_alloc(c, &c->lh, (int64_t)A * M);
    rc |= dev_alloc(c, &c->counts, A);
    for (int s = 0; s < 3 && !rc; ++s) {
        rc |= dev_alloc(c, &c->mu[s], mu_buffer_elems(N, M, P));
        rc |= dev_alloc(c, &c->m[s], PN); rc |= dev_alloc(c, &c->v[s], PN);
        rc |= dev_alloc(c, &c->lse[s], N);
    }
    for (int s = 0; s < 3 && !rc; ++s) rc |= dev_alloc(c, &c->cvec[s], PN);
    rc |= dev_alloc(c, &c->snap[0], PN);
    rc |= dev_alloc(c, &c->snap[1], PN);
    rc |= dev_alloc(c, &c->snp_partials, (int64_t)snp_tile_grid(N) * (2 * (2 * P + 2) + 6));
    if (snp_pass_can_stash(M, P, 1)) {
        const char *st = std::getenv("VILMA_TILE_SUMS");        // =0: always delta_kernel (A/B)
        if (!(st && st[0] == '0'))
            rc |= dev_alloc(c, &c->sum_partials, tile_sums_elems(N, A, M, 2));
    }
    // per-wave rows plus the scratch rows of every pass of the column reduction (exact)
    rc |= dev_alloc(c, &c->delta_partials,
                    std::max(delta_partial_rows(N), init_partial_rows(N)) * A * M);
    rc |= dev_alloc(c, &c->diff_partials,
                    std::max((int64_t)mean_diff_grid(PN) * 6, (int64_t)snp_pass_grid(N)));
    if (rc) {
        g_create_error = c->err;
        vilma_destroy(c);
        return 1;
    }
    c->log_det_host.assign(M, 0.0);
    if (const char *gm = std::getenv("VILMA_EIGEN_GROUP_MB")) {
        const int v = std::atoi(gm);
        if (v >= 1) c->eigen_group_bytes = (int64_t)v << 20;
    }
    if (const char *ef = std::getenv("VILMA_EIG_FUSED")) c->eig_fused = ef[0] != '0';
    if (const char *mb = std::getenv("VILMA_EIG_MERGE_BELOW")) c->eig_merge_below = std::atoi(mb);
    if (const char *ew = std::getenv("VILMA_EIG_WAVE")) c->eig_wave = ew[0] != '0';
    if (const char *se = std::getenv("VILMA_EIG_SLAB_ELEMS")) {
        const int v = std::atoi(se);
        if (v >= 1024) g_eig_slab_elems = v;
    }
    if (const char *lo = std::getenv("VILMA_LD_ORDER")) {
        const int v = std::atoi(lo);
        if (v >= 0 && v <= 2) c->ld_order = v;
    }
    if (const char *cr = std::getenv("VILMA_LD_CHUNK_ROWS")) {
        const int v = std::atoi(cr);
        if (v >= 128) c->chunk_rows = std::min((v + 31) / 32 * 32, 512);    // LD_MAX_CHUNK_ROWS (kernels.hip)
    }
    if (const char *lt = std::getenv("VILMA_LD_TILE")) {
        int tr = 0, ts = 1;
        if (std::sscanf(lt, "%d,%d", &tr, &ts) >= 1) {
            ts = std::max(1, std::min(ts, 4));                  // LD_TILE_MAX_SLABS
            tr = std::min(tr, 512) / (128 * ts) * (128 * ts);   // LD_TILE_MAX_ROWS; whole column strips
            c->tile_rows = tr;
            c->tile_slabs = ts;
            c->tile_auto = false;
        }
    }
    // VILMA_OVERLAP=0 keeps everything on the caller's stream (A/B measurements)
    if (const char *po = std::getenv("VILMA_DEBUG_POISON")) c->poison = po[0] != '0' && po[0] != 0;
    const char *ov = std::getenv("VILMA_OVERLAP");
    c->overlap = !(ov && ov[0] == '0');
    // The side stream is created at HIGH priority: HIP maps streams onto a small pool of hardware
    // queues per priority level, and two streams that land on the same queue serialise.  With
    // RCCL / torch streams alive in the process that is what happened to a normal-priority side
    // stream (its kernels ran after the LD product instead of beside it: +40 us per trial on an
    // 8-GPU shard).  A different priority means a different queue pool, and the short
    // responsibility kernels get their waves ahead of the long LD stream.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (c->overlap &&
        (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
         hipEventCreateWithFlags(&c->ev_snp, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess)) {
        g_create_error = "cannot create the side stream";
        vilma_destroy(c);
        return 1;
    }
    *out = c;
    return 0;
}

void vilma_destroy(vilma_ctx *c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    sweep_destroy(c);
    prof_resolve(c);
    free_ready(c);
    for (auto &co : c->ld) dev_free(co.store);
    dev_free(c->repack_tmp);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->ev_snp) (void)hipEventDestroy(c->ev_snp);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->side) (void)hipStreamDestroy(c->side);
    void *ptrs[] = {c->adj, c->se, c->sld, c->scal, c->annot, c->invperm, c->prec, c->log_det,
                    c->lh, c->counts, c->sum_partials, c->mu[0], c->mu[1], c->mu[2], c->m[0], c->m[1], c->m[2], c->v[0],
                    c->v[1], c->v[2], c->lse[0], c->lse[1], c->lse[2], c->snap[0], c->snap[1], c->cvec[0], c->cvec[1], c->cvec[2], c->snp_partials, c->delta_partials, c->diff_partials};
    for (void *p : ptrs) dev_free(p);
    delete c;
}

int vilma_set_snp_data(vilma_ctx *c, const double *adj, const double *se, const double *sld,
                       const double *scalings, const int32_t *annot) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    const size_t b = (size_t)c->P * c->N * sizeof(double);
    HIPCHK(c, hipMemcpy(c->adj, adj, b, hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(c->se, se, b, hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(c->sld, sld, b, hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(c->scal, scalings, b, hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(c->annot, annot, (size_t)c->N * sizeof(int32_t), hipMemcpyDefault));
    c->have_moments = false;
    return 0;
}

int vilma_set_mixture(vilma_ctx *c, const double *prec, const double *log_det) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    HIPCHK(c, hipMemcpy(c->prec, prec, (size_t)c->M * c->P * c->P * sizeof(double), hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(c->log_det, log_det, (size_t)c->M * sizeof(double), hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(c->log_det_host.data(), log_det, (size_t)c->M * sizeof(double), hipMemcpyDefault));
    c->have_moments = false;
    return 0;
}

int vilma_set_tau(vilma_ctx *c, const double *tau) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    for (int p = 0; p < c->P; ++p) {
        if (!(tau[p] > 0.0) || !std::isfinite(tau[p])) return fail(c, "error_scaling must be positive");
        c->tau[p] = tau[p];
    }
    return 0;
}

int vilma_set_hyper(vilma_ctx *c, const double *hyper) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    std::vector<double> lh((size_t)c->A * c->M);
    for (int a = 0; a < c->A; ++a)
        for (int k = 0; k < c->M; ++k) {
            const double h = hyper[(size_t)a * c->M + k];
            if (!(h > 0.0)) return fail(c, "hyper_delta entries must be positive");
            // det_lh: the same bits the device's M-step (mstep_row) would have produced
            lh[(size_t)a * c->M + k] = det_lh(h, c->log_det_host[k]);
        }
    // ordered behind kernels already queued on any stream of this device
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(c->lh, lh.data(), lh.size() * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int vilma_set_annotation_counts(vilma_ctx *c, const double *counts) {
    if (!c) return 1;
    HIPCHK(c, hipMemcpy(c->counts, counts, (size_t)c->A * sizeof(double), hipMemcpyDefault));
    return 0;
}

int vilma_mstep(vilma_ctx *c, void *stream, const double *sums_dev, double *hyper_dev) {
    if (!c) return 1;
    launch_mstep(sums_dev, c->counts, c->log_det, c->A, c->M, hyper_dev, c->lh, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int64_t vilma_ld_dense_elems(int n) {
    int64_t e = 0;
    for (int J = 0; J < n_slabs(n); ++J) e += sym_panel_elems(n, J);
    return e;
}
int64_t vilma_ld_lowrank_elems(int n, int r) {
    // U then s [ld(r)]: U is [n x ld(r)] row-major or [r][pad2(n)] column-major (fused product)
    return (int64_t)pad2(n) * pad_ld(r) + pad_ld(r);
}

int vilma_ld_begin(vilma_ctx *c, int cohort, int n_blocks, int64_t n_ld, const int64_t *perm,
                   int64_t total_elems) {
    if (!c) return 1;
    if (cohort < 0 || cohort >= c->P) return fail(c, "cohort out of range");
    if (n_ld < 0 || n_ld > c->N || n_blocks < 0 || total_elems < 0) return fail(c, "bad LD sizes");
    HIPCHK(c, hipDeviceSynchronize());
    free_ready(c);
    CohortLd &co = c->ld[cohort];
    dev_free(co.store);
    co = CohortLd();
    co.begun = true; co.n_blocks = n_blocks; co.n_ld = n_ld; co.store_elems = total_elems;
    if (dev_alloc(c, &co.store, total_elems + 2)) return 1;
    // perm must be a permutation of 0..N-1; invperm[snp] = LD position
    std::vector<int32_t> inv((size_t)c->N, -1);
    for (int64_t t = 0; t < c->N; ++t) {
        const int64_t sidx = perm[t];
        if (sidx < 0 || sidx >= c->N || inv[(size_t)sidx] != -1)
            return fail(c, "perm and missing should together contain all of the indices");
        inv[(size_t)sidx] = (int32_t)t;
    }
    HIPCHK(c, hipMemcpy(c->invperm + (size_t)cohort * c->N, inv.data(),
                        (size_t)c->N * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

int vilma_ld_add_dense(vilma_ctx *c, int cohort, int n, const double *R) {
    if (!c) return 1;
    if (cohort < 0 || cohort >= c->P) return fail(c, "cohort out of range");
    CohortLd &co = c->ld[cohort];
    if (!co.begun || co.ended) return fail(c, "vilma_ld_begin not called for this cohort");
    if (n < 1) return fail(c, "block size must be positive");
    if ((int)co.blocks.size() >= co.n_blocks) return fail(c, "more blocks than announced");
    if (co.next_start + (int64_t)n > co.n_ld) return fail(c, "blocks exceed n_ld");
    const int64_t need = vilma_ld_dense_elems(n);
    if (co.store_used + need > co.store_elems) return fail(c, "LD store overflow (total_elems)");
    // only the lower triangle is kept, slab by slab: rows 128J.. of columns [128J, 128J+w)
    double *dst = co.store + co.store_used;
    for (int J = 0; J < n_slabs(n); ++J) {
        const int w = slab_width(n, J), j0 = 128 * J;
        HIPCHK(c, hipMemcpy2D(dst, (size_t)pad_ld(w) * sizeof(double), R + (size_t)j0 * n + j0,
                              (size_t)n * sizeof(double), (size_t)w * sizeof(double),
                              (size_t)(n - j0), hipMemcpyDefault));
        dst += sym_panel_elems(n, J);
    }
    // the copies above run on the NULL stream and a device-to-device hipMemcpy may return before it
    // has finished: the caller is free to release (or a caching allocator to reuse) R the moment
    // this returns, so wait here
    HIPCHK(c, hipStreamSynchronize(nullptr));
    BlockRec b{0, n, n, co.store_used, 0, co.next_start, 0};
    co.blocks.push_back(b);
    co.store_used += need;
    co.next_start += n;
    co.alg_bytes += (int64_t)8 * n * n;
    return 0;
}

int vilma_ld_add_lowrank(vilma_ctx *c, int cohort, int n, int r, const double *U, const double *s) {
    if (!c) return 1;
    if (cohort < 0 || cohort >= c->P) return fail(c, "cohort out of range");
    CohortLd &co = c->ld[cohort];
    if (!co.begun || co.ended) return fail(c, "vilma_ld_begin not called for this cohort");
    if (n < 1 || r < 1) return fail(c, "block size and rank must be positive");
    if ((int)co.blocks.size() >= co.n_blocks) return fail(c, "more blocks than announced");
    if (co.next_start + (int64_t)n > co.n_ld) return fail(c, "blocks exceed n_ld");
    const int64_t need = vilma_ld_lowrank_elems(n, r);
    if (co.store_used + need > co.store_elems) return fail(c, "LD store overflow (total_elems)");
    // only U and s are stored.  Fused product: U column-major [r][pad2(n)] (pad row zero); blocks
    // too tall for it keep the row-major U [n x ld(r)] of the two-pass kernels
    const int W = c->eig_fused ? eig_rows_per_thread(n) : 0;
    double *dU = co.store + co.store_used;
    double *dS = dU + (int64_t)pad2(n) * pad_ld(r);
    if (W > 0) {
        const int64_t need_tmp = (int64_t)n * r;
        if (need_tmp > c->repack_elems) {
            dev_free(c->repack_tmp);
            c->repack_tmp = nullptr;
            c->repack_elems = 0;
            if (dev_alloc(c, &c->repack_tmp, need_tmp + need_tmp / 2)) return 1;
            c->repack_elems = need_tmp + need_tmp / 2;
        }
        HIPCHK(c, hipMemcpy(c->repack_tmp, U, (size_t)need_tmp * sizeof(double), hipMemcpyDefault));
        launch_repack_columns(c->repack_tmp, n, r, pad2(n), dU, nullptr);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemcpy2D(dU, (size_t)pad_ld(r) * sizeof(double), U, (size_t)r * sizeof(double),
                              (size_t)r * sizeof(double), (size_t)n, hipMemcpyDefault));
    }
    HIPCHK(c, hipMemcpy(dS, s, (size_t)r * sizeof(double), hipMemcpyDefault));
    HIPCHK(c, hipStreamSynchronize(nullptr));      // as in vilma_ld_add_dense: U and s may be released now
    BlockRec b{1, n, r, co.store_used, co.store_used + (int64_t)pad2(n) * pad_ld(r), co.next_start,
               (int32_t)co.t_used, W};
    co.blocks.push_back(b);
    co.store_used += need;
    co.next_start += n;
    co.t_used += pad_ld(r);
    // partial row sums S[slab][n]
    {
        // partial row sums S[slab][n]; a fused block of one slab needs none (make_items)
        const int cols = W > 0 ? eig_slab_cols(n, r, W) : 128;
        const int ns = (r + cols - 1) / cols;
        co.s_eig_used += (W > 0 && ns == 1) ? 0 : (int64_t)ns * pad2(n);
    }
    co.alg_bytes += (int64_t)8 * n * r;
    return 0;
}

int vilma_ld_end(vilma_ctx *c, int cohort) {
    if (!c) return 1;
    if (cohort < 0 || cohort >= c->P) return fail(c, "cohort out of range");
    CohortLd &co = c->ld[cohort];
    if (!co.begun) return fail(c, "vilma_ld_begin not called for this cohort");
    if ((int)co.blocks.size() != co.n_blocks) return fail(c, "fewer blocks than announced");
    if (co.next_start != co.n_ld) return fail(c, "blocks do not cover n_ld");
    co.ended = true;
    return 0;
}

int vilma_ld_bytes(const vilma_ctx *c, int64_t *alg, int64_t *stored) {
    if (!c) return 1;
    int64_t a = 0, s = 0;
    for (const auto &co : c->ld) { a += co.alg_bytes; s += co.store_used * 8; }
    if (alg) *alg = a;
    if (stored) *stored = s;
    return 0;
}

int vilma_ld_matvec(vilma_ctx *c, void *stream, int cohort, const double *x, double *y) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (cohort >= c->P) return fail(c, "cohort out of range");
    if (ensure_ready(c)) return 1;
    hipStream_t s = (hipStream_t)stream;
    double *pl = c->pool[c->mom_ta];          // the trial pool doubles as workspace
    launch_gather_x(x, c->invperm, pl, (int)c->N, c->P, s);
    run_ld(c, s, pl, nullptr, cohort);
    launch_scatter_y(pl + (int64_t)c->P * c->N, c->invperm, y, (int)c->N, c->P, s);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int vilma_ld_matvec2(vilma_ctx *c, void *stream, int cohort, const double *x0, const double *x1,
                     double *y0, double *y1) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;
    if (cohort >= c->P) return fail(c, "cohort out of range");
    if (ensure_ready(c)) return 1;
    hipStream_t s = (hipStream_t)stream;
    double *pa = c->pool[c->mom_ta], *pb = c->pool[c->mom_tb];   // the trial pools double as workspace
    launch_gather_x(x0, c->invperm, pa, (int)c->N, c->P, s);
    launch_gather_x(x1, c->invperm, pb, (int)c->N, c->P, s);
    run_ld(c, s, pa, pb, cohort);
    launch_scatter_y(pa + (int64_t)c->P * c->N, c->invperm, y0, (int)c->N, c->P, s);
    launch_scatter_y(pb + (int64_t)c->P * c->N, c->invperm, y1, (int)c->N, c->P, s);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// The vi_mu buffers may be laid out tile by tile (kernels.hip, MU_TILED); the caller's array is the
// reference's [M][P][N].  Rows of it travel through a staging chunk on the device (a few rows of N
// at a time: no buffer of the context is used as scratch, so a pending trial survives).
static int mu_convert(vilma_ctx *c, double *buf, double *host, bool to_device) {
    const int MP = c->M * c->P;
    const int R = (int)std::max<int64_t>(1, std::min<int64_t>(MP, ((int64_t)8 << 20) / std::max<int64_t>(c->N, 1)));
    double *stage = nullptr;
    HIPCHK(c, hipMalloc((void **)&stage, (size_t)R * c->N * sizeof(double)));
    int rc = 0;
    for (int r0 = 0; r0 < MP && !rc; r0 += R) {
        const int rows = std::min(R, MP - r0);
        const size_t bytes = (size_t)rows * c->N * sizeof(double);
        if (to_device) {
            if (hipMemcpy(stage, host + (size_t)r0 * c->N, bytes, hipMemcpyDefault) != hipSuccess) { rc = 1; break; }
            launch_mu_tile(buf, stage, c->N, MP, r0, rows, true, nullptr);
            if (hipDeviceSynchronize() != hipSuccess) rc = 1;
        } else {
            launch_mu_tile(buf, stage, c->N, MP, r0, rows, false, nullptr);
            if (hipDeviceSynchronize() != hipSuccess) { rc = 1; break; }
            if (hipMemcpy(host + (size_t)r0 * c->N, stage, bytes, hipMemcpyDefault) != hipSuccess) rc = 1;
        }
    }
    (void)hipFree(stage);
    if (rc) return fail(c, "vi_mu layout conversion failed");
    return 0;
}

int vilma_set_mu(vilma_ctx *c, const double *vi_mu) {
    if (c) c->pure_c = -1;
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    HIPCHK(c, hipDeviceSynchronize());
    if (mu_is_tiled()) {
        if (mu_convert(c, c->mu[c->mu_cur], const_cast<double *>(vi_mu), true)) return 1;
    } else {
        HIPCHK(c, hipMemcpy(c->mu[c->mu_cur], vi_mu, (size_t)c->M * c->P * c->N * sizeof(double),
                            hipMemcpyDefault));
    }
    c->have_moments = false;
    return 0;
}

int vilma_get_mu(vilma_ctx *c, double *vi_mu) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    HIPCHK(c, hipDeviceSynchronize());
    if (mu_is_tiled()) {
        if (mu_convert(c, c->mu[c->mu_cur], vi_mu, false)) return 1;
    } else {
        HIPCHK(c, hipMemcpy(vi_mu, c->mu[c->mu_cur], (size_t)c->M * c->P * c->N * sizeof(double),
                            hipMemcpyDefault));
    }
    return 0;
}

int vilma_get_delta(vilma_ctx *c, double *vi_delta) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!c->have_moments) return fail(c, "no accepted evaluation of the current state");
    HIPCHK(c, hipDeviceSynchronize());
    // the trial vi_mu buffer ([M][P][N] >= [M][N]) is free between evaluations: use it as scratch
    double *scratch = c->mu[c->mu_ta];
    DeltaArgs a;
    fill_delta_args(c, a, scratch);
    launch_delta_write(a, nullptr);
    HIPCHK(c, hipDeviceSynchronize());
    // transposed on the device, a slice of SNPs at a time through a staging buffer (the host loop
    // over [M][N] took 2 s at 1 M SNPs x 582 components)
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(c->N, ((int64_t)32 << 20) / std::max(c->M, 1)));
    double *stage = nullptr;
    HIPCHK(c, hipMalloc((void **)&stage, (size_t)chunk * c->M * sizeof(double)));
    int rc = 0;
    for (int64_t i0 = 0; i0 < c->N && !rc; i0 += chunk) {
        const int n = (int)std::min<int64_t>(chunk, c->N - i0);
        launch_transpose_km(scratch, stage, c->N, c->M, i0, n, nullptr);
        if (hipMemcpy(vi_delta + (size_t)i0 * c->M, stage, (size_t)n * c->M * sizeof(double),
                      hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    }
    (void)hipFree(stage);
    if (rc) return fail(c, "vilma_get_delta: device to host copy failed");
    return 0;
}

int vilma_get_vi_sigma(vilma_ctx *c, const double *error_scaling, double *vi_sigma) {
    if (!c || !vi_sigma) return 1;
    if (error_scaling)
        for (int p = 0; p < c->P; ++p)
            if (!(error_scaling[p] > 0.0) || !std::isfinite(error_scaling[p]))
                return fail(c, "error_scaling must be positive");
    if (vilma_sweep_drain(c)) return 1;      // the staging buffer is a trial buffer: nothing may be queued
    HIPCHK(c, hipDeviceSynchronize());
    double *stage = c->mu[c->mu_ta];         // M P N doubles: M / P components of [P][P][N] at a time
    const int nk_max = std::max(1, c->M / c->P);
    TauArg tau;
    for (int p = 0; p < VILMA_MAX_P; ++p) tau.v[p] = p < c->P ? (error_scaling ? error_scaling[p] : c->tau[p]) : 1.0;
    const size_t per_k = (size_t)c->P * c->P * c->N;
    for (int k0 = 0; k0 < c->M; k0 += nk_max) {
        const int nk = std::min(nk_max, c->M - k0);
        launch_vi_sigma(c->P, c->prec, c->sld, tau, c->N, k0, nk, stage, nullptr);
        HIPCHK(c, hipMemcpy(vi_sigma + (size_t)k0 * per_k, stage, (size_t)nk * per_k * sizeof(double),
                            hipMemcpyDeviceToHost));
    }
    return 0;
}

int vilma_get_moments(vilma_ctx *c, double *mean, double *var) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!c->have_moments) return fail(c, "no accepted evaluation of the current state");
    HIPCHK(c, hipDeviceSynchronize());
    const size_t b = (size_t)c->P * c->N * sizeof(double);
    if (mean) HIPCHK(c, hipMemcpy(mean, c->m[c->mom_cur], b, hipMemcpyDefault));
    if (var) HIPCHK(c, hipMemcpy(var, c->v[c->mom_cur], b, hipMemcpyDefault));
    return 0;
}

int vilma_eval(vilma_ctx *c, void *stream, double *totals_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    c->trial_tainted = false;
    return evaluate(c, (hipStream_t)stream, false, 0.0, totals_dev);
}

int vilma_eval_diff(vilma_ctx *c, void *stream, double *totals_dev, double *out_sum3_dev,
                    double *out_max3_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!out_sum3_dev || !out_max3_dev) return fail(c, "vilma_eval_diff needs both outputs");
    c->trial_tainted = false;
    return evaluate(c, (hipStream_t)stream, false, 0.0, totals_dev, out_sum3_dev, out_max3_dev);
}

int vilma_eval_given_delta(vilma_ctx *c, void *stream, const double *delta_km_dev,
                           double *totals_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (ensure_ready(c)) return 1;
    if (!c->have_moments) return fail(c, "no accepted evaluation of the current state");
    hipStream_t s = (hipStream_t)stream;
    SnpKernelArgs a;
    fill_snp_args(c, a, 0.0);
    c->snp_marked = false;          // nothing of this evaluation may overlap on the side stream
    launch_snp_given_delta(a, delta_km_dev, c->lse[c->mom_cur], c->diff_partials,
                           totals_dev + VILMA_NTOTALS(c->P), s);
    run_ld(c, s, c->pool[c->mom_ta], nullptr, -1);
    launch_finalize(c->snp_partials, snp_pass_grid(c->N), c->P, c->dot_partials, c->dot_stride,
                    c->dot_start.data(), 1, totals_dev, nullptr, nullptr, nullptr, nullptr, 0, 0,
                    nullptr, nullptr, s);
    HIPCHK(c, hipGetLastError());
    c->trial_tainted = true;        // these moments belong to no state the line search may accept
    return 0;
}

int vilma_get_trial_moments(vilma_ctx *c, double *mean, double *var) {
    if (!c) return 1;
    if (!c->ready) return fail(c, "no trial state");
    HIPCHK(c, hipDeviceSynchronize());
    const size_t b = (size_t)c->P * c->N * sizeof(double);
    if (mean) HIPCHK(c, hipMemcpy(mean, c->m[c->mom_ta], b, hipMemcpyDefault));
    if (var) HIPCHK(c, hipMemcpy(var, c->v[c->mom_ta], b, hipMemcpyDefault));
    return 0;
}

int vilma_init_state(vilma_ctx *c, void *stream, const double *fake_mu, double *sums_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipDeviceSynchronize());
    // the snapshot buffer ([P][N]) is free until the sweep loop starts: staging for fake_mu
    double *stage = c->snap[c->snap_cur];
    HIPCHK(c, hipMemcpy(stage, fake_mu, (size_t)c->P * c->N * sizeof(double), hipMemcpyDefault));
    InitArgs a;
    a.N = (int32_t)c->N; a.M = c->M; a.A = c->A; a.P = c->P;
    a.fake_mu = stage; a.sld = c->sld; a.annot = c->annot;
    a.prec = c->prec; a.log_det = c->log_det;
    a.mu_out = c->mu[c->mu_cur];
    a.c_out = c->cvec[c->mu_cur];
    a.partials = c->delta_partials;
    for (int p = 0; p < VILMA_MAX_P; ++p) a.tau.v[p] = p < c->P ? c->tau[p] : 1.0;
    launch_init_state(a, sums_dev, s);
    HIPCHK(c, hipGetLastError());
    c->have_moments = false;
    c->pure_c = c->mu_cur;
    for (int p = 0; p < VILMA_MAX_P; ++p) c->pure_tau[p] = a.tau.v[p];
    return 0;
}

int vilma_trial_beta(vilma_ctx *c, void *stream, double step, double *totals_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    c->trial_tainted = false;
    return evaluate(c, (hipStream_t)stream, true, step, totals_dev);
}

int vilma_trial_beta2(vilma_ctx *c, void *stream, double step_a, double step_b,
                      double *totals_a_dev, double *totals_b_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!totals_a_dev || !totals_b_dev) return fail(c, "vilma_trial_beta2 needs both outputs");
    c->trial_tainted = false;
    return evaluate(c, (hipStream_t)stream, true, step_a, totals_a_dev, nullptr, nullptr, step_b,
                    totals_b_dev);
}

int vilma_accept(vilma_ctx *c, int take_mu) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!c->ready) return fail(c, "nothing to accept");
    if (take_mu < 0 || take_mu > 2) return fail(c, "vilma_accept: take_mu must be 0, 1 or 2");
    if (c->trial_tainted)
        return fail(c, "the trial state was evaluated with a caller-supplied vi_delta and cannot "
                       "be accepted");
    if (take_mu == 2) {
        if (!c->have_b) return fail(c, "no second candidate: the last trial was not vilma_trial_beta2");
        std::swap(c->mom_cur, c->mom_tb);
        std::swap(c->mu_cur, c->mu_tb);
    } else {
        std::swap(c->mom_cur, c->mom_ta);
        if (take_mu) std::swap(c->mu_cur, c->mu_ta);
    }
    if (take_mu) c->pure_c = -1;
    c->have_b = false;
    c->have_moments = true;
    return 0;
}

int vilma_delta_sums(vilma_ctx *c, void *stream, double *sums_dev, int which) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (which == VILMA_STATE_CURRENT && !c->have_moments)
        return fail(c, "no accepted evaluation of the current state");
    if (which != VILMA_STATE_CURRENT && !c->ready) return fail(c, "no trial state");
    DeltaArgs a;
    if (which == VILMA_STATE_TRIAL_BETA_B && !c->have_b) return fail(c, "no second candidate");
    // the trial of a beta step has its own vi_mu; the trial of a plain evaluation shares it
    fill_delta_args(c, a, c->delta_partials,
                    which == VILMA_STATE_TRIAL_BETA ? 1 : which == VILMA_STATE_TRIAL_BETA_B ? 2 : 0,
                    which == VILMA_STATE_CURRENT ? 0 : which == VILMA_STATE_TRIAL_BETA_B ? 2 : 1);
    // a trial state's sums depend on its per-SNP pass alone: overlap them with its LD product
    hipStream_t s = (hipStream_t)stream;
    hipStream_t q = which == VILMA_STATE_CURRENT ? s : side_begin(c, s);
    launch_delta_sums(a, sums_dev, q);
    side_end(c, s, q);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int vilma_trial_sums_available(const vilma_ctx *c) { return c ? c->tile_sums_ns : 0; }

int vilma_trial_sums(vilma_ctx *c, void *stream, double *sums_a_dev, double *sums_b_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (c->tile_sums_ns < 1) return fail(c, "the last trial left no per-tile sums (use vilma_delta_sums)");
    if (!sums_a_dev) return fail(c, "vilma_trial_sums: sums_a_dev is required");
    if (sums_b_dev && c->tile_sums_ns < 2) return fail(c, "no second candidate");
    reduce_tile_sums(c, (hipStream_t)stream, sums_a_dev, sums_b_dev);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int vilma_mean_diff(vilma_ctx *c, void *stream, double *out_sum3_dev, double *out_max3_dev) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!c->have_moments) return fail(c, "no accepted evaluation of the current state");
    hipStream_t s = (hipStream_t)stream;
    hipStream_t q = side_begin(c, s);
    launch_mean_diff(c->m[c->mom_cur], c->scal, c->snap[c->snap_cur], (int64_t)c->P * c->N,
                     c->diff_partials, out_sum3_dev, out_max3_dev, true, q);
    side_end(c, s, q);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int vilma_snapshot_mean(vilma_ctx *c, void *stream) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;      // nothing may be queued ahead of the state in use
    if (!c->have_moments) return fail(c, "no accepted evaluation of the current state");
    c->snp_marked = false;      // the snapshot is written on `stream`: a following mean_diff stays there
    launch_mean_diff(c->m[c->mom_cur], c->scal, c->snap[c->snap_cur], (int64_t)c->P * c->N,
                     c->diff_partials, nullptr, nullptr, false, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int vilma_fetch(vilma_ctx *c, void *stream, const double *src_dev, double *dst_host, int64_t n) {
    if (!c) return 1;
    if (n <= 0) return 0;
    if (n > c->pinned_elems) {
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr;
        c->pinned_elems = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->pinned, (size_t)n * sizeof(double), hipHostMallocDefault));
        c->pinned_elems = n;
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipMemcpyAsync(c->pinned, src_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::memcpy(dst_host, c->pinned, (size_t)n * sizeof(double));
    return 0;
}

int vilma_prof_enable(vilma_ctx *c, int enable) {
    if (!c) return 1;
    c->prof = enable > 0 ? enable : 0;
    c->prof_tick = 0;
    return 0;
}

int vilma_prof_read(vilma_ctx *c, double *ms_total, int64_t *launches, int reset) {
    if (!c) return 1;
    HIPCHK(c, hipDeviceSynchronize());
    prof_resolve(c);
    for (int k = 0; k < VILMA_PROF_KINDS; ++k) {
        if (ms_total) ms_total[k] = c->prof_ms[k];
        if (launches) launches[k] = c->prof_launches[k];
        if (reset) { c->prof_ms[k] = 0.0; c->prof_launches[k] = 0; }
    }
    return 0;
}

int vilma_prof_stream_store(vilma_ctx *c, void *stream, int passes, double *ms_per_pass,
                            int64_t *bytes_per_pass) {
    if (!c) return 1;
    if (passes < 1) return fail(c, "passes must be positive");
    if (vilma_sweep_drain(c)) return 1;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    int64_t bytes = 0;
    double total = 0.0;
    for (int it = 0; it <= passes; ++it) {          // the first pass warms up and is not counted
        bytes = 0;
        HIPCHK(c, hipEventRecord(e0, st));
        for (const CohortLd &co : c->ld) {
            const int64_t chunks = co.store_used / 4096;        // whole 32 KB steps
            if (!co.store || chunks == 0) continue;
            hipLaunchKernelGGL(store_stream_kernel, dim3((unsigned)std::min<int64_t>(chunks, 4096)),
                               dim3(256), 0, st, (const probe_v2d *)co.store, chunks, c->diff_partials);
            bytes += chunks * 32768;
        }
        HIPCHK(c, hipEventRecord(e1, st));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (it > 0) total += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms_per_pass) *ms_per_pass = total / passes;
    if (bytes_per_pass) *bytes_per_pass = bytes;
    return 0;
}

int vilma_prof_stream_pattern(vilma_ctx *c, void *stream, int passes, int chunk_kb, int scattered,
                              int grid, int writes, double *ms_per_pass, int64_t *bytes_per_pass) {
    if (!c) return 1;
    if (passes < 1 || chunk_kb < 32 || chunk_kb % 32 || grid < 1)
        return fail(c, "vilma_prof_stream_pattern: passes >= 1, chunk_kb a multiple of 32, grid >= 1");
    if (vilma_sweep_drain(c)) return 1;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    const int steps = chunk_kb / 32;
    int64_t bytes = 0;
    double total = 0.0;
    // the sprinkled writes land in the product scratch (2 x s_stride doubles): one store of 8
    // doubles per 8 KiB read = store bytes / 128
    double *wbuf = nullptr;
    if (writes) {
        int64_t need = 0;
        for (const CohortLd &co : c->ld) need = std::max<int64_t>(need, co.store_used / 128 + 64 + (int64_t)grid * (chunk_kb + 32));
        if (ensure_ready(c)) return 1;
        if (need > 2 * c->s_stride) return fail(c, "the product scratch is too small for the write probe");
        wbuf = c->sym_scratch;
    }
    for (int it = 0; it <= passes; ++it) {          // the first pass warms up and is not counted
        bytes = 0;
        HIPCHK(c, hipEventRecord(e0, st));
        for (const CohortLd &co : c->ld) {
            const int64_t chunks = co.store_used / (4096 * (int64_t)steps);
            if (!co.store || chunks == 0) continue;
            int64_t mult = 1;
            if (scattered) {            // an odd multiplier near chunks * 0.618, coprime to chunks
                mult = ((int64_t)(0.6180339887 * (double)chunks)) | 1;
                auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
                while (gcd(mult, chunks) != 1) mult += 2;
            }
            hipLaunchKernelGGL(store_pattern_kernel, dim3((unsigned)std::min<int64_t>(chunks, grid)),
                               dim3(256), 0, st, (const probe_v2d *)co.store, chunks, steps, mult,
                               c->diff_partials, writes, wbuf,
                               // mode 6: room per workgroup for everything it would have stored
                               (int64_t)((chunks * steps + std::min<int64_t>(chunks, grid) - 1) /
                                         std::min<int64_t>(chunks, grid) + steps) * 32);
            bytes += chunks * steps * 32768;
        }
        HIPCHK(c, hipEventRecord(e1, st));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (it > 0) total += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms_per_pass) *ms_per_pass = total / passes;
    if (bytes_per_pass) *bytes_per_pass = bytes;
    return 0;
}

int vilma_prof_ld_tile(vilma_ctx *c, int *tile_rows, int *tile_slabs, int *n_items) {
    if (!c) return 1;
    if (ensure_ready(c)) return 1;
    if (tile_rows) *tile_rows = c->tile_rows;
    if (tile_slabs) *tile_slabs = c->tile_rows > 0 ? c->tile_slabs : 0;
    if (n_items) *n_items = c->tile_rows > 0 ? c->all.n_tile : c->all.n_sym;
    return 0;
}

int vilma_prof_ld_order(vilma_ctx *c, int order) {
    if (!c) return 1;
    if (order < 0 || order > 2) return fail(c, "vilma_prof_ld_order: order must be 0, 1 or 2");
    if (vilma_sweep_drain(c)) return 1;
    HIPCHK(c, hipDeviceSynchronize());
    c->ld_order = order;
    if (!c->ready) return 0;
    // the same items in another order: offsets, scratch and partial slots do not move
    std::vector<int32_t> dstart;
    int32_t slot = 0, t_base = 0, s_base = 0;
    return build_item_lists(c, dstart, slot, t_base, s_base);
}

int vilma_prof_ld_trace(vilma_ctx *c, double *buf_dev, int64_t capacity_rows) {
    if (!c) return 1;
    HIPCHK(c, hipDeviceSynchronize());
    if (set_ld_trace(buf_dev, capacity_rows))
        return fail(c, "this build of the library has no LD trace (compile with -DLD_TRACE=1)");
    return 0;
}

}  // extern "C"
