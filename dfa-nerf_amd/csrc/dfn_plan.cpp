// dfn_plan.cpp - host-side pack planner: for every element of the kernel-ready weight stream, which entry
// of the flat decoder parameter buffer it is (or -1 for a structural zero).
//
// The op order below MUST mirror dfn_mlp.h (mlp_head / mlp_torso / mlp_trunk): each gemm_group there
// consumes G*KU fragments in [ku][g] order; here emit_group() appends the same fragments.  The plan is pure
// host code (no GPU needed), so tests/test_pack_plan.py can emulate the kernel's dataflow on the CPU
// from the plan alone and compare with the oracle.
#include <cstdint>
#include <functional>
#include <vector>

#include "dfn_layout.h"
#include "dfn_plan.h"

namespace dfn {

namespace {

struct Src {
    int pid;    // parameter tensor, or -1 = zero
    int row;
};
using RowFn = std::function<Src(int out_row)>;    // output feature -> (weight tensor, row)
using ColFn = std::function<int(int slot)>;        // input-vector slot -> weight column, or -1

struct Builder {
    int tier;
    std::vector<int32_t>& plan;
    long frags = 0;

    // one tile group of G output tiles starting at output row row0, over KU k-units of one input vector
    void emit_group(int G, int row0, int KU, const RowFn& rows, const ColFn& cols) {
        const int E = tier_E(tier);
        for (int ku = 0; ku < KU; ++ku)
            for (int g = 0; g < G; ++g) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const Src src = rows(row0 + 32 * g + i);
                    for (int e = 0; e < E; ++e) {
                        const int slot = kslot_to_slot(tier, ku, h, e);
                        const int col = cols(slot);
                        int32_t v = -1;
                        if (src.pid >= 0 && col >= 0) {
                            const ParamShape sh = param_shape(src.pid);
                            if (src.row < sh.rows && col < sh.cols)
                                v = param_offset(src.pid) + src.row * sh.cols + col;
                        }
                        plan.push_back(v);
                    }
                }
                ++frags;
            }
    }
    // layer(): OT tiles in pairs over one vector
    void emit_layer(int OT, int KU, const RowFn& rows, const ColFn& cols) {
        for (int tg = 0; tg < OT / 2; ++tg) emit_group(2, 64 * tg, KU, rows, cols);
    }
    // layer_skip(): per pair, main segment then skip segment
    void emit_layer_skip(int OT, int KU, const RowFn& rows, const ColFn& cols, int KU2, const RowFn& rows2,
                         const ColFn& cols2) {
        for (int tg = 0; tg < OT / 2; ++tg) {
            emit_group(2, 64 * tg, KU, rows, cols);
            emit_group(2, 64 * tg, KU2, rows2, cols2);
        }
    }
};

RowFn rows_of(int pid) {
    return [pid](int r) { return Src{pid, r}; };
}
ColFn ident(int ncols, int col0 = 0) {
    return [ncols, col0](int s) { return s < ncols ? col0 + s : -1; };
}

// the trunk shared by all fields: L1..L3, L4+skip, L5..L7, view(+sigma), out
void emit_trunk(Builder& b, int UPT, int pid_pskip, int KU_P, const ColFn& pcols) {
    const int KU_ACT = 8 * UPT, KU_VIEW = UPT;
    const ColFn act = ident(HID);
    const int blk[7] = {P_BLK0_W, P_BLK1_W, P_BLK2_W, P_BLK3_W, P_BLK4_W, P_BLK5_W, P_BLK6_W};
    for (int l = 0; l < 3; ++l) b.emit_layer(8, KU_ACT, rows_of(blk[l]), act);
    b.emit_layer_skip(8, KU_ACT, rows_of(blk[3]), act, KU_P, rows_of(pid_pskip), pcols);
    for (int l = 4; l < 7; ++l) b.emit_layer(8, KU_ACT, rows_of(blk[l]), act);
    // view layer: 4 pairs of feat_view rows, each [act segment, view segment]; then the sigma tile
    const ColFn view = [](int s) { return view_slot_to_ref(s); };
    for (int tg = 0; tg < 4; ++tg) {
        b.emit_group(2, 64 * tg, KU_ACT, rows_of(P_FEATV_W), act);
        b.emit_group(2, 64 * tg, KU_VIEW, rows_of(P_FCV_W), view);
    }
    const RowFn sig_rows = [](int r) { return r == 0 ? Src{P_SIGMA_W, 0} : Src{-1, 0}; };
    const RowFn none = [](int) { return Src{-1, 0}; };
    b.emit_group(1, 0, KU_ACT, sig_rows, act);
    b.emit_group(1, 0, KU_VIEW, none, view);
    // rgb head
    const RowFn out_rows = [](int r) { return r < 3 ? Src{P_FEATO_W, r} : Src{-1, 0}; };
    b.emit_group(1, 0, KU_ACT, out_rows, act);
}

}  // namespace

long build_pack_plan(int tier, int field, std::vector<int32_t>& plan) {
    plan.clear();
    Builder b{tier, plan};
    const int UPT = tier_UPT(tier);
    const int KU_PE = 2 * UPT, KU_D = 2 * UPT, KU_PD = 4 * UPT;
    const ColFn pe = [](int s) { return pe_slot_to_ref(s); };
    if (field == FIELD_HEAD || field == 2 /* listener */) {
        const int pid_in = field == FIELD_HEAD ? P_FCIN_W : P_FCINL_W;
        const int pid_sk = field == FIELD_HEAD ? P_FCPSK_W : P_FCPSKL_W;
        b.emit_layer(8, KU_PE, rows_of(pid_in), pe);
        emit_trunk(b, UPT, pid_sk, KU_PE, pe);
    } else {
        const ColFn d64 = ident(DH);
        b.emit_layer(2, KU_PE, rows_of(P_DE0_W), pe);                      // E0
        b.emit_layer(2, KU_PE, rows_of(P_DS0_W), pe);                      // S0
        b.emit_layer(2, KU_D, rows_of(P_DE1_W), d64);                      // E1
        b.emit_layer(2, KU_D, rows_of(P_DS1_W), d64);                      // S1
        b.emit_layer(2, KU_D, rows_of(P_DE2_W), d64);                      // E2
        b.emit_layer(2, KU_D, rows_of(P_DS2_W), d64);                      // S2
        b.emit_layer_skip(2, KU_D, rows_of(P_DE3_W), d64, KU_PE, rows_of(P_DESK_W), pe);   // E3 + ESKIP
        b.emit_layer(2, KU_D, rows_of(P_DS3_W), d64);                      // S3 (its skip is a constant)
        b.emit_layer(2, KU_D, rows_of(P_DE4_W), d64);                      // E4
        b.emit_layer(2, KU_D, rows_of(P_DS4_W), d64);                      // S4
        b.emit_layer(2, KU_D, rows_of(P_DEO_W), d64);                      // EO (60 rows valid)
        b.emit_layer(2, KU_D, rows_of(P_DSO_W), d64);                      // SO (42 rows valid)
        // pd: slots 0..59 deformed PE -> columns 0..59 ; slots 64..105 deformed signal -> columns 60..101
        const ColFn pd = [](int s) {
            if (s < NPE) return s;
            if (s >= 64 && s < 64 + NET) return NPE + (s - 64);
            return -1;
        };
        b.emit_layer(8, KU_PD, rows_of(P_FCINT_W), pd);
        emit_trunk(b, UPT, P_FCPSKT_W, KU_PD, pd);
    }
    // pad to whole slabs (the kernel always DMA-loads whole slabs)
    const long frag_elems = 64L * tier_E(tier);
    const long n_frags = b.frags;
    const long padded = (n_frags + SLAB_FRAGS - 1) / SLAB_FRAGS * SLAB_FRAGS;
    plan.resize(padded * frag_elems, -1);
    return n_frags;
}

}  // namespace dfn
