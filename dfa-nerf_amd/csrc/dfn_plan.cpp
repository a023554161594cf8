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
    // transposed group (backward streams): G tiles of forward-INPUT slots starting at slot0, contraction over the
    // forward-OUTPUT rows carried as KU k-units of a gradient vector
    void emit_group_T(int G, int slot0, int KU, const RowFn& rows, const ColFn& cols) {
        const int E = tier_E(tier);
        for (int ku = 0; ku < KU; ++ku)
            for (int g = 0; g < G; ++g) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int col = cols(slot0 + 32 * g + i);
                    for (int e = 0; e < E; ++e) {
                        const Src src = rows(kslot_to_slot(tier, ku, h, e));
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
    void emit_layer_T(int OT, int KU, const RowFn& rows, const ColFn& cols) {
        for (int tg = 0; tg < OT / 2; ++tg) emit_group_T(2, 64 * tg, KU, rows, cols);
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

// backward (transposed) stream: op order of dfn_bwd.h
void emit_bwd(Builder& b, int UPT, bool torso) {
    const int KU_ACT = 8 * UPT, KU_T = UPT, KU_D = 2 * UPT;
    const ColFn act = ident(HID);
    const ColFn pd = [](int s) {
        if (s < NPE) return s;
        if (s >= 64 && s < 64 + NET) return NPE + (s - 64);
        return -1;
    };
    const RowFn out_rows = [](int r) { return r < 3 ? Src{P_FEATO_W, r} : Src{-1, 0}; };
    const RowFn sig_rows = [](int r) { return r == 0 ? Src{P_SIGMA_W, 0} : Src{-1, 0}; };
    b.emit_layer_T(8, KU_T, out_rows, act);                                   // feat_out^T
    for (int tg = 0; tg < 4; ++tg) {                                          // [feat_view ; sigma_out]^T
        b.emit_group_T(2, 64 * tg, KU_ACT, rows_of(P_FEATV_W), act);
        b.emit_group_T(2, 64 * tg, KU_T, sig_rows, act);
    }
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK6_W), act);
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK5_W), act);
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK4_W), act);
    if (torso) b.emit_layer_T(4, KU_ACT, rows_of(P_FCPSKT_W), pd);            // skip path -> d pd
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK3_W), act);
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK2_W), act);
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK1_W), act);
    b.emit_layer_T(8, KU_ACT, rows_of(P_BLK0_W), act);
    if (torso) {
        const ColFn d64 = ident(DH);
        b.emit_layer_T(4, KU_ACT, rows_of(P_FCINT_W), pd);
        const int seq[10] = {P_DEO_W, P_DSO_W, P_DE4_W, P_DS4_W, P_DE3_W, P_DS3_W, P_DE2_W, P_DS2_W, P_DE1_W, P_DS1_W};
        for (int k = 0; k < 10; ++k) b.emit_layer_T(2, KU_D, rows_of(seq[k]), d64);
    }
}

}  // namespace

// ---- weight-gradient GEMM list, scatter map and bias-gradient rows (row constants: dfn_mlp.h RecMap,
// dfn_bwd.h GradMap; bias blob offsets: dfn_mlp.h Prog) -----------------------------------------------------------
namespace {
struct RM {   // RecMap
    static constexpr int T_A0 = 0, T_H = 8 * 256, T_VIEW = 9 * 256, T_ROWS = 9 * 256 + 32;
    static constexpr int H_PE = 0, H_TRUNK = 64;
    static constexpr int S_PE = 0, S_D0 = 64, S_PD = 64 + 640, S_TRUNK = S_PD + 128;
};
struct GM {   // GradMap
    static constexpr int T_DY0 = 0, T_DY4 = 4 * 256, T_G4 = 5 * 256, T_DY5 = 6 * 256, T_DYV = 9 * 256,
                         T_DSIG = 10 * 256, T_DYO = 10 * 256 + 32;
    static constexpr int S_DE0 = 0, S_DS0 = 64, S_DE1 = 128, S_DS1 = 192, S_DE2 = 256, S_DS2 = 320, S_DE3 = 384,
                         S_GE3 = 448, S_DS3 = 512, S_DE4 = 576, S_DS4 = 640, S_DEO = 704, S_DSO = 768, S_GS3 = 832,
                         S_TRUNK = 896;
};
}  // namespace

void build_wgrad_plan(int field, std::vector<WOpHost>& ops, std::vector<int32_t>& map, std::vector<int32_t>& bias_rows) {
    ops.clear();
    map.clear();
    bias_rows.clear();
    const bool torso = field == FIELD_TORSO;
    const ColFn pe = [](int s) { return pe_slot_to_ref(s); };
    const ColFn view = [](int s) { return view_slot_to_ref(s); };
    const ColFn act = ident(HID);
    const ColFn d64 = ident(DH);
    const ColFn pd = [](int s) {
        if (s < NPE) return s;
        if (s >= 64 && s < 64 + NET) return NPE + (s - 64);
        return -1;
    };
    auto add = [&](int a_row, int M, int b_row, int N, const RowFn& rows, const ColFn& cols) {
        int owner = 1;      // dy_T row ranges are either identical or disjoint between GEMMs
        for (const WOpHost& q : ops)
            if (q.a_row == a_row) owner = 0;
        WOpHost o{a_row, M, b_row, N, (int)map.size(), owner};
        ops.push_back(o);
        for (int m = 0; m < M; ++m) {
            const Src src = rows(m);
            for (int n = 0; n < N; ++n) {
                const int col = cols(n);
                int32_t v = -1;
                if (src.pid >= 0 && col >= 0) {
                    const ParamShape sh = param_shape(src.pid);
                    if (src.row < sh.rows && col < sh.cols) v = param_offset(src.pid) + src.row * sh.cols + col;
                }
                map.push_back(v);
            }
        }
    };
    const int gt = torso ? GM::S_TRUNK : 0, rt = torso ? RM::S_TRUNK : RM::H_TRUNK;
    const RowFn sig_rows = [](int r) { return r == 0 ? Src{P_SIGMA_W, 0} : Src{-1, 0}; };
    const RowFn out_rows = [](int r) { return r < 3 ? Src{P_FEATO_W, r} : Src{-1, 0}; };
    if (!torso) {
        // head: the positional-encoding columns of fc_in / fc_p_skips (the signal columns multiply a per-frame constant: their
        // gradient comes out of the fold's backward); listener (field 2, decoder.py:306-307, 322-323): its own two layers
        const bool lis = field == 2;
        add(gt + GM::T_DY0, 256, RM::H_PE, 64, rows_of(lis ? P_FCINL_W : P_FCIN_W), pe);
        add(gt + GM::T_G4, 256, RM::H_PE, 64, rows_of(lis ? P_FCPSKL_W : P_FCPSK_W), pe);
    } else {
        const int dE[5] = {GM::S_DE0, GM::S_DE1, GM::S_DE2, GM::S_DE3, GM::S_DE4};
        const int dS[5] = {GM::S_DS0, GM::S_DS1, GM::S_DS2, GM::S_DS3, GM::S_DS4};
        const int wE[5] = {P_DE0_W, P_DE1_W, P_DE2_W, P_DE3_W, P_DE4_W}, wS[5] = {P_DS0_W, P_DS1_W, P_DS2_W, P_DS3_W, P_DS4_W};
        add(dE[0], 64, RM::S_PE, 64, rows_of(wE[0]), pe);
        add(dS[0], 64, RM::S_PE, 64, rows_of(wS[0]), pe);
        for (int l = 1; l < 5; ++l) {       // layer l consumes ve_{l-1} / vs_{l-1} = deformation vectors 2(l-1), 2(l-1)+1
            add(dE[l], 64, RM::S_D0 + 64 * (2 * (l - 1)), 64, rows_of(wE[l]), d64);
            add(dS[l], 64, RM::S_D0 + 64 * (2 * (l - 1) + 1), 64, rows_of(wS[l]), d64);
        }
        add(GM::S_GE3, 64, RM::S_PE, 64, rows_of(P_DESK_W), pe);
        add(GM::S_DEO, 64, RM::S_D0 + 64 * 8, 64, rows_of(P_DEO_W), d64);
        add(GM::S_DSO, 64, RM::S_D0 + 64 * 9, 64, rows_of(P_DSO_W), d64);
        add(gt + GM::T_DY0, 256, RM::S_PD, 128, rows_of(P_FCINT_W), pd);
        add(gt + GM::T_G4, 256, RM::S_PD, 128, rows_of(P_FCPSKT_W), pd);
    }
    const int blk[7] = {P_BLK0_W, P_BLK1_W, P_BLK2_W, P_BLK3_W, P_BLK4_W, P_BLK5_W, P_BLK6_W};
    for (int l = 0; l < 7; ++l) {
        const int dy = (l < 4) ? GM::T_DY0 + 256 * (l + 1) : GM::T_DY5 + 256 * (l - 4);   // dy_{l+1}: DY4 sits at T_DY4
        add(gt + dy, 256, rt + RM::T_A0 + 256 * l, 256, rows_of(blk[l]), act);
    }
    add(gt + GM::T_DYV, 256, rt + RM::T_A0 + 256 * 7, 256, rows_of(P_FEATV_W), act);
    add(gt + GM::T_DSIG, 32, rt + RM::T_A0 + 256 * 7, 256, sig_rows, act);
    add(gt + GM::T_DYV, 256, rt + RM::T_VIEW, 32, rows_of(P_FCV_W), view);
    add(gt + GM::T_DYO, 32, rt + RM::T_H, 256, out_rows, act);

    // bias blob element e -> row of dy_T whose sum over points is its gradient (blob layout [tile][half][16])
    auto feat_of = [](int e) { return 32 * (e >> 5) + tile_feat((e >> 4) & 1, e & 15); };
    auto vec = [&](int row0, int n) {
        for (int e = 0; e < n; ++e) bias_rows.push_back(row0 + feat_of(e));
        // the fused weight+bias gradient pass produces a row block's sums in the GEMM that reads it; a block that no
        // GEMM reads (its layer multiplies a per-frame constant only, e.g. fc_signal_skips) gets a GEMM with N = 0
        bool read = false;
        for (const WOpHost& q : ops)
            if (q.a_row <= row0 && row0 + n <= q.a_row + q.M) read = true;
        if (!read) ops.push_back(WOpHost{row0, n, 0, 0, (int)map.size(), 1});
    };
    if (torso) {
        const int seq[14] = {GM::S_DE0, GM::S_DS0, GM::S_DE1, GM::S_DS1, GM::S_DE2, GM::S_DS2, GM::S_DE3, GM::S_GE3,
                             GM::S_DS3, GM::S_GS3, GM::S_DE4, GM::S_DS4, GM::S_DEO, GM::S_DSO};
        for (int k = 0; k < 14; ++k) vec(seq[k], 64);
    }
    vec(gt + GM::T_DY0, 256);                                   // IN
    for (int l = 1; l <= 4; ++l) vec(gt + GM::T_DY0 + 256 * l, 256);     // L1..L4 (DY4 = T_DY0 + 1024)
    vec(gt + GM::T_G4, 256);                                    // SKIP
    for (int l = 0; l < 3; ++l) vec(gt + GM::T_DY5 + 256 * l, 256);      // L5..L7
    vec(gt + GM::T_DYV, 256);                                   // VIEW: feat_view rows ...
    vec(gt + GM::T_DSIG, 32);                                   // ... + the sigma tile
    vec(gt + GM::T_DYO, 32);                                    // OUT
}

long build_bwd_plan(int tier, int field, std::vector<int32_t>& plan) {
    plan.clear();
    Builder b{tier, plan};
    emit_bwd(b, tier_UPT(tier), field == FIELD_TORSO);
    const long frag_elems = 64L * tier_E(tier);
    const long n_frags = b.frags;
    const long padded = (n_frags + SLAB_FRAGS - 1) / SLAB_FRAGS * SLAB_FRAGS;
    plan.resize(padded * frag_elems, -1);
    return n_frags;
}

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
