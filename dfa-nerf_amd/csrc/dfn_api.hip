// dfn_api.hip - the C ABI of libdfanerf.so (declared in include/dfanerf.h).
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "dfanerf.h"
#include "dfn_layout.h"
#include "dfn_misc.h"
#include "dfn_mlp.h"
#include "dfn_params.h"
#include "dfn_plan.h"
#include "dfn_signal.h"
#include "dfn_train.h"

using namespace dfn;

namespace {

thread_local std::string g_err;
unsigned long long* g_clock_probe = nullptr;      // dfn_debug_clock_probe (debug only)

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(DFN_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

bool tier_ok(int tier) { return tier == DFN_TIER_F32 || tier == DFN_TIER_BF16 || tier == DFN_TIER_F16; }
// the training entry points: the f16 tier is inference only (gradients underflow its 5-bit exponent)
bool train_tier_ok(int tier) { return tier == DFN_TIER_F32 || tier == DFN_TIER_BF16; }
bool field_ok(int field) { return field >= 0 && field <= 2; }
int prog_field(int field) { return field == DFN_FIELD_TORSO ? FIELD_TORSO : FIELD_HEAD; }
// training entry points: head, torso and (round 6) the listener, i.e. the head's program on fc_in_listener / fc_p_skips_listener
// (decoder.py:306-307, 322-323: `signal is None`).  Its dX chain IS the head's: the head's backward stream holds no input layer
// (the positional encoding gets no gradient), so the transposed weight stream and the kernel are shared (bwd_field).
bool train_field_ok(int field) { return field == DFN_FIELD_HEAD || field == DFN_FIELD_TORSO || field == DFN_FIELD_LISTENER; }
int bwd_field(int field) { return field == DFN_FIELD_TORSO ? 1 : 0; }

// cached pack plans: host copy + lazily uploaded device copy (per device the first caller uses)
struct PlanEntry {
    std::vector<int32_t> host;
    long n_frags = 0;
    int32_t* dev = nullptr;
};
std::mutex g_plan_mu;
PlanEntry g_plans[3][3];

PlanEntry& plan_of(int tier, int field) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    PlanEntry& e = g_plans[tier][field];
    if (e.host.empty()) e.n_frags = build_pack_plan(tier, field, e.host);
    return e;
}

}  // namespace

namespace {
struct BwdPlanEntry {
    std::vector<int32_t> host;
    long n_frags = 0;
    int32_t* dev = nullptr;
};
BwdPlanEntry g_bwd_plans[2][2];     // [tier][bwd_field]
struct WgradEntry {
    bool built = false;
    std::vector<WOpHost> ops;
    std::vector<int32_t> map, bias_rows;
    // f32 tier: the 256 x 256 GEMMs (wgrad_full_kernel) and the work items of every other GEMM (wgrad_narrow_kernel)
    std::vector<int> full_ops;
    std::vector<WNItem> nitems;
    std::string plan_error;           // a GEMM shape the f32 kernels are not instantiated for
    WOp* ops_dev = nullptr;
    int32_t* map_dev = nullptr;
    int32_t* rows_dev = nullptr;
    int32_t* eof_dev = nullptr;      // dy_T row -> bias element
    int* full_ops_dev = nullptr;
    WNItem* nitems_dev = nullptr;
    // 16-bit tier: one workgroup per (GEMM, slice of the points), the slice count PER GEMM (balanced split, see wgrad_items)
    // Two splits (WGRAD_SPLITS): [0] the reference's step (<= WGRAD_SMALL_NP points), [1] larger calls (the hierarchical step)
    WItem* items_dev[2] = {nullptr, nullptr};
    int n_items[2] = {0, 0};
    unsigned char* blk_n_dev[2] = {nullptr, nullptr};    // slices of the GEMM that owns each 256-element block of the dense C array
    unsigned char* bias_n_dev[2] = {nullptr, nullptr};   // slices of the GEMM that produces each bias element's row sum
    int32_t* sig_rows_dev = nullptr; // dfn_signal_grad: dy_T rows / bias elements behind d(signal)
    int32_t* sig_elems_dev = nullptr;
    int n_sig = 0;
    int ksplit_uploaded = 0;
};
WgradEntry g_wgrad[3];               // head, torso, listener
#ifndef DFN_WGRAD_KSPLIT_F32
#define DFN_WGRAD_KSPLIT_F32 32
#endif
constexpr int WGRAD_KSPLIT = DFN_WGRAD_KSPLIT_F32;
constexpr long WGRAD_SMALL_NP = 196608;  // 16-bit tier: calls up to this many points use the split with more spare compute units

#ifndef DFN_WS_KSPLIT_MAX
#define DFN_WS_KSPLIT_MAX 32
#endif
constexpr int WS_KSPLIT_MAX = DFN_WS_KSPLIT_MAX;       // slices the workspace is sized for (>= every tier's split)
static_assert(DFN_WGRAD_KSPLIT_F32 <= DFN_WS_KSPLIT_MAX, "the f32 tier's split fits the workspace");
// head 16 / torso 18 (round 3, whole step, interleaved A/B over 600 steps x 4: torso 16 / 17 / 18 / 20 = 1.1056 / 1.1021 / 1.0975 /
// 1.114 ms; head 19: worse).  Round 2: the kernel alone takes the same time for 16 ... 32 slices (0.81-0.82 ms for both fields, HBM-bound), the second stage
// reads a third less and the whole training step is 1.2 % faster than with 24 (interleaved A/B, bench.py --workload c4)
// Balanced split (round 4).  Round 3 cut every GEMM into the same 16 / 18 slices of the points: the head's 13 GEMMs made 208
// workgroups on 256 compute units - ONE round, whose length is the 256 x 256 GEMMs' (512 operand bytes per point, 128 steps)
// while the workgroups of the narrow GEMMs (288-320 bytes per point) finished early and 48 compute units had none.  Now the
// number of slices of a GEMM is proportional to its operand rows M + N, so that every workgroup streams about the same bytes
// and the launch fills the chip's compute units once: head 23 slices for a 256 x 256 GEMM (89 steps), 13-14 for the narrow
// ones.  DFN_WGRAD_KSPLIT[_H|_T] (developer overrides) select a uniform split instead.
void wgrad_items(const std::vector<WOpHost>& ops, int field, int target_wgs, std::vector<WItem>& items,
                 std::vector<int>& n_of) {
    const int uni = [&] {
        const char* e = getenv(field ? "DFN_WGRAD_KSPLIT_T" : "DFN_WGRAD_KSPLIT_H");
        if (!e) e = getenv("DFN_WGRAD_KSPLIT");
        const int k = e ? atoi(e) : 0;
        return k > 0 && k <= 32 ? k : 0;
    }();
    // cost of a GEMM per point: its operand bytes; a floor for the narrow ones (a step of theirs costs a barrier and a DMA
    // round trip whatever it moves)
    // (a GEMM with N = 0 only sums the rows of its dY block; it runs the general loop: measured 51 us where a 256 x 256 GEMM
    // takes 142 - tools/wl_trace.py)
    // Measured per-workgroup time x slices (tools/wl_trace.py with the MX-fp4 activations, us x slices / 5): a dY row costs 1, an
    // activation row 1/2 (32 vs 16 bytes per point tile) - 256 x 256: 380, 256 x 128: 326, 256 x 64: 286, 64 x 64: 94 - except
    // the shapes whose steps are latency- rather than byte-bound: 256 x 32: 244, 32 x 256: 164, N = 0 (M = 64): 180
    auto cost = [](const WOpHost& o) {
        if (o.N == 0) return 2.8 * o.M;
        if (o.M == 256 && o.N == 32) return 245.0;
        if (o.M == 32 && o.N == 256) return 170.0;
        return (double)std::max(o.M + o.N / 2, 90);
    };
    double total = 0;
    for (const WOpHost& o : ops) total += cost(o);
    n_of.assign(ops.size(), 1);
    int sum = 0;
    for (size_t i = 0; i < ops.size(); ++i) {
        n_of[i] = uni ? uni : std::min(32, std::max(1, (int)(target_wgs * cost(ops[i]) / total)));     // (floor: the sum stays <= target)
        sum += n_of[i];
    }
    // hand the workgroups the floor left over to the GEMMs with the most bytes per workgroup
    while (!uni && sum < target_wgs) {
        int best = -1;
        double worst = 0;
        for (size_t i = 0; i < ops.size(); ++i)
            if (n_of[i] < 32 && cost(ops[i]) / n_of[i] > worst) worst = cost(ops[i]) / n_of[i], best = (int)i;
        if (best < 0) break;
        ++n_of[best];
        ++sum;
    }
    std::vector<int> order(ops.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(ops[a]) / n_of[a] > cost(ops[b]) / n_of[b]; });
    items.clear();
    for (int op : order)
        for (int k = 0; k < n_of[op]; ++k) items.push_back(WItem{op, k, n_of[op], 0});
}
int wgrad_ksplit_bf16(int field) {     // slices of the points per GEMM, bf16 tier (DFN_WGRAD_KSPLIT[_H|_T]: developer overrides)
    static const int v[2] = {
        [] { const char* e = getenv("DFN_WGRAD_KSPLIT_H"); if (!e) e = getenv("DFN_WGRAD_KSPLIT"); const int k = e ? atoi(e) : 0; return k > 0 && k <= WS_KSPLIT_MAX ? k : 16; }(),
        [] { const char* e = getenv("DFN_WGRAD_KSPLIT_T"); if (!e) e = getenv("DFN_WGRAD_KSPLIT"); const int k = e ? atoi(e) : 0; return k > 0 && k <= WS_KSPLIT_MAX ? k : 18; }()};
    return v[field == FIELD_TORSO ? 1 : 0];
}

template <typename T> hipError_t upload(T** dev, const T* host, size_t n) {
    T* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, n * sizeof(T));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {          // never publish a table that was not filled
        (void)hipFree(d);
        return e;
    }
    *dev = d;
    return hipSuccess;
}
WgradEntry& wgrad_of(int field) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    WgradEntry& w = g_wgrad[field];
    if (!w.built) {
        build_wgrad_plan(field, w.ops, w.map, w.bias_rows);
        // f32 tier: every GEMM is cut into WGRAD_KSPLIT slices of the points (one partial array per slice: the reduction adds
        // the same number of slices for every element).  Items of the narrow launch: blocks of a shape's row tiles x slices,
        // the shapes with the longest workgroups first (dfn_train.h: WN_4x4 < WN_4x2 < ... is that order)
        for (int shape = 0; shape < WN_COUNT; ++shape)
            for (size_t i = 0; i < w.ops.size(); ++i) {
                const WOpHost& o = w.ops[i];
                const int sh = wn_shape_of(o.M, o.N);
                if (sh == -2) w.plan_error = "weight-gradient GEMM " + std::to_string(o.M) + " x " + std::to_string(o.N) + ": no f32 kernel for this shape";
                if (sh == -1 && shape == 0) w.full_ops.push_back((int)i);
                if (sh != shape) continue;
                for (int m0 = 0; m0 < o.M / 32; m0 += wn_shape_mt(shape))
                    for (int ks = 0; ks < WGRAD_KSPLIT; ++ks) w.nitems.push_back(WNItem{(int)i, ks, m0, shape});
            }
        w.built = true;
    }
    return w;
}
}  // namespace

extern "C" {

const char* dfn_last_error(void) { return g_err.c_str(); }
#ifdef DFN_DEV_BUILD      // (dfn_devguard.h: a library built with developer switches says so, and dfanerf._lib refuses it in-tree)
const char* dfn_version(void) { return "dfanerf 0.2 gfx950 DEV"; }
#else
const char* dfn_version(void) { return "dfanerf 0.2 gfx950"; }
#endif

long dfn_packed_bytes(int tier, int field) {
    if (!tier_ok(tier) || !field_ok(field)) return fail(DFN_E_ARG, "dfn_packed_bytes: bad tier/field");
    ProgramInfo pi;
    program_info(tier, prog_field(field), &pi);
    return (long)pi.n_slabs * SLAB_BYTES;
}

long dfn_pack_plan(int tier, int field, int32_t* plan_host, long capacity) {
    if (!tier_ok(tier) || !field_ok(field)) return fail(DFN_E_ARG, "dfn_pack_plan: bad tier/field");
    PlanEntry& e = plan_of(tier, field);
    ProgramInfo pi;
    program_info(tier, prog_field(field), &pi);
    if (e.n_frags != pi.n_frags)
        return fail(DFN_E_ARG, "dfn_pack_plan: planner and kernel disagree on the fragment count (" +
                                   std::to_string(e.n_frags) + " vs " + std::to_string(pi.n_frags) + ")");
    const long n = (long)e.host.size();
    if (plan_host) {
        if (capacity < n) return fail(DFN_E_SIZE, "dfn_pack_plan: capacity too small");
        std::memcpy(plan_host, e.host.data(), n * sizeof(int32_t));
    }
    return n;
}

// device copy of the forward pack plan of (tier, field), uploaded on first use
static int fwd_plan_dev(int tier, int field, const int32_t** dev, long* n_out) {
    const long n = dfn_pack_plan(tier, field, nullptr, 0);
    if (n < 0) return (int)n;
    PlanEntry& e = plan_of(tier, field);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!e.dev) {
        hipError_t err = hipMalloc((void**)&e.dev, n * sizeof(int32_t));
        if (err != hipSuccess) return hip_fail(err, "hipMalloc(plan)");
        err = hipMemcpy(e.dev, e.host.data(), n * sizeof(int32_t), hipMemcpyHostToDevice);
        if (err != hipSuccess) return hip_fail(err, "hipMemcpy(plan)");
    }
    *dev = e.dev;
    *n_out = n;
    return DFN_OK;
}

int dfn_pack_weights(int tier, int field, const float* params, void* packed, void* stream) {
    if (!tier_ok(tier) || !field_ok(field) || !params || !packed)
        return fail(DFN_E_ARG, "dfn_pack_weights: bad argument");
    if (param_offset(P_COUNT) != N_DECODER_PARAMS) return fail(DFN_E_ARG, "internal: parameter table size");
    const int32_t* plan;
    long n;
    const int rc = fwd_plan_dev(tier, field, &plan, &n);
    if (rc != DFN_OK) return rc;
    hipError_t err = launch_pack(plan, params, packed, n, tier, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "pack_kernel");
    return DFN_OK;
}

long dfn_bias_floats(int tier, int field) {
    if (!tier_ok(tier) || !field_ok(field)) return fail(DFN_E_ARG, "dfn_bias_floats: bad tier/field");
    ProgramInfo pi;
    program_info(tier, prog_field(field), &pi);
    return pi.n_bias;
}

int dfn_fold_bias(int tier, int field, const float* params, const float* signal, const float* z_shape,
                  const float* z_app, float* bias, void* stream) {
    if (!tier_ok(tier) || !field_ok(field) || !params || !z_shape || !z_app || !bias)
        return fail(DFN_E_ARG, "dfn_fold_bias: bad argument");
    if (field != DFN_FIELD_LISTENER && !signal) return fail(DFN_E_ARG, "dfn_fold_bias: signal is NULL");
    const int n = (int)dfn_bias_floats(tier, field);
    hipError_t err = launch_fold(field, params, signal, z_shape, z_app, bias, n, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "fold_kernel");
    return DFN_OK;
}

int dfn_fold_bias_bwd(int tier, int field, const float* params, const float* signal, const float* z_shape,
                      const float* z_app, const float* dbias, float* grad_flat, float* d_signal, void* stream) {
    if (!tier_ok(tier) || !field_ok(field) || !params || !z_shape || !z_app || !dbias || !grad_flat)
        return fail(DFN_E_ARG, "dfn_fold_bias_bwd: bad argument");
    if (field != DFN_FIELD_LISTENER && !signal) return fail(DFN_E_ARG, "dfn_fold_bias_bwd: signal is NULL");
    const int n = (int)dfn_bias_floats(tier, field);
    hipError_t err = launch_fold_bwd(field, params, signal, z_shape, z_app, dbias, grad_flat, d_signal, n,
                                     (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "fold_bwd_kernel");
    return DFN_OK;
}

int dfn_adam_multi(const DfnAdamItem* items_dev, const int32_t* chunks_dev, int n_chunks, float lr, double beta1,
                   double beta2, float eps, float bias_c1, float bias_c2_sqrt, void* stream) {
    if (n_chunks < 0 || (n_chunks > 0 && (!items_dev || !chunks_dev)) || !(bias_c1 > 0.f) || !(bias_c2_sqrt > 0.f))
        return fail(DFN_E_ARG, "dfn_adam_multi: bad argument");
    hipError_t err = launch_adam_multi(items_dev, chunks_dev, n_chunks, lr, beta1, beta2, eps, bias_c1, bias_c2_sqrt,
                                       (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "adam_multi_kernel");
    return DFN_OK;
}

static int render_fwd_impl(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                           const float* bias_head, const float* bias_torso, const float* bg_f32,
                           const uint8_t* bg_u8, const int32_t* pix_index, float* rgb_head, float* rgb_com,
                           float* weights_head, float* weights_com, float* z_vals, int out_u8, void* stream) {
    if (!tier_ok(tier) || !frame || !packed_head || !bias_head || !rgb_head)
        return fail(DFN_E_ARG, "dfn_render_fwd: bad argument");
    const DfnFrame& F = *frame;
    // --N_samples (MAIN:612-619): 32, 64 or 128 coarse samples; the hierarchical sampler (a 64-lane wave program) needs 64
    if (F.n_coarse != 32 && F.n_coarse != 64 && F.n_coarse != 128) return fail(DFN_E_ARG, "dfn_render_fwd: n_coarse must be 32, 64 or 128");
    if (F.n_fine != 0 && F.n_fine != 64 && F.n_fine != 128)
        return fail(DFN_E_ARG, "dfn_render_fwd: n_fine must be 0, 64 or 128");
    if (F.n_fine != 0 && F.n_coarse != 64) return fail(DFN_E_ARG, "dfn_render_fwd: the hierarchical mode (n_fine > 0) needs n_coarse = 64");
    if (F.fields != 1 && F.fields != 2) return fail(DFN_E_ARG, "dfn_render_fwd: fields must be 1 or 2");
    if (F.fields == 2 && (!packed_torso || !bias_torso || !rgb_com))
        return fail(DFN_E_ARG, "dfn_render_fwd: torso inputs / rgb_com missing for fields == 2");
    if (!bg_f32 && !bg_u8) return fail(DFN_E_ARG, "dfn_render_fwd: no background given");
    if (F.ray_count <= 0) return DFN_OK;
    if (F.H <= 0 || F.W <= 0 || (!pix_index && (F.ray_begin < 0 || F.ray_begin + F.ray_count > F.H * F.W)))
        return fail(DFN_E_ARG, "dfn_render_fwd: ray range outside the image");
    // the kernel reads [head | torso] biases from one LDS image: they must be adjacent in memory
    ProgramInfo ph, pt;
    program_info(tier, FIELD_HEAD, &ph);
    program_info(tier, FIELD_TORSO, &pt);
    if (F.fields == 2 && bias_torso != bias_head + ph.n_bias)
        return fail(DFN_E_ARG, "dfn_render_fwd: bias_torso must directly follow bias_head in memory");
    RenderArgs A;
    A.frame = F;
    A.wblob[0] = (const char*)packed_head;
    A.wblob[1] = (const char*)(F.fields == 2 ? packed_torso : packed_head);
    A.nslab[0] = ph.n_slabs;
    A.nslab[1] = F.fields == 2 ? pt.n_slabs : ph.n_slabs;
    A.bias = bias_head;
    A.bg_f32 = bg_f32;
    A.bg_u8 = bg_u8;
    A.pix_index = pix_index;
    A.rgb_head = rgb_head;
    A.rgb_com = rgb_com;
    A.w_head = weights_head;
    A.w_com = weights_com;
    A.z_out = z_vals;
    A.out_u8 = out_u8;
    A.samples_out = nullptr;
    A.ranks_out = nullptr;
    A.act_T[0] = A.act_T[1] = nullptr;
    A.masks[0] = A.masks[1] = nullptr;
    A.NP = 0;
    A.act_e4m3 = 0;
    A.loss = DfnTrainLoss{};
    A.clock_probe = g_clock_probe;
    hipError_t err = launch_render(tier, A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "render_kernel");
    return DFN_OK;
}

int dfn_render_fwd(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                   const float* bias_head, const float* bias_torso, const float* bg_f32,
                   const uint8_t* bg_u8, const int32_t* pix_index, float* rgb_head, float* rgb_com,
                   float* weights_head, float* weights_com, float* z_vals, void* stream) {
    return render_fwd_impl(tier, frame, packed_head, packed_torso, bias_head, bias_torso, bg_f32, bg_u8, pix_index,
                           rgb_head, rgb_com, weights_head, weights_com, z_vals, 0, stream);
}

int dfn_render_fwd_u8(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                      const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                      const int32_t* pix_index, uint8_t* rgb8_head, uint8_t* rgb8_com, void* stream) {
    return render_fwd_impl(tier, frame, packed_head, packed_torso, bias_head, bias_torso, bg_f32, bg_u8, pix_index,
                           (float*)rgb8_head, (float*)rgb8_com, nullptr, nullptr, nullptr, 1, stream);
}

// ---- training ---------------------------------------------------------------------------------------------------
long dfn_train_rows(int field, int what) {
    if (!train_field_ok(field)) return fail(DFN_E_ARG, "dfn_train_rows: bad field");
    const bool t = field == DFN_FIELD_TORSO;
    switch (what) {
    case 0: return t ? 64 + 640 + 128 + 9 * 256 + 32 : 64 + 9 * 256 + 32;          // activation rows (RecMap)
    case 1: return t ? 896 + 10 * 256 + 64 : 10 * 256 + 64;                          // gradient rows (GradMap)
    case 2: return t ? 10 + 36 : 36;                                                  // mask dwords per pass
    case 3: {        // workspace floats: split-K partials of the weight gradients + of the bias gradients
        const long W = (long)wgrad_of(field).map.size(), nb = dfn_bias_floats(DFN_TIER_BF16, field);
        return WS_KSPLIT_MAX * W + std::max(WS_KSPLIT_MAX, BIAS_GRAD_SLICES) * nb;
    }
    case 4: return (long)BIAS_GRAD_SLICES * dfn_bias_floats(DFN_TIER_BF16, field);   // dfn_bias_grad workspace floats
    case 5: return (long)SIG_ROW_SLICES * 512 + dfn_bias_floats(DFN_TIER_BF16, field);   // dfn_signal_grad workspace floats
    // 16-bit tier: bytes per 32-point tile of the MX-fp8 arrays act_T / dy_T (rows x 32 e4m3 bytes + the scale block)
    case 6: return act_tile_bytes(t ? 64 + 640 + 128 + 9 * 256 + 32 : 64 + 9 * 256 + 32, true);      // act_T of the FUSED step in its default format (MX-fp4: 16 bytes per row)
    case 7: return rec8_tile_bytes(t ? 896 + 10 * 256 + 64 : 10 * 256 + 64);
    // act_T in e4m3 (rows x 32 bytes + the scale block): dfn_decoder_train_fwd (Decoder.forward on explicit points under autograd)
    // and the fused step with DFN_TRAIN_ACT_E4M3
    case 8: return act_tile_bytes(t ? 64 + 640 + 128 + 9 * 256 + 32 : 64 + 9 * 256 + 32, false);
    default: return fail(DFN_E_ARG, "dfn_train_rows: bad selector");
    }
}

long dfn_packed_bwd_bytes(int tier, int field) {
    if (!train_tier_ok(tier) || !train_field_ok(field)) return fail(DFN_E_ARG, "dfn_packed_bwd_bytes: bad tier/field");
    ProgramInfo pi;
    bwd_program_info(tier, bwd_field(field), &pi);
    return (long)pi.n_slabs * SLAB_BYTES;
}

// device copy of the transposed (backward) pack plan of (tier, field), built and uploaded on first use
static int bwd_plan_dev(int tier, int field, const int32_t** dev, long* n_out) {
    BwdPlanEntry& e = g_bwd_plans[tier][field];
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (e.host.empty()) {
        e.n_frags = build_bwd_plan(tier, field, e.host);
        ProgramInfo pi;
        bwd_program_info(tier, field, &pi);
        if (e.n_frags != pi.n_frags)
            return fail(DFN_E_ARG, "backward planner and kernel disagree on the fragment count (" +
                                       std::to_string(e.n_frags) + " vs " + std::to_string(pi.n_frags) + ")");
    }
    if (!e.dev) {
        hipError_t err = upload(&e.dev, e.host.data(), e.host.size());
        if (err != hipSuccess) return hip_fail(err, "upload(bwd plan)");
    }
    *dev = e.dev;
    *n_out = (long)e.host.size();
    return DFN_OK;
}

int dfn_pack_weights_bwd(int tier, int field, const float* params, void* packed_T, void* stream) {
    if (!train_tier_ok(tier) || !train_field_ok(field) || !params || !packed_T)
        return fail(DFN_E_ARG, "dfn_pack_weights_bwd: bad argument");
    const int32_t* plan;
    long n;
    const int rc = bwd_plan_dev(tier, bwd_field(field), &plan, &n);
    if (rc != DFN_OK) return rc;
    hipError_t err = launch_pack(plan, params, packed_T, n, tier, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "pack_kernel(bwd)");
    return DFN_OK;
}

int dfn_train_prepare(int tier, const float* params, const float* signal_head, const float* signal_torso,
                      const float* z_shape, const float* z_app, void* packed_head, void* packed_torso, void* packed_T_head,
                      void* packed_T_torso, float* bias_head, float* bias_torso, void* stream) {
    if (!train_tier_ok(tier) || !params || !signal_head || !signal_torso || !z_shape || !z_app || !packed_head ||
        !packed_torso || !packed_T_head || !packed_T_torso || !bias_head || !bias_torso)
        return fail(DFN_E_ARG, "dfn_train_prepare: bad argument");
    if (param_offset(P_COUNT) != N_DECODER_PARAMS) return fail(DFN_E_ARG, "internal: parameter table size");
    PrepareJobs J{};
    J.params = params;
    J.tier = tier;
    void* outs[4] = {packed_head, packed_torso, packed_T_head, packed_T_torso};
    for (int k = 0; k < 4; ++k) {
        const int32_t* plan;
        const int rc = k < 2 ? fwd_plan_dev(tier, k, &plan, &J.n[k]) : bwd_plan_dev(tier, k - 2, &plan, &J.n[k]);
        if (rc != DFN_OK) return rc;
        J.plan[k] = plan;
        J.out[k] = outs[k];
    }
    J.sig[0] = signal_head;  J.sig[1] = signal_torso;
    J.bias[0] = bias_head;   J.bias[1] = bias_torso;
    for (int f = 0; f < 2; ++f) {
        J.zs[f] = z_shape + 256 * f;
        J.za[f] = z_app + 256 * f;
        J.nb[f] = (int)dfn_bias_floats(tier, f);
    }
    hipError_t err = launch_prepare(J, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "prepare_kernel");
    return DFN_OK;
}

static int train_fwd_impl(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                          const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                          const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                          uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, float* z_all, uint8_t* ranks,
                          bool hier, void* stream, const DfnTrainLoss* loss = nullptr, bool with_loss = false) {
    const char* who = hier ? "dfn_train_fwd_hier" : "dfn_train_fwd";
    // the format flag rides in the tier argument (include/dfanerf.h: DFN_TRAIN_ACT_E4M3); 16-bit tier only
    const int act_e4m3 = (tier & DFN_TRAIN_ACT_E4M3) != 0;
    if (tier >= 0) tier &= ~DFN_TRAIN_ACT_E4M3;
    if (act_e4m3 && tier != DFN_TIER_BF16) return fail(DFN_E_ARG, std::string(who) + ": DFN_TRAIN_ACT_E4M3 applies to DFN_TIER_BF16 only");
    if (with_loss && (!loss || !loss->img_head || !loss->img_com || !loss->d_rgb_head || !loss->d_rgb_com || !loss->losses ||
                      !loss->workspace))
        return fail(DFN_E_ARG, std::string(who) + "_loss: bad loss argument");
    if (!train_tier_ok(tier) || !frame || !packed_head || !packed_torso || !bias_head || !bias_torso || !rgb_head ||
        !rgb_com || !samples || !act_head || !masks_head || !act_torso || !masks_torso || (hier && (!z_all || !ranks)))
        return fail(DFN_E_ARG, std::string(who) + ": bad argument");
    const DfnFrame& F = *frame;
    if (!hier && ((F.n_coarse != 32 && F.n_coarse != 64 && F.n_coarse != 128) || F.n_fine != 0 || F.fields != 2))
        return fail(DFN_E_ARG, "dfn_train_fwd: the training step is coarse-only (64 samples), two fields (MAIN:855-899); "
                               "dfn_train_fwd_hier is the hierarchical variant");
    if (hier && (F.n_coarse != 64 || (F.n_fine != 64 && F.n_fine != 128) || F.fields != 2))
        return fail(DFN_E_ARG, "dfn_train_fwd_hier: 64 coarse + 64 or 128 fine samples, two fields");
    if (!bg_f32 && !bg_u8) return fail(DFN_E_ARG, std::string(who) + ": no background given");
    if (F.ray_count <= 0) return DFN_OK;
    const long NP = (long)F.ray_count * (F.n_coarse + F.n_fine);
    if (dfn_train_rows(1, 0) * NP >= (1L << 32)) return fail(DFN_E_ARG, std::string(who) + ": too many rays per call");
    ProgramInfo ph, pt;
    program_info(tier, FIELD_HEAD, &ph);
    program_info(tier, FIELD_TORSO, &pt);
    if (bias_torso != bias_head + ph.n_bias)
        return fail(DFN_E_ARG, std::string(who) + ": bias_torso must directly follow bias_head in memory");
    RenderArgs A;
    A.frame = F;
    A.out_u8 = 0;
    A.wblob[0] = (const char*)packed_head;
    A.wblob[1] = (const char*)packed_torso;
    A.nslab[0] = ph.n_slabs;
    A.nslab[1] = pt.n_slabs;
    A.bias = bias_head;
    A.bg_f32 = bg_f32;
    A.bg_u8 = bg_u8;
    A.pix_index = pix_index;
    A.rgb_head = rgb_head;
    A.rgb_com = rgb_com;
    A.w_head = A.w_com = nullptr;
    A.z_out = hier ? z_all : nullptr;
    A.samples_out = samples;
    A.ranks_out = hier ? ranks : nullptr;
    A.act_T[0] = act_head;
    A.act_T[1] = act_torso;
    A.masks[0] = masks_head;
    A.masks[1] = masks_torso;
    A.NP = NP;
    A.act_e4m3 = act_e4m3;
    A.loss = with_loss ? *loss : DfnTrainLoss{};
    A.clock_probe = nullptr;
    hipError_t err = launch_render(tier, A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "render_kernel(train)");
    return DFN_OK;
}

// per-workgroup partial sums [2][workgroups] + the ticket; a workgroup renders at least 4 rays in every tier
long dfn_train_loss_floats(int ray_count) { return ray_count <= 0 ? 4 : 2L * ((ray_count + 3) / 4) + 4; }
int dfn_train_fwd_loss(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                       const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                       const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                       uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, const DfnTrainLoss* loss, void* stream) {
    return train_fwd_impl(tier, frame, packed_head, packed_torso, bias_head, bias_torso, bg_f32, bg_u8, pix_index, rgb_head,
                          rgb_com, samples, act_head, masks_head, act_torso, masks_torso, nullptr, nullptr, false, stream,
                          loss, true);
}
int dfn_train_fwd_hier_loss(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                            const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                            const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                            uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, float* z_all, uint8_t* ranks,
                            const DfnTrainLoss* loss, void* stream) {
    return train_fwd_impl(tier, frame, packed_head, packed_torso, bias_head, bias_torso, bg_f32, bg_u8, pix_index, rgb_head,
                          rgb_com, samples, act_head, masks_head, act_torso, masks_torso, z_all, ranks, true, stream, loss,
                          true);
}

int dfn_train_fwd(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                  const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                  const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                  uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, void* stream) {
    return train_fwd_impl(tier, frame, packed_head, packed_torso, bias_head, bias_torso, bg_f32, bg_u8, pix_index, rgb_head,
                          rgb_com, samples, act_head, masks_head, act_torso, masks_torso, nullptr, nullptr, false, stream);
}

int dfn_train_fwd_hier(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                       const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                       const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                       uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, float* z_all, uint8_t* ranks,
                       void* stream) {
    return train_fwd_impl(tier, frame, packed_head, packed_torso, bias_head, bias_torso, bg_f32, bg_u8, pix_index, rgb_head,
                          rgb_com, samples, act_head, masks_head, act_torso, masks_torso, z_all, ranks, true, stream);
}

int dfn_sample_pixels(int H, int W, int n, int rect_num, const int32_t* rect, uint64_t seed, uint64_t counter,
                      int32_t* pix_index, int32_t* status, void* stream) {
    if (H <= 0 || W <= 0 || n <= 0 || rect_num < 0 || rect_num > n || !pix_index || (rect_num > 0 && !rect))
        return fail(DFN_E_ARG, "dfn_sample_pixels: bad argument");
    if ((long)H * W > 0x7fffffffL || n > SAMPLE_PIXELS_CANDIDATES / 2)
        return fail(DFN_E_ARG, "dfn_sample_pixels: at most 2^31 - 1 pixels and 4096 rays per call");
    hipError_t err = launch_sample_pixels(H, W, n, rect_num, rect, seed, counter, pix_index, status, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "sample_pixels_kernel");
    return DFN_OK;
}

int dfn_mse_loss_u8(const float* rgb_head, const float* rgb_com, const uint8_t* img_head, const uint8_t* img_com,
                    const int32_t* pix_index, int n, float* losses, float* d_rgb_head, float* d_rgb_com, void* stream) {
    if (!rgb_head || !rgb_com || !img_head || !img_com || !pix_index || !losses || !d_rgb_head || !d_rgb_com || n <= 0)
        return fail(DFN_E_ARG, "dfn_mse_loss_u8: bad argument");
    hipError_t err = launch_mse_loss(rgb_head, rgb_com, img_head, img_com, pix_index, n, losses, d_rgb_head, d_rgb_com,
                                     (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "mse_loss_kernel");
    return DFN_OK;
}

static int composite_bwd_impl(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                              const float* samples, const float* d_rgb_head, const float* d_rgb_com, float* dsamples,
                              float* zero_buf, long zero_floats, void* stream) {
    if (!frame || !samples || !d_rgb_head || !dsamples || (!bg_f32 && !bg_u8))
        return fail(DFN_E_ARG, "dfn_composite_bwd: bad argument");
    if ((frame->n_coarse != 32 && frame->n_coarse != 64 && frame->n_coarse != 128) || frame->n_fine != 0)
        return fail(DFN_E_ARG, "dfn_composite_bwd: 32, 64 or 128 coarse samples, no fine ones (dfn_composite_bwd_hier)");
    if (frame->ray_count <= 0) return DFN_OK;
    CompositeBwdArgs A;
    A.frame = *frame;
    A.pix_index = pix_index;
    A.bg_f32 = bg_f32;
    A.bg_u8 = bg_u8;
    A.samples = samples;
    A.d_rgb_head = d_rgb_head;
    A.d_rgb_com = d_rgb_com;
    A.dsamples = dsamples;
    A.z_all = nullptr;
    A.ranks = nullptr;
    A.zero_buf = zero_buf;
    A.zero_floats = zero_buf ? zero_floats : 0;
    hipError_t err = launch_composite_bwd(A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "composite_bwd_kernel");
    return DFN_OK;
}
int dfn_composite_bwd(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                      const float* samples, const float* d_rgb_head, const float* d_rgb_com, float* dsamples,
                      void* stream) {
    return composite_bwd_impl(frame, pix_index, bg_f32, bg_u8, samples, d_rgb_head, d_rgb_com, dsamples, nullptr, 0, stream);
}
int dfn_composite_bwd_z(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                        const float* samples, const float* d_rgb_head, const float* d_rgb_com, float* dsamples,
                        float* zero_buf, long zero_floats, void* stream) {
    if (zero_buf && (((unsigned long)zero_buf & 15) || zero_floats < 0))
        return fail(DFN_E_ARG, "dfn_composite_bwd_z: zero_buf must be 16-byte aligned");
    return composite_bwd_impl(frame, pix_index, bg_f32, bg_u8, samples, d_rgb_head, d_rgb_com, dsamples, zero_buf, zero_floats, stream);
}

static int composite_bwd_hier_impl(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                                   const float* samples, const float* z_all, const uint8_t* ranks, const float* d_rgb_head,
                                   const float* d_rgb_com, float* dsamples, float* zero_buf, long zero_floats, void* stream) {
    if (!frame || !samples || !z_all || !ranks || !d_rgb_head || !dsamples || (!bg_f32 && !bg_u8))
        return fail(DFN_E_ARG, "dfn_composite_bwd_hier: bad argument");
    if (frame->n_coarse != 64 || (frame->n_fine != 64 && frame->n_fine != 128))
        return fail(DFN_E_ARG, "dfn_composite_bwd_hier: 64 coarse + 64 or 128 fine samples");
    if (frame->ray_count <= 0) return DFN_OK;
    CompositeBwdArgs A;
    A.frame = *frame;
    A.pix_index = pix_index;
    A.bg_f32 = bg_f32;
    A.bg_u8 = bg_u8;
    A.samples = samples;
    A.d_rgb_head = d_rgb_head;
    A.d_rgb_com = d_rgb_com;
    A.dsamples = dsamples;
    A.z_all = z_all;
    A.ranks = ranks;
    A.zero_buf = zero_buf;
    A.zero_floats = zero_buf ? zero_floats : 0;
    hipError_t err = launch_composite_bwd_hier(A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "composite_bwd_hier_kernel");
    return DFN_OK;
}
int dfn_composite_bwd_hier(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                           const float* samples, const float* z_all, const uint8_t* ranks, const float* d_rgb_head,
                           const float* d_rgb_com, float* dsamples, void* stream) {
    return composite_bwd_hier_impl(frame, pix_index, bg_f32, bg_u8, samples, z_all, ranks, d_rgb_head, d_rgb_com, dsamples, nullptr, 0,
                                   stream);
}
int dfn_composite_bwd_hier_z(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                             const float* samples, const float* z_all, const uint8_t* ranks, const float* d_rgb_head,
                             const float* d_rgb_com, float* dsamples, float* zero_buf, long zero_floats, void* stream) {
    if (zero_buf && (((unsigned long)zero_buf & 15) || zero_floats < 0))
        return fail(DFN_E_ARG, "dfn_composite_bwd_hier_z: zero_buf must be 16-byte aligned");
    return composite_bwd_hier_impl(frame, pix_index, bg_f32, bg_u8, samples, z_all, ranks, d_rgb_head, d_rgb_com, dsamples, zero_buf,
                                   zero_floats, stream);
}

int dfn_mlp_bwd(int tier, int field, const void* packed_T, const float* samples, const float* dsamples,
                const uint32_t* masks, long NP, void* dy_T, void* stream) {
    if (!train_tier_ok(tier) || !train_field_ok(field) || !packed_T || !samples || !dsamples || !masks || !dy_T ||
        NP <= 0 || NP % 32)
        return fail(DFN_E_ARG, "dfn_mlp_bwd: bad argument");
    field = bwd_field(field);
    ProgramInfo pi;
    bwd_program_info(tier, field, &pi);
    MlpBwdArgs A;
    A.wblob_T = (const char*)packed_T;
    A.nslab = pi.n_slabs;
    A.samples = samples;
    A.dsamples = dsamples;
    A.masks = masks;
    A.dy_T = dy_T;
    A.NP = NP;
    hipError_t err = launch_mlp_bwd(tier, field, A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "mlp_bwd_kernel");
    return DFN_OK;
}

// inverse of the bias row table: dy_T row -> bias element (each row feeds at most one)
static int ensure_eof(WgradEntry& w, int tier, int field) {
    if ((long)w.bias_rows.size() != dfn_bias_floats(tier, field)) return fail(DFN_E_ARG, "internal: bias row table size");
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!w.eof_dev) {
        const int rows = (int)dfn_train_rows(field, 1);
        std::vector<int32_t> e_of(rows, -1);
        for (size_t e = 0; e < w.bias_rows.size(); ++e) {
            const int r = w.bias_rows[e];
            if (r < 0) continue;
            if (r >= rows || e_of[r] >= 0) return fail(DFN_E_ARG, "internal: bias row table is not one-to-one");
            e_of[r] = (int32_t)e;
        }
        hipError_t e = upload(&w.eof_dev, e_of.data(), e_of.size());
        if (e != hipSuccess) return hip_fail(e, "upload(bias rows)");
    }
    return DFN_OK;
}

// stages: 1 = the GEMMs (partial sums into the workspace), 2 = the reduction of the slices, 3 = both
static int weight_grad_impl(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP, float* workspace,
                            float* grad_flat, float* dbias, void* stream, const char* who, int stages = 3, int which = 3) {
    const bool gemm = stages & 1, red = stages & 2;
    if (which < 1 || which > 3 || (which != 3 && (tier != DFN_TIER_F32 || red)))
        return fail(DFN_E_ARG, std::string(who) + ": `which` selects the f32 tier's two GEMM launches (1: 256 x 256, 2: the others, 3: both)");
    if (act_format != DFN_ACT_E4M3 && act_format != DFN_ACT_E2M1)
        return fail(DFN_E_ARG, std::string(who) + ": act_format must be DFN_ACT_E4M3 or DFN_ACT_E2M1");
    if (!train_tier_ok(tier) || !train_field_ok(field) || (gemm && (!dy_T || !act_T)) || !workspace || (red && !grad_flat) ||
        NP <= 0 || NP % 32)
        return fail(DFN_E_ARG, std::string(who) + ": bad argument (NP must be a multiple of 32)");
    WgradEntry& w = wgrad_of(field);
    if (tier == DFN_TIER_F32 && !w.plan_error.empty()) return fail(DFN_E_ARG, std::string(who) + ": " + w.plan_error);
    hipStream_t st = (hipStream_t)stream;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (!w.ops_dev) {
            std::vector<WOp> ops(w.ops.size());
            for (size_t i = 0; i < ops.size(); ++i)
                ops[i] = WOp{w.ops[i].a_row, w.ops[i].M, w.ops[i].b_row, w.ops[i].N, w.ops[i].c_off, w.ops[i].bias_owner};
            int cus = 256;
            {
                int dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                    cus = prop.multiProcessorCount;
            }
            // some compute units are left to what runs NEXT to the GEMMs - the torso's dX chain next to the head's GEMMs, the
            // single-workgroup kernels of the conditioning networks' chains (their backward, Adam, the next step's encoders) next to
            // both: a launch of exactly one workgroup per compute unit, 144 KiB of LDS and 2 x 236 registers per SIMD each, leaves
            // them no slot until it ends.  Measured, whole step, interleaved on three boxes: the reference's step (131,072 points)
            // with 8 / 16 / 24 spare 0.99-1.02 ms, with 32 / 40 / 48 / 64: 0.95-0.97 (the GEMMs alone: 301 -> 307 us for both
            // fields with 32, 318 with 64); the hierarchical step (393,216 points: its GEMMs are three times as long, what runs next to
            // them is not) 2.756 with 8, 2.80 with 24, 2.788 with 40.  Hence two splits, by the size of the call: 32 / 8 spare.  (The
            // split only changes the ORDER of the sums; the 200-step bf16-vs-f32 curve of tests/test_gpu_train.py, two chaotic
            // trajectories, moves with it: worst step 1.95 % (32) ... 3.3 % (8) ... 5.2 % (40), final loss 0.014-1.0 %.)
            // DFN_WGRAD_SPARE_CUS[_HEAD | _TORSO]: developer override (both splits)
            const char* env = getenv(field == FIELD_TORSO ? "DFN_WGRAD_SPARE_CUS_TORSO" : "DFN_WGRAD_SPARE_CUS_HEAD");
            if (!env) env = getenv("DFN_WGRAD_SPARE_CUS");
            std::vector<WItem> items[2];
            std::vector<unsigned char> blk_n[2], bias_n[2];
            for (int c = 0; c < 2; ++c) {
                const int spare = env ? atoi(env) : (c == 0 ? 32 : 8);
                const int cus_c = (spare >= 0 && spare < cus / 2) ? cus - spare : cus;
                std::vector<int> n_of;
                wgrad_items(w.ops, field, cus_c, items[c], n_of);
                // slices per 256-element block of C (every GEMM's C region is a multiple of 1024 elements: one GEMM per block) and
                // per bias element (the GEMM that owns its dy_T row block)
                blk_n[c].assign((w.map.size() + 255) / 256, 1);
                bias_n[c].assign(w.bias_rows.size(), 1);
                for (size_t i = 0; i < w.ops.size(); ++i) {
                    const WOpHost& o = w.ops[i];
                    if (o.c_off % 256) return fail(DFN_E_ARG, "internal: a GEMM's C region is not block-aligned");
                    for (long bb = o.c_off / 256; bb < (o.c_off + (long)o.M * o.N + 255) / 256; ++bb) blk_n[c][bb] = (unsigned char)n_of[i];
                    if (o.bias_owner)
                        for (size_t e = 0; e < w.bias_rows.size(); ++e)
                            if (w.bias_rows[e] >= o.a_row && w.bias_rows[e] < o.a_row + o.M) bias_n[c][e] = (unsigned char)n_of[i];
                }
            }
            // upload into temporaries and publish every pointer only after ALL uploads succeeded: the guard above is keyed on
            // ops_dev, and a later call must never launch with a table that is still null
            WOp* d_ops = nullptr;
            int32_t *d_map = nullptr, *d_rows = nullptr;
            int* d_full = nullptr;
            WNItem* d_nitems = nullptr;
            WItem* d_items[2] = {nullptr, nullptr};
            unsigned char *d_blk[2] = {nullptr, nullptr}, *d_bn[2] = {nullptr, nullptr};
            hipError_t e = upload(&d_ops, ops.data(), ops.size());
            if (e == hipSuccess) e = upload(&d_map, w.map.data(), w.map.size());
            if (e == hipSuccess && !w.full_ops.empty()) e = upload(&d_full, w.full_ops.data(), w.full_ops.size());
            if (e == hipSuccess && !w.nitems.empty()) e = upload(&d_nitems, w.nitems.data(), w.nitems.size());
            if (e == hipSuccess && !w.rows_dev) e = upload(&d_rows, w.bias_rows.data(), w.bias_rows.size());
            for (int c = 0; c < 2; ++c) {
                if (e == hipSuccess) e = upload(&d_items[c], items[c].data(), items[c].size());
                if (e == hipSuccess) e = upload(&d_blk[c], blk_n[c].data(), blk_n[c].size());
                if (e == hipSuccess) e = upload(&d_bn[c], bias_n[c].data(), bias_n[c].size());
            }
            if (e != hipSuccess) {
                (void)hipFree(d_ops); (void)hipFree(d_map); (void)hipFree(d_full); (void)hipFree(d_nitems); (void)hipFree(d_rows);
                for (int c = 0; c < 2; ++c) { (void)hipFree(d_items[c]); (void)hipFree(d_blk[c]); (void)hipFree(d_bn[c]); }
                return hip_fail(e, "upload(wgrad plan)");
            }
            w.map_dev = d_map;
            w.full_ops_dev = d_full;
            w.nitems_dev = d_nitems;
            if (d_rows) w.rows_dev = d_rows;
            for (int c = 0; c < 2; ++c) {
                w.items_dev[c] = d_items[c];
                w.n_items[c] = (int)items[c].size();
                w.blk_n_dev[c] = d_blk[c];
                w.bias_n_dev[c] = d_bn[c];
            }
            w.ops_dev = d_ops;
        }
    }
    // Split-K without atomics: every (GEMM, slice of the points) writes its own slice of the partial arrays in the
    // workspace, the reduce kernels add the slices in index order -> bit-reproducible gradients.
    const long W = (long)w.map.size(), n_tiles = NP / 32;
    const int sc = NP > WGRAD_SMALL_NP ? 1 : 0;                    // which of the two splits (above)
    const int nb = (int)w.bias_rows.size();
    const int ks = tier == DFN_TIER_BF16 ? wgrad_ksplit_bf16(field) : WGRAD_KSPLIT;
    if (tier == DFN_TIER_BF16 && (n_tiles & 1))
        return fail(DFN_E_ARG, std::string(who) + ": the 16-bit tier contracts pairs of 32-point tiles: NP must be a multiple of 64");
    const long units = tier == DFN_TIER_BF16 ? n_tiles / 2 : n_tiles;      // what a slice is made of: tile pairs / tiles
    const long per = (units + ks - 1) / ks;
    const int valid = (int)((units + per - 1) / per);             // slices that hold points (the others write nothing)
    float* c_parts = workspace;
    float* b_parts = workspace + (long)WS_KSPLIT_MAX * W;
    hipError_t err = hipSuccess;
    if (dbias) {
        const int rc = ensure_eof(w, tier, field);
        if (rc != DFN_OK) return rc;
    }
    // The row sums (bias gradients) ride along in the GEMMs, partial per slice like the products: bf16 tier two cheap MFMAs
    // per step; f32 tier on the vector ALU from the operand registers (round 5: a streaming row-sum kernel of its own read
    // dy_T a second time, 0.27 ms per field).
    const bool fuse = dbias && tier == DFN_TIER_BF16;
    const bool ride = dbias && tier == DFN_TIER_F32;
    if (!gemm) {
    } else if (tier == DFN_TIER_BF16)
        err = launch_wgrad_bf16(field, act_format == DFN_ACT_E2M1, w.ops_dev, w.items_dev[sc], w.n_items[sc], dy_T, act_T, NP, c_parts, W,
                                fuse ? w.eof_dev : nullptr, fuse ? b_parts : nullptr, nb, st);
    else
        err = launch_wgrad(tier, field, w.ops_dev, w.full_ops_dev, (which & 1) ? (int)w.full_ops.size() : 0, w.nitems_dev,
                           (which & 2) ? (int)w.nitems.size() : 0, dy_T, act_T, NP, ks, c_parts, W, ride ? w.eof_dev : nullptr,
                           ride ? b_parts : nullptr, nb, st);
    if (err != hipSuccess) return hip_fail(err, "wgrad_kernel");
    if (!red) return DFN_OK;
    if (ride) {
        err = launch_reduce_bias(w.rows_dev, b_parts, nb, valid, dbias, st);
        if (err != hipSuccess) return hip_fail(err, "reduce_bias_kernel");
    }
    if (fuse) {
        err = launch_reduce_both(w.map_dev, c_parts, W, W, valid, grad_flat, w.rows_dev, b_parts, nb, dbias, w.blk_n_dev[sc],
                                 w.bias_n_dev[sc], units, st);
        if (err != hipSuccess) return hip_fail(err, "reduce_both_kernel");
        return DFN_OK;
    }
    if (tier == DFN_TIER_BF16) {          // (weights only) the per-GEMM slice counts of the balanced split
        err = launch_reduce_both(w.map_dev, c_parts, W, W, valid, grad_flat, nullptr, nullptr, 0, nullptr, w.blk_n_dev[sc], nullptr,
                                 units, st);
        if (err != hipSuccess) return hip_fail(err, "reduce_both_kernel");
        return DFN_OK;
    }
    err = launch_reduce_scatter(w.map_dev, c_parts, W, W, valid, grad_flat, st);
    if (err != hipSuccess) return hip_fail(err, "reduce_scatter_kernel");
    return DFN_OK;
}

long dfn_wgrad_plan(int field, int what, int32_t* out, long capacity) {
    if (!train_field_ok(field) || what < 0 || what > 2) return fail(DFN_E_ARG, "dfn_wgrad_plan: bad field / selector");
    WgradEntry& w = wgrad_of(field);
    std::vector<int32_t> ops;
    if (what == 0)
        for (const WOpHost& o : w.ops) ops.insert(ops.end(), {o.a_row, o.M, o.b_row, o.N, o.c_off, o.bias_owner});
    const std::vector<int32_t>& v = what == 0 ? ops : what == 1 ? w.map : w.bias_rows;
    if (out) {
        if (capacity < (long)v.size()) return fail(DFN_E_SIZE, "dfn_wgrad_plan: capacity too small");
        std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
    }
    return (long)v.size();
}

// (the two entry points without a format argument consume what the FUSED step records: dfn_train_fwd / dfn_train_fwd_hier)
int dfn_weight_grad(int tier, int field, const void* dy_T, const void* act_T, long NP, float* workspace,
                    float* grad_flat, void* stream) {
    return weight_grad_impl(tier, field, DFN_ACT_E2M1, dy_T, act_T, NP, workspace, grad_flat, nullptr, stream,
                            "dfn_weight_grad");
}

int dfn_weight_bias_grad(int tier, int field, const void* dy_T, const void* act_T, long NP, float* workspace,
                         float* grad_flat, float* dbias, void* stream) {
    if (!dbias) return fail(DFN_E_ARG, "dfn_weight_bias_grad: dbias is NULL");
    return weight_grad_impl(tier, field, DFN_ACT_E2M1, dy_T, act_T, NP, workspace, grad_flat, dbias, stream,
                            "dfn_weight_bias_grad");
}

int dfn_weight_bias_grad_fmt(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP, float* workspace,
                             float* grad_flat, float* dbias, void* stream) {
    if (!dbias) return fail(DFN_E_ARG, "dfn_weight_bias_grad_fmt: dbias is NULL");
    return weight_grad_impl(tier, field, act_format, dy_T, act_T, NP, workspace, grad_flat, dbias, stream, "dfn_weight_bias_grad_fmt");
}

int dfn_weight_bias_grad_partials(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP,
                                  float* workspace, float* dbias, void* stream) {
    if (!dbias) return fail(DFN_E_ARG, "dfn_weight_bias_grad_partials: dbias is NULL");
    return weight_grad_impl(tier, field, act_format, dy_T, act_T, NP, workspace, nullptr, dbias, stream,
                            "dfn_weight_bias_grad_partials", 1);
}
int dfn_weight_bias_grad_partials_part(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP,
                                       float* workspace, float* dbias, int which, void* stream) {
    if (!dbias) return fail(DFN_E_ARG, "dfn_weight_bias_grad_partials_part: dbias is NULL");
    return weight_grad_impl(tier, field, act_format, dy_T, act_T, NP, workspace, nullptr, dbias, stream,
                            "dfn_weight_bias_grad_partials_part", 1, which);
}
int dfn_weight_bias_grad_reduce(int tier, int field, long NP, float* workspace, float* grad_flat, float* dbias, void* stream) {
    if (!dbias) return fail(DFN_E_ARG, "dfn_weight_bias_grad_reduce: dbias is NULL");
    return weight_grad_impl(tier, field, DFN_ACT_E4M3, nullptr, nullptr, NP, workspace, grad_flat, dbias, stream,
                            "dfn_weight_bias_grad_reduce", 2);
}

int dfn_bias_grad(int tier, int field, const void* dy_T, long NP, float* workspace, float* dbias, void* stream) {
    if (!train_tier_ok(tier) || !train_field_ok(field) || !dy_T || !workspace || !dbias || NP <= 0)
        return fail(DFN_E_ARG, "dfn_bias_grad: bad argument");
    WgradEntry& w = wgrad_of(field);
    {
        const int rc = ensure_eof(w, tier, field);
        if (rc != DFN_OK) return rc;
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (!w.rows_dev) {
            hipError_t e = upload(&w.rows_dev, w.bias_rows.data(), w.bias_rows.size());
            if (e != hipSuccess) return hip_fail(e, "upload(bias rows)");
        }
    }
    hipError_t err = launch_bias_grad(tier, field, w.eof_dev, w.rows_dev, (int)w.bias_rows.size(), dy_T, NP, workspace, dbias,
                                      (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "bias_grad_kernel");
    return DFN_OK;
}

int dfn_zero_async(void* p, long bytes, void* stream) {
    if (!p || bytes < 0) return fail(DFN_E_ARG, "dfn_zero_async: bad argument");
    if (bytes == 0) return DFN_OK;
    // dword-aligned buffers (the gradient buffers): ONE launch (hipMemsetAsync of a 3.8-MB buffer is two fill kernels, 12 us
    // between the compositing backward and the dX chain of a 1.1-ms step)
    if (((unsigned long)p & 15) == 0 && (bytes & 3) == 0) {
        hipError_t e = launch_zero_words((unsigned*)p, bytes / 4, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "zero_words_kernel");
        return DFN_OK;
    }
    hipError_t err = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "hipMemsetAsync");
    return DFN_OK;
}

int dfn_signal_grad(int tier, int field, const float* params, const void* dy_T, long NP, float* workspace, float* d_signal,
                    void* stream) {
    if (!train_tier_ok(tier) || (field != 0 && field != 1) || !params || !dy_T || !workspace || !d_signal || NP <= 0 ||
        NP % 32)
        return fail(DFN_E_ARG, "dfn_signal_grad: bad argument (NP must be a multiple of 32)");
    WgradEntry& w = wgrad_of(field);
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (!w.sig_rows_dev) {
            int elems[512];
            const int n = sig_term_elements(field, elems);
            std::vector<int32_t> rows(n), el(elems, elems + n);
            for (int i = 0; i < n; ++i) {
                rows[i] = w.bias_rows[elems[i]];
                if (rows[i] < 0 || n % 64) return fail(DFN_E_ARG, "internal: a signal-term bias element without a gradient row");
                // sig_rows8_kernel: 32 consecutive entries = the rows of one aligned 32-row block
                if ((rows[i] >> 5) != (rows[i & ~31] >> 5)) return fail(DFN_E_ARG, "internal: signal rows not in whole 32-row blocks");
            }
            int32_t *d_rows = nullptr, *d_el = nullptr;      // published together, after both uploads succeeded
            hipError_t e = upload(&d_rows, rows.data(), rows.size());
            if (e == hipSuccess) e = upload(&d_el, el.data(), el.size());
            if (e != hipSuccess) {
                (void)hipFree(d_rows); (void)hipFree(d_el);
                return hip_fail(e, "upload(signal rows)");
            }
            w.n_sig = n;
            w.sig_elems_dev = d_el;
            w.sig_rows_dev = d_rows;
        }
    }
    float* parts = workspace;
    float* dbias = workspace + (long)SIG_ROW_SLICES * 512;      // behind the partial sums' whole area (launch_signal_rows sizes the split)
    hipError_t err = launch_signal_rows(tier, field, w.sig_rows_dev, w.sig_elems_dev, w.n_sig, dy_T, NP, parts, dbias,
                                        (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "sig_rows_kernel");
    err = launch_fold_bwd_sig(field, params, dbias, d_signal, true, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "fold_bwd_sig_kernel");
    return DFN_OK;
}

static int encode_signal_impl(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                              const float* exps, int n_total, const int32_t* frame_ids, int n_frames, int smo_size, float* out,
                              float* keep, void* stream) {
    if (!aud_params || !exp_params || !auds || !exps || !frame_ids || !out || n_total <= 0 || n_frames < 0)
        return fail(DFN_E_ARG, "dfn_encode_signal: bad argument");
    if (keep && n_frames != 1) return fail(DFN_E_ARG, "dfn_encode_signal_keep: one frame (the training step's)");
    if (smo_size < 0 || smo_size > 8 || (smo_size & 1)) return fail(DFN_E_ARG, "dfn_encode_signal: smo_size must be 0, 2, 4, 6 or 8");
    if (smo_size > 0 && !att_params) return fail(DFN_E_ARG, "dfn_encode_signal: attention parameters missing");
    if (n_frames == 0) return DFN_OK;
    hipError_t err = launch_encode_signal(aud_params, exp_params, att_params, auds, exps, n_total, frame_ids, n_frames,
                                          smo_size, out, keep, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "encode_signal_kernel");
    return DFN_OK;
}
int dfn_encode_signal(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                      const float* exps, int n_total, const int32_t* frame_ids, int n_frames, int smo_size, float* out,
                      void* stream) {
    return encode_signal_impl(aud_params, exp_params, att_params, auds, exps, n_total, frame_ids, n_frames, smo_size, out, nullptr,
                              stream);
}
int dfn_encode_signal_keep(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                           const float* exps, int n_total, const int32_t* frame_id, int smo_size, float* out, float* keep,
                           void* stream) {
    if (!keep) return fail(DFN_E_ARG, "dfn_encode_signal_keep: keep is NULL");
    return encode_signal_impl(aud_params, exp_params, att_params, auds, exps, n_total, frame_id, 1, smo_size, out, keep, stream);
}
long dfn_encode_signal_keep_floats(void) { return SIG_KEEP_FLOATS; }

int dfn_encode_signal_torso(const float* att_params, const float* poses, int pose_stride, int n_total,
                            const int32_t* frame_ids, int n_frames, int smo_size, float* out, void* stream) {
    if (!poses || !frame_ids || !out || n_total <= 0 || n_frames < 0 || (pose_stride != 12 && pose_stride != 16))
        return fail(DFN_E_ARG, "dfn_encode_signal_torso: bad argument");
    if (smo_size < 0 || smo_size > 8 || (smo_size & 1))
        return fail(DFN_E_ARG, "dfn_encode_signal_torso: smo_size must be 0, 2, 4, 6 or 8");
    if (smo_size > 0 && !att_params) return fail(DFN_E_ARG, "dfn_encode_signal_torso: attention parameters missing");
    if (n_frames == 0) return DFN_OK;
    hipError_t err = launch_encode_signal_torso(att_params, poses, pose_stride, n_total, frame_ids, n_frames, smo_size, out,
                                                (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "encode_signal_torso_kernel");
    return DFN_OK;
}

static int encode_signal_bwd_impl(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                                  const float* exps, int n_total, int frame, int smo_size, const float* d_out, float* g_aud,
                                  float* g_exp, float* g_att, bool set, void* stream, const float* kept = nullptr) {
    if (!aud_params || !exp_params || !auds || !exps || !d_out || !g_aud || !g_exp || n_total <= 0)
        return fail(DFN_E_ARG, "dfn_encode_signal_bwd: bad argument");
    if (smo_size < 0 || smo_size > 8 || (smo_size & 1)) return fail(DFN_E_ARG, "dfn_encode_signal_bwd: smo_size must be 0, 2, 4, 6 or 8");
    if (smo_size > 0 && (!att_params || !g_att)) return fail(DFN_E_ARG, "dfn_encode_signal_bwd: attention buffers missing");
    hipError_t err = launch_encode_signal_bwd(aud_params, exp_params, att_params, auds, exps, n_total, frame, smo_size,
                                              d_out, g_aud, g_exp, g_att, set, kept, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "encode_signal_bwd_kernel");
    return DFN_OK;
}
int dfn_encode_signal_bwd(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                          const float* exps, int n_total, int frame, int smo_size, const float* d_out, float* g_aud,
                          float* g_exp, float* g_att, void* stream) {
    return encode_signal_bwd_impl(aud_params, exp_params, att_params, auds, exps, n_total, frame, smo_size, d_out, g_aud, g_exp,
                                  g_att, false, stream);
}
int dfn_encode_signal_bwd_kept(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                               const float* exps, int n_total, int frame, int smo_size, const float* d_out, const float* kept,
                               float* g_aud, float* g_exp, float* g_att, void* stream) {
    if (!kept) return fail(DFN_E_ARG, "dfn_encode_signal_bwd_kept: kept is NULL");
    return encode_signal_bwd_impl(aud_params, exp_params, att_params, auds, exps, n_total, frame, smo_size, d_out, g_aud, g_exp,
                                  g_att, false, stream, kept);
}
int dfn_encode_signal_bwd_set(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                              const float* exps, int n_total, int frame, int smo_size, const float* d_out, float* g_aud,
                              float* g_exp, float* g_att, void* stream) {
    return encode_signal_bwd_impl(aud_params, exp_params, att_params, auds, exps, n_total, frame, smo_size, d_out, g_aud, g_exp,
                                  g_att, true, stream);
}

static int encode_signal_torso_bwd_impl(const float* att_params, const float* poses, int pose_stride, int n_total, int frame,
                                        int smo_size, const float* d_out, float* g_att, bool set, void* stream) {
    if (!poses || !d_out || n_total <= 0 || (pose_stride != 12 && pose_stride != 16))
        return fail(DFN_E_ARG, "dfn_encode_signal_torso_bwd: bad argument");
    if (smo_size < 0 || smo_size > 8 || (smo_size & 1))
        return fail(DFN_E_ARG, "dfn_encode_signal_torso_bwd: smo_size must be 0, 2, 4, 6 or 8");
    if (smo_size == 0) return DFN_OK;          // no parameter takes part before --nosmo_iters
    if (!att_params || !g_att) return fail(DFN_E_ARG, "dfn_encode_signal_torso_bwd: attention buffers missing");
    hipError_t err = launch_encode_signal_torso_bwd(att_params, poses, pose_stride, n_total, frame, smo_size, d_out, g_att, set,
                                                    (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "encode_signal_torso_bwd_kernel");
    return DFN_OK;
}
int dfn_encode_signal_torso_bwd(const float* att_params, const float* poses, int pose_stride, int n_total, int frame,
                                int smo_size, const float* d_out, float* g_att, void* stream) {
    return encode_signal_torso_bwd_impl(att_params, poses, pose_stride, n_total, frame, smo_size, d_out, g_att, false, stream);
}
int dfn_encode_signal_torso_bwd_set(const float* att_params, const float* poses, int pose_stride, int n_total, int frame,
                                    int smo_size, const float* d_out, float* g_att, void* stream) {
    return encode_signal_torso_bwd_impl(att_params, poses, pose_stride, n_total, frame, smo_size, d_out, g_att, true, stream);
}

int dfn_decoder_fwd(int tier, int field, const void* packed, const float* bias, const float* points,
                    const float* dirs, long n, float* feat, float* sigma, void* stream) {
    if (!tier_ok(tier) || !field_ok(field) || !packed || !bias || !points || !dirs || !feat || !sigma)
        return fail(DFN_E_ARG, "dfn_decoder_fwd: bad argument");
    if (n <= 0) return DFN_OK;
    ProgramInfo pi;
    program_info(tier, prog_field(field), &pi);
    DecoderArgs A;
    A.wblob = (const char*)packed;
    A.nslab = pi.n_slabs;
    A.field = prog_field(field);
    A.bias = bias;
    A.n_bias = pi.n_bias;
    A.points = points;
    A.dirs = dirs;
    A.n_points = n;
    A.feat = feat;
    A.sigma = sigma;
    A.samples = nullptr;
    A.act_T = nullptr;
    A.masks = nullptr;
    hipError_t err = launch_decoder(tier, A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "decoder_kernel");
    return DFN_OK;
}

int dfn_decoder_train_fwd(int tier, int field, const void* packed, const float* bias, const float* points,
                          const float* dirs, long n, float* feat, float* sigma, float* samples, void* act_T,
                          uint32_t* masks, void* stream) {
    if (!train_tier_ok(tier) || !train_field_ok(field) || !packed || !bias || !points ||
        !dirs || !feat || !sigma || !samples || !act_T || !masks)
        return fail(DFN_E_ARG, "dfn_decoder_train_fwd: bad argument (tiers f32 / bf16, fields head / torso / listener)");
    if (n <= 0) return DFN_OK;
    const long NP = (n + 31) / 32 * 32;
    if (dfn_train_rows(1, 0) * NP >= (1L << 32)) return fail(DFN_E_ARG, "dfn_decoder_train_fwd: too many points per call");
    ProgramInfo pi;
    program_info(tier, prog_field(field), &pi);
    DecoderArgs A;
    A.wblob = (const char*)packed;
    A.nslab = pi.n_slabs;
    A.field = prog_field(field);
    A.bias = bias;
    A.n_bias = pi.n_bias;
    A.points = points;
    A.dirs = dirs;
    A.n_points = n;
    A.feat = feat;
    A.sigma = sigma;
    A.samples = samples;
    A.act_T = act_T;
    A.masks = masks;
    hipError_t err = launch_decoder(tier, A, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "decoder_kernel(train)");
    return DFN_OK;
}

int dfn_get_rays(int H, int W, float focal, float cx, float cy, const float* c2w_host, float* rays_o,
                 float* rays_d, void* stream) {
    return dfn_get_rays_strided(H, W, 1, focal, cx, cy, c2w_host, rays_o, rays_d, stream);
}

int dfn_get_rays_strided(int H, int W, int stride, float focal, float cx, float cy, const float* c2w_host, float* rays_o,
                         float* rays_d, void* stream) {
    if (H <= 0 || W <= 0 || stride <= 0 || H / stride <= 0 || W / stride <= 0 || !c2w_host || !rays_o || !rays_d)
        return fail(DFN_E_ARG, "dfn_get_rays: bad argument");
    hipError_t err = launch_get_rays(H, W, stride, focal, cx, cy, c2w_host, rays_o, rays_d, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "get_rays_kernel");
    return DFN_OK;
}

int dfn_ndc_rays(int H, int W, float focal, float z_near, const float* rays_o, const float* rays_d, long n,
                 float* out_o, float* out_d, void* stream) {
    if (!rays_o || !rays_d || !out_o || !out_d || n < 0) return fail(DFN_E_ARG, "dfn_ndc_rays: bad argument");
    if (n == 0) return DFN_OK;
    hipError_t err = launch_ndc_rays(H, W, focal, z_near, rays_o, rays_d, n, out_o, out_d, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "ndc_rays_kernel");
    return DFN_OK;
}

int dfn_sample_pdf(const float* bins, const float* weights, long R, int nb, int ns, const float* u, float* samples,
                   void* stream) {
    if (!bins || !weights || !samples || R < 0 || ns <= 0) return fail(DFN_E_ARG, "dfn_sample_pdf: bad argument");
    if (nb < 2 || nb > 256) return fail(DFN_E_ARG, "dfn_sample_pdf: need 2 <= nb <= 256");
    if (R == 0) return DFN_OK;
    hipError_t err = launch_sample_pdf(bins, weights, R, nb, ns, u, samples, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "sample_pdf_kernel");
    return DFN_OK;
}

int dfn_composite(const float* sigma, const float* feat, int K, long N, float* sigma_sum, float* feat_w,
                  void* stream) {
    if (!sigma || !feat || !sigma_sum || !feat_w || K < 1 || N < 0) return fail(DFN_E_ARG, "dfn_composite: bad argument");
    if (N == 0) return DFN_OK;
    hipError_t err = launch_composite(sigma, feat, K, N, sigma_sum, feat_w, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "composite_kernel");
    return DFN_OK;
}

int dfn_volume_weights(const float* z, const float* ray, const float* sigma, long R, int S, float last_dist,
                       float* weights, void* stream) {
    if (!z || !ray || !sigma || !weights || R < 0) return fail(DFN_E_ARG, "dfn_volume_weights: bad argument");
    if (S < 1 || S > 1024) return fail(DFN_E_ARG, "dfn_volume_weights: need 1 <= S <= 1024");
    if (R == 0) return DFN_OK;
    hipError_t err = launch_volume_weights(z, ray, sigma, R, S, last_dist, weights, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "volume_weights_kernel");
    return DFN_OK;
}

int dfn_composite_grad(const float* sigma, const float* feat, int K, long N, const float* d_sigma_sum, const float* d_feat_w,
                       float* d_sigma, float* d_feat, void* stream) {
    if (!sigma || !feat || !d_sigma || !d_feat || K < 1 || N < 0) return fail(DFN_E_ARG, "dfn_composite_grad: bad argument");
    if (N == 0) return DFN_OK;
    hipError_t err = launch_composite_grad(sigma, feat, K, N, d_sigma_sum, d_feat_w, d_sigma, d_feat, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "composite_grad_kernel");
    return DFN_OK;
}

int dfn_volume_weights_grad(const float* z, const float* ray, const float* sigma, long R, int S, float last_dist,
                            const float* d_weights, float* d_sigma, void* stream) {
    if (!z || !ray || !sigma || !d_weights || !d_sigma || R < 0)
        return fail(DFN_E_ARG, "dfn_volume_weights_grad: bad argument");
    if (S < 1 || S > 1024) return fail(DFN_E_ARG, "dfn_volume_weights_grad: need 1 <= S <= 1024");
    if (R == 0) return DFN_OK;
    hipError_t err = launch_volume_weights_grad(z, ray, sigma, R, S, last_dist, d_weights, d_sigma, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "volume_weights_grad_kernel");
    return DFN_OK;
}

int dfn_to8b(const float* x, long n, uint8_t* out, void* stream) {
    if (!x || !out || n < 0) return fail(DFN_E_ARG, "dfn_to8b: bad argument");
    if (n == 0) return DFN_OK;
    hipError_t err = launch_to8b(x, n, out, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "to8b_kernel");
    return DFN_OK;
}

int dfn_debug_mfma_chain(int tier, int lds_reads_per_2, int valu_per_2, const void* fragments, const void* operands_b, int iters,
                         int blocks, float* out, uint64_t* clock, void* stream) {
    if ((tier != DFN_TIER_BF16 && tier != DFN_TIER_F16) || !fragments || !operands_b || !out || !clock || iters <= 0 || blocks <= 0)
        return fail(DFN_E_ARG, "dfn_debug_mfma_chain: bad argument");
    hipError_t err = launch_mfma_chain(tier == DFN_TIER_F16, lds_reads_per_2, valu_per_2, fragments, operands_b, iters, blocks, out,
                                       (unsigned long long*)clock, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "mfma_chain_kernel (variants: (0, 0) and (2, 4))");
    return DFN_OK;
}

int dfn_debug_clock_probe(uint64_t* probe) {
    g_clock_probe = (unsigned long long*)probe;
    return DFN_OK;
}

int dfn_debug_mfma_layout(float* out, void* stream) {
    if (!out) return fail(DFN_E_ARG, "dfn_debug_mfma_layout: bad argument");
    hipError_t err = launch_mfma_probe(out, (hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "mfma_probe_kernel");
    return DFN_OK;
}

}  // extern "C"
