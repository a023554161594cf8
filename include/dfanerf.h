/* dfanerf.h - C ABI of the MI355X-native DFA-NeRF rendering path (libdfanerf.so).
 *
 * The reference (ShunyuYao/DFA-NeRF) has no plugin / operator / FFI interface on this path: the
 * renderer is plain Python calling ATen.  This header is therefore the interface a
 * maintainer would bind INSTEAD of the Python bodies listed below; every entry point names the
 * reference code it replaces (paths relative to NeRFs/DFANeRF/ of the reference tree):
 *
 *   MAIN = run_nerf_com_trainExpLater.py   HELP = run_nerf_helpers.py   DEC = decoder.py
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless a parameter is documented as host memory;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising the device; the library never allocates memory the caller can see;
 *   - return value 0 = ok, negative = error (DFN_E_*); dfn_last_error() gives a thread-local message;
 *   - tensors are row-major fp32 unless stated; ray index = y*W + x (MAIN:635-636).
 *   - only the architecture built by scripts/test_obama.sh / train_obama.sh is supported:
 *     Decoder(hidden 256, z_dim 256, 8 blocks, skip at 4, dim_signal 96, dim_et_embed 42,
 *     10/4 PE octaves, deformation field on, expression off)  -> 955,242 decoder parameters.
 */
#ifndef DFANERF_H_
#define DFANERF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFN_OK 0
#define DFN_E_ARG (-1)      /* bad argument / unsupported configuration */
#define DFN_E_HIP (-2)      /* HIP runtime error (message has hipGetErrorString) */
#define DFN_E_SIZE (-3)     /* buffer too small */

#define DFN_TIER_F32 0      /* v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered accumulate */
#define DFN_TIER_BF16 1     /* v_mfma_f32_32x32x16_bf16: bf16 operands, f32 accumulate */
#define DFN_TIER_F16 2      /* v_mfma_f32_32x32x16_f16: f16 operands (10 mantissa bits), f32 accumulate; same rate as bf16.
                             * Inference entry points only (pack / fold / render / decoder): the training entry points
                             * return DFN_E_ARG for it (gradients underflow f16's exponent range). */

#define DFN_FIELD_HEAD 0        /* DEC:303-305  fc_in / fc_p_skips       */
#define DFN_FIELD_TORSO 1       /* DEC:297-299, 308-309, 324-325 deform_net + fc_in_torso / fc_p_skips_torso */
#define DFN_FIELD_LISTENER 2    /* DEC:306-307, 322-323 fc_in_listener / fc_p_skips_listener (signal None) */

#define DFN_N_DECODER_PARAMS 955242

const char* dfn_last_error(void);
/* library build info: "dfanerf <version> gfx950".  ABI notes - 0.2 (round 6): + dfn_wgrad_plan, dfn_get_rays_strided, dfn_weight_bias_grad_partials_part; DFN_FIELD_LISTENER accepted by the
 * training entry points; DfnFrame.n_coarse 32 / 64 / 128.  Since round 5 (still "0.1" then): dfn_weight_bias_grad_partials only fills
 * the workspace's per-slice partials in EVERY tier - dbias is written by dfn_weight_bias_grad_reduce (a caller of _partials alone gets
 * no bias gradient; tests/test_gpu_wgrad.py holds the pair to the one-call form bit for bit). */
const char* dfn_version(void);

/* ---- geometry of one frame (host struct, passed by value inside the calls below) -------------------- */
typedef struct DfnFrame {
    float pose[12];        /* head camera-to-world, rows of the 3x4 (MAIN:634 poses[img_i,:3,:4]) */
    float pose_body[12];   /* torso/body camera-to-world (MAIN:645 pose_body[:3,:4]) */
    int H, W;
    float focal, cx, cy;   /* LOAD:35-36 */
    float z_near, z_far;   /* MAIN:612-618 */
    float last_dist;       /* --last_dist, MAIN:171 */
    int ray_begin;         /* first ray (y*W+x) rendered when pix_index == NULL */
    int ray_count;         /* number of rays rendered by this call */
    int n_coarse;          /* --N_samples (MAIN:612-619): 32, 64 or 128; 64 when n_fine > 0 */
    int n_fine;            /* 0 (live reference renderer) or 64/128 */
    int fields;            /* 1 = head only, 2 = head + torso composite (MAIN:681-709) */
    int concate_bg;        /* --concate_bg, MAIN:669-671, 678-679, 692-694 */
} DfnFrame;

/* ---- weights ---------------------------------------------------------------------------------------
 * `params` is decoder.state_dict() flattened in registration order (DEC:207-251), 955,242 floats.
 * dfn_packed_bytes gives the size of the kernel-ready stream for one (tier, field); dfn_pack_weights
 * fills it (a gather through a cached plan: run it again after every optimizer step). */
long dfn_packed_bytes(int tier, int field);
int dfn_pack_weights(int tier, int field, const float* params, void* packed, void* stream);
/* Host-side view of the same plan, for tests: plan[i] = flat index into `params` of packed element i,
 * or -1 for a structural zero.  Returns the element count (call with plan == NULL to query). */
long dfn_pack_plan(int tier, int field, int32_t* plan_host, long capacity);

/* Per-frame constants folded into bias vectors (replaces the per-point `signal.expand` + cat at
 * DEC:293-295, and fc_z / fc_z_skips / fc_z_view at DEC:311, 318, 332 which are per-frame constants).
 *   sig_head  [96]  encode_signal(...)[0]  (MAIN:28-75); NULL for the listener field
 *   sig_torso [42]  encode_signal_torso(...) (MAIN:78-111)
 *   z_shape, z_app [256] the latent rows used by this field (MAIN:664-665 / 673-674)
 * Output `bias`: dfn_bias_floats(tier, field) floats. */
long dfn_bias_floats(int tier, int field);
int dfn_fold_bias(int tier, int field, const float* params, const float* signal, const float* z_shape,
                  const float* z_app, float* bias, void* stream);

/* ---- per-frame conditioning signals (forward; training differentiates the torch twins) ---------------------------
 * encode_signal, MAIN:28-75 (object 0): AudioNet_W2L (HELP:165-178) and ExpressionEnc (HELP:182-193) on the window
 * [i - smo/2, i + smo/2) of audio [n_total,512] / expression [n_total,64] features, rows outside the sequence are
 * ZERO INPUT rows (MAIN:36-57), then AudioAttNet(96, smo) (HELP:210-240) -> out [n_frames,96].  smo_size == 0 is the
 * branch before --nosmo_iters: cat(AudNet(aud[i]), ExpNet(exp[i])).  *_params: each network's state_dict()
 * flattened in registration order (172,480 / 3,136 / 5,169 floats for smo 4).  frame_ids: device int32. */
int dfn_encode_signal(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                      const float* exps, int n_total, const int32_t* frame_ids, int n_frames, int smo_size, float* out,
                      void* stream);
/* encode_signal_torso, MAIN:78-111: rot_to_euler + translation (MAIN:182-204) of the window of poses
 * ([n_total] matrices of pose_stride = 16 (4x4) or 12 (3x4) floats), zero rows outside, get_embedder(3,0)
 * (HELP:21-70) on each half -> 42, AudioAttNet(42, smo) -> out [n_frames,42]; smo_size == 0: the single frame. */
int dfn_encode_signal_torso(const float* att_params, const float* poses, int pose_stride, int n_total,
                            const int32_t* frame_ids, int n_frames, int smo_size, float* out, void* stream);

/* Backward of the two encoders for ONE frame (upstream trains one frame per step, MAIN:737-941; there it is torch
 * autograd): d_out [96] / [42] -> parameter gradients, ADDED into g_* (flat, laid out like the *_params).  The inputs
 * (features, poses) are data and get no gradient.  smo_size == 0: g_att untouched (the attention net is unused). */
int dfn_encode_signal_bwd(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                          const float* exps, int n_total, int frame, int smo_size, const float* d_out, float* g_aud,
                          float* g_exp, float* g_att, void* stream);
int dfn_encode_signal_torso_bwd(const float* att_params, const float* poses, int pose_stride, int n_total, int frame,
                                int smo_size, const float* d_out, float* g_att, void* stream);
/* The training step's pair: dfn_encode_signal_keep = dfn_encode_signal for ONE frame that also leaves the activations the backward
 * needs in `keep` (dfn_encode_signal_keep_floats() floats), dfn_encode_signal_bwd_kept = dfn_encode_signal_bwd reading them
 * instead of running AudioNet's forward a second time, and computing AudioNet's two big weight gradients in a second,
 * many-workgroup launch (same arithmetic: same gradients bit for bit; frame and smo_size as in the keep call; the tail of `kept`
 * is the backward's scratch: the buffer is written by the call although it is declared const). */
int dfn_encode_signal_keep(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                           const float* exps, int n_total, const int32_t* frame_id, int smo_size, float* out, float* keep,
                           void* stream);
long dfn_encode_signal_keep_floats(void);
int dfn_encode_signal_bwd_kept(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                               const float* exps, int n_total, int frame, int smo_size, const float* d_out, const float* kept,
                               float* g_aud, float* g_exp, float* g_att, void* stream);
/* The same, WRITING the gradients instead of adding them (every element of g_aud / g_exp - and of g_att when smo_size > 0 -
 * has exactly one writer per call): the buffers need no zero fill in front of the call.  smo_size == 0: g_att untouched. */
int dfn_encode_signal_bwd_set(const float* aud_params, const float* exp_params, const float* att_params, const float* auds,
                              const float* exps, int n_total, int frame, int smo_size, const float* d_out, float* g_aud,
                              float* g_exp, float* g_att, void* stream);
int dfn_encode_signal_torso_bwd_set(const float* att_params, const float* poses, int pose_stride, int n_total, int frame,
                                    int smo_size, const float* d_out, float* g_att, void* stream);

/* ---- the fused renderer: replaces the frame loop MAIN:611-713 (and its training twin MAIN:829-899) --
 * packed_head/packed_torso, bias_head/bias_torso: from the calls above (torso ones may be NULL when
 * frame.fields == 1).  bg: background as f32 [H*W,3] in [0,1] (MAIN:477) or u8 [H*W,3]; give one.
 * pix_index: optional int32 [ray_count] of pixel ids y*W+x (training: MAIN:818-836); NULL = contiguous.
 * rgb_head [ray_count,3] (MAIN:713 rgb_head), rgb_com [ray_count,3] (MAIN:712 rgb; NULL if fields==1).
 * weights_head / weights_com: optional [ray_count, n_coarse+n_fine] volume weights of the final pass.
 * z_vals: optional [ray_count, n_coarse+n_fine] sample depths of the final pass (the merged, sorted
 * z of row H; the coarse z when n_fine == 0). */
int dfn_render_fwd(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                   const float* bias_head, const float* bias_torso, const float* bg_f32,
                   const uint8_t* bg_u8, const int32_t* pix_index, float* rgb_head, float* rgb_com,
                   float* weights_head, float* weights_com, float* z_vals, void* stream);

/* Same launch with the output stage fused: rgb8_* [ray_count,3] uint8 =
 * to8b(rgb) = (255 * clip(x, 0, 1)) truncated (HELP:17, MAIN:712-732), written from the kernel epilogue. */
int dfn_render_fwd_u8(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                      const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                      const int32_t* pix_index, uint8_t* rgb8_head, uint8_t* rgb8_com, void* stream);

/* ---- training step: replaces loss.backward() through MAIN:855-899 + DEC:277-349 (torch autograd upstream) ------
 * One step = dfn_train_fwd (the fused renderer with its recorder on: coarse samples, both fields) ->
 * [caller: loss and d loss / d rgb] -> dfn_composite_bwd -> per field: dfn_mlp_bwd (dX chain on transposed weight
 * streams, writes every pre-activation gradient feature-major), dfn_weight_grad (gradient GEMMs over the sample
 * points, scattered into a flat buffer laid out like `params`), dfn_bias_grad (gradient of the folded bias blob;
 * the caller chains it into fc_z / fc_z_skips / fc_z_view / the signal columns / the conditioning networks).
 * Buffers, NP = 64 * ray_count (a multiple of 64 in the 16-bit tier):
 *   samples, dsamples  f32 [NP][8]  (sigma_h, rgb_h[3], sigma_t, rgb_t[3]) and their gradients
 *   act_<field>        inputs of every GEMM.   f32 tier: f32 [NP/32][dfn_train_rows(field,0)][32] (feature-major per 32-point
 *                      tile).  16-bit tier (DFN_TIER_BF16), default: MX-fp4, u8 [NP/32 + 1][dfn_train_rows(field,6)]: per tile
 *                      one 512-BYTE block per 32 feature rows - e2m1 nibbles, point-major ([point][half][8 bytes = 16 nibbles,
 *                      nibble r = accumulator register r]: feature = the MFMA C/D map), 16 bytes per row and tile - then a
 *                      128-byte block of E8M0 scales, one per 32-row block: the narrowest B-operand format of
 *                      v_mfma_scale_f32_32x32x64_f8f6f4, which the weight-gradient GEMMs run on.  ONE TILE MORE than NP/32 must
 *                      be allocated (readable, contents ignored): the GEMMs' 1-KiB DMA pieces carry two 512-byte blocks and
 *                      may read up to 512 bytes behind the last block of the last tile.
 *                      With DFN_TRAIN_ACT_E4M3 or'ed into `tier` (below): MX-fp8 e4m3, u8 [NP/32 + 1][dfn_train_rows(field,8)],
 *                      1-KiB blocks of 32 bytes per row and tile in the same point-major order
 *   masks_<field>      u32 [NP/32][dfn_train_rows(field,2)][64]   ReLU bits
 *   dy_T               pre-activation gradients: MX-fp8 e4m3 always, u8 [NP/32][dfn_train_rows(field,7)] (rows
 *                      dfn_train_rows(field,1); f32 tier: f32 [NP/32][rows][32])
 *   workspace          f32 [dfn_train_rows(field,3)]  split-K partial slices; the reduction adds them in a fixed
 *                      order (no float atomics): the gradients are bit-reproducible run to run                      */
long dfn_train_rows(int field, int what);
/* Format of the activations the fused 16-bit training forward records, chosen PER CALL: or this flag into the `tier` argument of
 * dfn_train_fwd / _hier / _loss / _hier_loss (DFN_TIER_BF16 | DFN_TRAIN_ACT_E4M3) and the recorder writes e4m3 (8 bits) instead
 * of e2m1 (4 bits); pass DFN_ACT_E4M3 to dfn_weight_bias_grad_fmt / _partials for that step's weight gradients.  The run-time
 * opt-out of the narrow format: A/B runs of the two on real data in one process (a 1-mantissa-bit activation feeds every weight
 * gradient; evidence for it: tests/test_gpu_convergence.py, profiles/r05_convergence.txt).  Every other entry point rejects the flag. */
#define DFN_TRAIN_ACT_E4M3 0x100
long dfn_packed_bwd_bytes(int tier, int field);
int dfn_pack_weights_bwd(int tier, int field, const float* params, void* packed_T, void* stream);

/* Everything a training step derives from the parameters before its forward, in one launch: dfn_fold_bias for both fields
 * (z_shape / z_app: [2][256], row 0 head, row 1 torso) and dfn_pack_weights + dfn_pack_weights_bwd for both fields.  Same
 * outputs as the six separate calls (which stay, for single-field callers). */
int dfn_train_prepare(int tier, const float* params, const float* signal_head, const float* signal_torso,
                      const float* z_shape, const float* z_app, void* packed_head, void* packed_torso, void* packed_T_head,
                      void* packed_T_torso, float* bias_head, float* bias_torso, void* stream);
int dfn_train_fwd(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                  const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                  const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                  uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, void* stream);
/* The hierarchical variant of the training forward (SURVEY.md 8(a) row H under autograd; 8(d) "report also the hierarchical
 * variant"; the composition of MAIN:119-124 `N_importance` + HELP:537-581 sample_pdf in the NeRF lineage): frame->n_fine = 64
 * or 128.  The fine depths are constants (sample_pdf's output is detached, as the lineage does), the loss sees the images
 * composited over the merged 64 + n_fine samples.  Every point is evaluated once with the recorder on, in EVALUATION order
 * (per ray: the 64 coarse points, then the n_fine fine ones): NP = (64 + n_fine) * ray_count for samples / act / masks /
 * dy_T.  Two more outputs feed dfn_composite_bwd_hier: z_all f32 [ray_count][64 + n_fine] (merged, sorted depths) and
 * ranks u8 [ray_count][64 + n_fine] (merged rank of evaluated point i).  dfn_mlp_bwd / dfn_weight_bias_grad / ... are
 * the same calls with that NP. */
int dfn_train_fwd_hier(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                       const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                       const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                       uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, float* z_all, uint8_t* ranks,
                       void* stream);
int dfn_composite_bwd_hier(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                           const float* samples, const float* z_all, const uint8_t* ranks, const float* d_rgb_head,
                           const float* d_rgb_com, float* dsamples, void* stream);
/* The pixel draw of a training step in one launch: replaces MAIN:786-820 (np.random.choice(replace=False) over the image,
 * or over (face rect | lower half) and its complement with --sample_rate > 0).  pix_index [n] receives n DISTINCT pixel
 * ids y*W+x, uniform over the subsets, in random order; with rect_num > 0 the first rect_num lie inside (rect | lower
 * half), the rest outside; rect = device int32 [4] = (y0, x0, h, w) of the frame's face rectangle (LOAD:sample_rects).
 * Counter-based generator: (seed, counter) -> the draw (the caller increments counter per step).  8192 candidates are
 * drawn per call: H*W < 2^31, n <= 4096, and every class must contain comfortably more pixels than it is asked for
 * (status[0..1], optional device int32 [2], returns the distinct candidates found inside / outside). */
int dfn_sample_pixels(int H, int W, int n, int rect_num, const int32_t* rect, uint64_t seed, uint64_t counter,
                      int32_t* pix_index, int32_t* status, void* stream);

/* The loss of a training step in one launch: replaces target[select_coords] (MAIN:791-800: the pixels pix_index of the
 * head and the composite ground-truth images, here uint8 [H*W,3] resident on the device, / 255 as LOAD:58-60), the two
 * img2mse (HELP:13; MAIN:902-907) and their autograd:
 *   losses [3]: losses[0] = mean((rgb_head - target_head)^2), losses[1] = mean((rgb_com - target_com)^2) (means over 3 n
 *   values), losses[2] = losses[1] + losses[0] (the step's loss, MAIN:902-907)
 *   d_rgb_head / d_rgb_com [n,3] = 2 (rgb - target) / (3 n)  = d (losses[0] + losses[1]) / d rgb
 * Fixed reduction order (bit-reproducible). */
int dfn_mse_loss_u8(const float* rgb_head, const float* rgb_com, const uint8_t* img_head, const uint8_t* img_com,
                    const int32_t* pix_index, int n, float* losses, float* d_rgb_head, float* d_rgb_com, void* stream);
/* The same loss formed in the EPILOGUE of the training forward (one launch less between the forward and the dX chain: a
 * single-workgroup loss kernel is 17-19 us of a 1-ms step): every ray's wave gathers its own target pixel, writes its row of
 * d_rgb_head / d_rgb_com and its squared error; a workgroup adds its rays' in ray order, the workgroup that finishes LAST
 * (a ticket in `workspace`) adds the workgroups' partial sums in workgroup order - fixed orders: bit-reproducible run to run;
 * the sums are the same real numbers as dfn_mse_loss_u8's in another order (the two agree to rounding, not bit for bit).
 *   workspace: dfn_train_loss_floats(ray_count) floats that must read ZERO before the first call; the call leaves the
 *   ticket zero again (launches sharing a workspace must not overlap).  No state inside the library.
 * frame->ray_count rays, pix_index as in the forward (null: ray_begin + r).                                            */
typedef struct DfnTrainLoss {
    const uint8_t* img_head;   /* uint8 [H*W,3] ground-truth frames resident on the device */
    const uint8_t* img_com;
    float* d_rgb_head;         /* out [ray_count,3] */
    float* d_rgb_com;          /* out [ray_count,3] */
    float* losses;             /* out [3], as dfn_mse_loss_u8 */
    float* workspace;
} DfnTrainLoss;
long dfn_train_loss_floats(int ray_count);
int dfn_train_fwd_loss(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                       const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                       const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                       uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, const DfnTrainLoss* loss, void* stream);
int dfn_train_fwd_hier_loss(int tier, const DfnFrame* frame, const void* packed_head, const void* packed_torso,
                            const float* bias_head, const float* bias_torso, const float* bg_f32, const uint8_t* bg_u8,
                            const int32_t* pix_index, float* rgb_head, float* rgb_com, float* samples, void* act_head,
                            uint32_t* masks_head, void* act_torso, uint32_t* masks_torso, float* z_all, uint8_t* ranks,
                            const DfnTrainLoss* loss, void* stream);
int dfn_composite_bwd(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                      const float* samples, const float* d_rgb_head, const float* d_rgb_com, float* dsamples,
                      void* stream);
/* The two compositing backward calls with a buffer the same launch fills with zeros (zero_floats floats at zero_buf, 16-byte
 * aligned; NULL = none): the step's flat gradient buffer, whose own fill launch sat between this call and the dX chain. */
int dfn_composite_bwd_z(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                        const float* samples, const float* d_rgb_head, const float* d_rgb_com, float* dsamples,
                        float* zero_buf, long zero_floats, void* stream);
int dfn_composite_bwd_hier_z(const DfnFrame* frame, const int32_t* pix_index, const float* bg_f32, const uint8_t* bg_u8,
                             const float* samples, const float* z_all, const uint8_t* ranks, const float* d_rgb_head,
                             const float* d_rgb_com, float* dsamples, float* zero_buf, long zero_floats, void* stream);
int dfn_mlp_bwd(int tier, int field, const void* packed_T, const float* samples, const float* dsamples,
                const uint32_t* masks, long NP, void* dy_T, void* stream);
/* The weight-gradient plan of `field` (0 head, 1 torso), for tests and tools: what == 0 -> the GEMM list, 6 int32 per GEMM
 * {a_row, M, b_row, N, c_off, bias_owner}: C[M x N] = dy_T rows [a_row, a_row + M) x act_T rows [b_row, b_row + N)^T summed over the
 * points, stored row-major at c_off of the dense partial array; what == 1 -> map[i] = index into the flat decoder parameter vector
 * the dense element i is added to (-1: structural padding); what == 2 -> bias_rows[e] = dy_T row whose sum over the points is the
 * gradient of bias-blob element e (-1: none).  Returns the number of int32 (out may be NULL to ask), negative on error.
 * (The backward of Decoder.forward, decoder.py:291-349: one GEMM per nn.Linear the field evaluates.) */
long dfn_wgrad_plan(int field, int what, int32_t* out, long capacity);
/* (16-bit tier: dfn_weight_grad / dfn_weight_bias_grad read act_T in the fused step's DEFAULT format, MX-fp4 - 16 bytes per row and
 * tile, dfn_train_rows(field, 6) bytes per tile, NP/32 + 1 tiles allocated, see above; for e4m3 arrays use the _fmt entry point) */
int dfn_weight_grad(int tier, int field, const void* dy_T, const void* act_T, long NP, float* workspace,
                    float* grad_flat, void* stream);
/* workspace: f32 [dfn_train_rows(field,4)] (partial row sums per slice of the points) */
int dfn_bias_grad(int tier, int field, const void* dy_T, long NP, float* workspace, float* dbias, void* stream);
/* dfn_weight_grad and dfn_bias_grad in ONE pass over dy_T (the GEMM that owns a block of dy_T rows multiplies it by a
 * tile of ones as well): same outputs, the gradient array is read once less. */
int dfn_weight_bias_grad(int tier, int field, const void* dy_T, const void* act_T, long NP, float* workspace,
                         float* grad_flat, float* dbias, void* stream);
/* The same with the format of act_T as an argument (16-bit tier; ignored in the f32 tier).  The FUSED step (dfn_train_fwd /
 * dfn_train_fwd_hier) records its activations as MX-fp4 - e2m1 nibbles, 16 bytes per row and tile: dfn_train_rows(field, 6)
 * bytes per tile, and the caller allocates one tile more than NP / 32 (1-KiB DMA pieces) - because a weight gradient of that
 * step sums >= 131,072 points and the rounding averages out (LABNOTES.md 7, round 4).  dfn_decoder_train_fwd (Decoder.forward on
 * explicit points under autograd: any number of points) records e4m3 - 32 bytes per row and tile, dfn_train_rows(field, 8).
 * dfn_weight_grad / dfn_weight_bias_grad take the fused step's format. */
#define DFN_ACT_E4M3 0
#define DFN_ACT_E2M1 1
int dfn_weight_bias_grad_fmt(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP, float* workspace,
                             float* grad_flat, float* dbias, void* stream);
/* dfn_weight_bias_grad_fmt in its two stages, for a caller that has something to wait for in between: _partials runs the GEMMs
 * (every (GEMM, slice of the points) writes its own slice of `workspace`; nothing outside it is touched), _reduce adds the slices
 * in index order into grad_flat (+=) and writes dbias.  The two fields of a decoder share most parameters, so their _reduce
 * stages must run one after the other (a fixed order: bit-reproducible sums) - their _partials stages need not: the training
 * step launches the torso's GEMMs without waiting for the head's reduction on the other stream (a cross-queue wait in front of a
 * 130-us kernel was 17 us of the step's critical path) and waits in front of the torso's reduction instead.  Same NP,
 * workspace and tier / field in both calls; both tiers: the row sums ride in the GEMMs as per-slice partials in `workspace`,
 * dbias is written by _reduce (pass the same dbias to both; _partials does not touch it). */
int dfn_weight_bias_grad_partials(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP,
                                  float* workspace, float* dbias, void* stream);
int dfn_weight_bias_grad_reduce(int tier, int field, long NP, float* workspace, float* grad_flat, float* dbias, void* stream);
/* f32 tier: _partials in its two launches, for a caller that runs them on different streams - which = 1: the 256 x 256 GEMMs (89 %
 * of a field's FLOPs, matrix-pipe-bound, 64 KiB of LDS), 2: all the others (HBM-bound, 72 KiB: a workgroup of each fits one
 * compute unit), 3: both = _partials.  The launches write disjoint pieces of `workspace`; _reduce needs both.  (The training
 * step puts the LAST field's narrow launch on the other field's stream, beside its 256 x 256 launch.)  16-bit tier: which = 3 only. */
int dfn_weight_bias_grad_partials_part(int tier, int field, int act_format, const void* dy_T, const void* act_T, long NP,
                                       float* workspace, float* dbias, int which, void* stream);
/* Backward of dfn_fold_bias (the fold is linear; upstream it is the autograd of DEC:293-295, 311, 318, 332):
 * dbias [dfn_bias_floats] -> grad_flat (+=, layout of `params`: fc_z / fc_z_skips / fc_z_view, the signal columns
 * of fc_in / fc_p_skips / the deformation nets, every bias) and d_signal (+=, [96] head / [42] torso; may be NULL).
 * z_shape / z_app get no gradient: they are constants upstream (never handed to an optimizer, MAIN:522-547). */
int dfn_fold_bias_bwd(int tier, int field, const float* params, const float* signal, const float* z_shape,
                      const float* z_app, const float* dbias, float* grad_flat, float* d_signal, void* stream);

/* d_signal (OVERWRITTEN, [96] head / [42] torso) of one field straight from the recorded pre-activation gradients: the row sums
 * of the few dy_T rows whose bias elements fold a signal term (head: fc_in / fc_p_skips, torso: four deformation
 * vectors; 8 % of dy_T), then the signal part of dfn_fold_bias_bwd.  Same value as dfn_weight_bias_grad +
 * dfn_fold_bias_bwd produce (up to f32 summation order), but it does not wait for the weight-gradient GEMMs: the
 * conditioning networks' backward (autograd of MAIN:28-111) can run on a second stream underneath them.
 * workspace: f32 [dfn_train_rows(field,5)], private to this call (not the dfn_weight_grad workspace). */
int dfn_signal_grad(int tier, int field, const float* params, const void* dy_T, long NP, float* workspace, float* d_signal,
                    void* stream);

/* hipMemsetAsync(p, 0, bytes) on `stream`: lets a host that juggles several streams zero a buffer on a given one without
 * switching its framework's current stream (torch: a context manager per fill). */
int dfn_zero_async(void* p, long bytes, void* stream);

/* ---- optimizer step: replaces torch.optim.Adam.step() of MAIN:522-547 / 924-931 (betas (0.9, 0.999), no weight
 * decay, no amsgrad) for a list of tensors in ONE launch.  items [n_items] and chunks [n_chunks] live in DEVICE memory;
 * chunk c = {item index, chunk index within the item} covers elements [chunk * DFN_ADAM_CHUNK, ...) of that item.
 *   exp_avg    = lerp(exp_avg, grad, 1 - beta1);   exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) grad^2
 *   param     -= (lr / bias_c1) * exp_avg / (sqrt(exp_avg_sq) / bias_c2_sqrt + eps)
 * with bias_c1 = 1 - beta1^t, bias_c2_sqrt = sqrt(1 - beta2^t) computed by the caller (t = step count from 1). */
#define DFN_ADAM_CHUNK 2048
typedef struct DfnAdamItem {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long n;
} DfnAdamItem;
int dfn_adam_multi(const DfnAdamItem* items_dev, const int32_t* chunks_dev, int n_chunks, float lr, double beta1,
                   double beta2, float eps, float bias_c1, float bias_c2_sqrt, void* stream);

/* ---- Decoder.forward on explicit points: replaces DEC:277-349 -------------------------------------------
 * points, dirs [n,3]; feat [n,3] (after sigmoid), sigma [n] (raw). */
int dfn_decoder_fwd(int tier, int field, const void* packed, const float* bias, const float* points,
                    const float* dirs, long n, float* feat, float* sigma, void* stream);

/* The same evaluation with the training recorder on: replaces DEC:277-349 when the reference's training loop calls
 * decoder(...) on explicit points under autograd (MAIN:855-866).  field: DFN_FIELD_HEAD or DFN_FIELD_TORSO; tiers
 * f32 / bf16.  With NP = n rounded up to a multiple of 32 (padding points repeat point n - 1):
 *   samples [NP][8] (this field's (sigma, rgb) in floats 0..3 (head) / 4..7 (torso); the other half untouched),
 *   act_T [NP/32][dfn_train_rows(field,0)][32], masks [NP/32][dfn_train_rows(field,2)][64].
 * Backward: fill dsamples [NP][8] with d loss / d (sigma, rgb) (zeros for the padding points), then dfn_mlp_bwd,
 * dfn_weight_bias_grad and dfn_fold_bias_bwd with NP as above. */
int dfn_decoder_train_fwd(int tier, int field, const void* packed, const float* bias, const float* points,
                          const float* dirs, long n, float* feat, float* sigma, float* samples, void* act_T,
                          uint32_t* masks, void* stream);

/* ---- building blocks kept for API parity ---------------------------------------------------------------- */
/* get_rays, HELP:449-465: rays_o, rays_d [H*W,3] for c2w (3x4, host memory, 12 floats). */
int dfn_get_rays(int H, int W, float focal, float cx, float cy, const float* c2w_host, float* rays_o,
                 float* rays_d, void* stream);
/* ... with HELP:449's `stride`: (H / stride) x (W / stride) rays through the pixel positions torch.linspace(0, W - 1, W / stride) x
 * torch.linspace(0, H - 1, H / stride) (not integers for stride > 1); rays_o, rays_d [(H / stride) * (W / stride), 3].  Dead upstream
 * (every caller passes stride 1); kept for the signature. */
int dfn_get_rays_strided(int H, int W, int stride, float focal, float cx, float cy, const float* c2w_host, float* rays_o,
                         float* rays_d, void* stream);
/* ndc_rays, HELP:484-503 (n rays). */
int dfn_ndc_rays(int H, int W, float focal, float z_near, const float* rays_o, const float* rays_d, long n,
                 float* out_o, float* out_d, void* stream);
/* sample_pdf, HELP:537-581: bins [R,nb], weights [R,nb-1] -> samples [R,ns].  u: optional [R,ns] uniform
 * draws (the reference's rand / pytest modes); NULL = det=True (linspace). nb <= 256. */
int dfn_sample_pdf(const float* bins, const float* weights, long R, int nb, int ns, const float* u,
                   float* samples, void* stream);
/* composite_function, MAIN:146-166: sigma [K,N], feat [K,N,3] -> sigma_sum [N], feat_w [N,3]. */
int dfn_composite(const float* sigma, const float* feat, int K, long N, float* sigma_sum, float* feat_w,
                  void* stream);
/* calc_volume_weights, MAIN:169-179: z [R,S], ray [R,3], sigma [R,S] -> weights [R,S]. S <= 1024. */
int dfn_volume_weights(const float* z, const float* ray, const float* sigma, long R, int S,
                       float last_dist, float* weights, void* stream);
/* Backward of the two building blocks above, for a reference-shaped training loop that differentiates through them
 * (MAIN:888-899: Decoder.forward -> composite_function -> calc_volume_weights -> sum -> img2mse -> loss.backward()); what
 * torch autograd computes through MAIN:146-166 / 169-179:
 *   dfn_composite_grad: (d_sigma_sum [N], d_feat_w [N,3]; either may be NULL = zero) -> d_sigma [K,N], d_feat [K,N,3].  The
 *     `denom_sigma[denom_sigma == 0] = 1e-4` of MAIN:160 is an in-place constant write: no gradient flows through the
 *     denominator at those samples.
 *   dfn_volume_weights_grad: d_weights [R,S] -> d_sigma [R,S] (relu'(sigma) = [sigma > 0]; z and ray are constants of the
 *     training loop, MAIN:838-841: no gradient is produced for them). */
int dfn_composite_grad(const float* sigma, const float* feat, int K, long N, const float* d_sigma_sum, const float* d_feat_w,
                       float* d_sigma, float* d_feat, void* stream);
int dfn_volume_weights_grad(const float* z, const float* ray, const float* sigma, long R, int S, float last_dist,
                            const float* d_weights, float* d_sigma, void* stream);
/* to8b, HELP:17: (255*clip(x,0,1)) truncated to u8. */
int dfn_to8b(const float* x, long n, uint8_t* out, void* stream);

/* ---- debug / self-test ------------------------------------------------------------------------------------ */
/* Runs one v_mfma_f32_32x32x16_bf16, one v_mfma_f32_32x32x2_f32 and one v_mfma_f32_32x32x16_f16 with
 * A = (row,k) / B = (k,col) probes and writes D as the kernels interpret it: out[3][32][32] (bf16, f32, f16).
 * Used by tests to pin the fragment maps. */
int dfn_debug_mfma_layout(float* out, void* stream);
/* Power-ceiling probe (bench.py `roofline.power_ceiling`): the renderer's inner loop reduced to its cost drivers - 8 waves per
 * workgroup, one workgroup per compute unit, the tier's v_mfma_f32_32x32x16 on two alternating accumulator sets with
 * `lds_reads_per_2` 1-KiB fragment reads from LDS and `valu_per_2` convert / max instructions per TWO MFMAs - on the caller's
 * operands: fragments = 32 KiB (32 A fragments in the tier's 16-bit type, e.g. a slab of the packed weight stream), operands_b =
 * 64 KiB.  Variants: (0, 0) the bare chain, (2, 4) the renderer's mix.  Each of the blocks x 8 waves issues 32 x iters MFMAs
 * (16384 MACs = 32768 FLOP each); out: blocks x 512 floats (sink); clock[0] / clock[1] = shader cycles / 100-MHz ticks of one wave. */
int dfn_debug_mfma_chain(int tier, int lds_reads_per_2, int valu_per_2, const void* fragments, const void* operands_b, int iters,
                         int blocks, float* out, uint64_t* clock, void* stream);
/* Measurement aid (process-global; NULL = off, the default): while set, every dfn_render_fwd* launch makes the workgroup
 * in the middle of its grid write probe[0] = shader cycles (s_memtime) and probe[1] = 100 MHz ticks (s_memrealtime) of
 * its own lifetime into this device array of two uint64: probe[0] / probe[1] x 0.1 GHz = the effective shader clock
 * under load (the launch is power-bound: bench.py reports it next to the roofline fraction). */
int dfn_debug_clock_probe(uint64_t* probe);

#ifdef __cplusplus
}
#endif
#endif /* DFANERF_H_ */
