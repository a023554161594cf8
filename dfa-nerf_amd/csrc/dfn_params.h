// dfn_params.h - kernel argument blocks (internal; the public structs live in include/dfanerf.h)
#pragma once
#include <hip/hip_runtime.h>
#include "dfanerf.h"
#include "dfn_layout.h"

namespace dfn {

struct RenderArgs {
    DfnFrame frame;
    const char* wblob[2];       // packed weight streams: head, torso
    int nslab[2];
    const float* bias;          // [head blob | torso blob], global
    const float* bg_f32;
    const unsigned char* bg_u8;
    const int* pix_index;
    float* rgb_head;
    float* rgb_com;
    float* w_head;
    float* w_com;
    float* z_out;
    int out_u8;                 // rgb_head / rgb_com point at uint8 [ray_count,3]: to8b in the epilogue (HELP:17)
    // training recorder (all null for inference): per-sample raw outputs and per-field activations / ReLU masks
    float* samples_out;         // [ray_count][n_coarse + n_fine][8], evaluation order (coarse points, then the fine ones)
    unsigned char* ranks_out;   // hierarchical training: [ray_count][n_coarse + n_fine] merged rank of every evaluated point
    void* act_T[2];
    unsigned* masks[2];
    long NP;
    int act_e4m3;               // 16-bit training forward: act_T as MX-fp8 e4m3 instead of MX-fp4 (DFN_TRAIN_ACT_E4M3)
    // training forward with the loss in its epilogue (dfn_train_fwd_loss; losses == null: off)
    DfnTrainLoss loss;
    // debug (dfn_debug_clock_probe): the workgroup in the middle of the grid writes {shader cycles, 100 MHz ticks} of its
    // own lifetime -> the effective shader clock UNDER LOAD of this launch.  Null = off.
    unsigned long long* clock_probe;
};

struct DecoderArgs {
    const char* wblob;
    int nslab;
    int field;                  // FIELD_HEAD (also listener weights) or FIELD_TORSO
    const float* bias;
    int n_bias;
    const float* points;
    const float* dirs;
    long n_points;
    float* feat;
    float* sigma;
    // training recorder (all null for inference): dfn_decoder_train_fwd
    float* samples;             // [ceil32(n)][8]: (sigma, rgb) in floats 0..3 (head) or 4..7 (torso)
    void* act_T;
    unsigned* masks;
};

hipError_t launch_render(int tier, const RenderArgs& A, hipStream_t st);
hipError_t launch_decoder(int tier, const DecoderArgs& A, hipStream_t st);
void program_info(int tier, int field, ProgramInfo* out);

}  // namespace dfn
