// Internal declarations shared by the libdsmi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <map>

#include "../../include/dsmi.h"

namespace dsmi {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWave = 64;  // gfx950 wavefront

// Optional per-dispatch timestamps: when both events are set the kernel is launched with
// hipExtLaunchKernelGGL, which stamps them with the dispatch's own begin/end times (the same
// clock rocprofv3's kernel trace reads), so elapsed(start, stop) is the kernel's duration
// without launch gaps.
struct EvPair { hipEvent_t start = nullptr, stop = nullptr; };
#define DSMI_LAUNCH(kern, grid, block, lds, stream, ev, ...)                                          \
    do {                                                                                              \
        if ((ev).start) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, (ev).start, (ev).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                          \
    } while (0)

// Experiment switches (timing builds with parts of a kernel removed, A/B forms that were measured and not adopted) exist only in a
// library built with -DDSMI_EXPERIMENTS (`make exp` -> ../lib/libdsmi_exp.so, which tools/exp/ load through DSMI_LIBRARY); the
// product library neither reads their environment variables nor carries their kernel instantiations.
inline const char* exp_env(const char* name) {
#ifdef DSMI_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---- conv geometry: reference model.py:359,372,389 ------------------------------
struct ConvSpec { int ci, co, kf, kt, sf, st, pf, pt; };
static const ConvSpec kConvSpecs[3] = {
    {1, 32, 41, 11, 2, 2, 20, 5},
    {32, 32, 21, 11, 2, 1, 10, 5},
    {32, 96, 21, 11, 2, 1, 10, 5},
};

// ---- launch wrappers (one per .hip file) ----------------------------------------

// conv.hip: fused Conv2d + bias + BatchNorm2d(eval) + Hardtanh(0,20) + time mask.
//   x  [B][ci][fi][xs]      (xs = time stride of the input rows)
//   y  [B][co][fo][ys]
//   wp packed weights (see pack_conv_weights), bias/bn_a/bn_b [co]
//   out_lens_dev[B]: output frames t >= out_lens[b] are written as 0.
struct ConvLaunch {
    const float* x; float* y; const float* wp; const float* bias; const float* bn_a; const float* bn_b;
    const int32_t* out_lens_dev;
    int B, ci, co, fi, fo, ti, to, xs, ys, layer;  // layer index selects the compile-time geometry
    uint16_t* y_sp = nullptr;   // optional: split channels-last output [B][fo][2][to][32] fp16 terms (hi, lo unscaled) (32-channel layers)
    EvPair ev;
};
void launch_conv(const ConvLaunch& p, hipStream_t s);

// conv_split.hip: the 32-input-channel conv layers on the fp16 MFMA with two-term split operands.
struct ConvSplitLaunch {
    const uint16_t* x_sp;       // [B][fi][2][ti][32] fp16 terms (previous layer's split output)
    const uint16_t* wp_sp;      // pack_conv_w_split
    const float* bias; const float* bn_a; const float* bn_b; const int32_t* out_lens_dev;
    float* y;                 // [B][co][fo][ys] fp32 (last conv layer) ...
    uint16_t* y_sp;             // ... or split channels-last for another split-fp16 conv layer (co == 32)
    int B, co, fi, fo, ti, to, ys;
    EvPair ev;
};
void launch_conv_split(const ConvSplitLaunch& p, hipStream_t s);
std::vector<uint16_t> pack_conv_w_split(const float* w, int co_total);
// Host-side weight packer: w [co][ci][kf][kt] -> kernel layout. Returns packed floats.
std::vector<float> pack_conv_weights(const float* w, int layer);

// conv1_split.hip: the first conv layer (1 input channel) on the fp16 MFMA with two-term split operands; takes the same
// launch description as launch_conv (layer 0) plus the pack_conv1_w_split image of the weights.
std::vector<uint16_t> pack_conv1_w_split(const float* w);
void launch_conv1_split(const ConvLaunch& p, const uint16_t* wp_sp, hipStream_t s);

// gemm.hip: C[m][n] = sum_k A[m][k] * W[n][k] + bias[n]   (fp32 MFMA 32x32x2)
enum GemmAMode {
    GEMM_A_ROWMAJOR = 0,   // A [M][K] row-major
    GEMM_A_SUM_BN = 1,     // A = (A1[m][k] + A2[m][k]) * alpha[k] + beta[k]   (A2 may be null)
    GEMM_A_CONV = 2        // A[(b,t)][k] = Y[b][k][t]  (conv layout, time stride ys); C row = t*B + b
};
struct GemmLaunch {
    int mode;
    const float* a; const float* a2; const float* alpha; const float* beta;
    const float* w; const float* bias; float* c;
    const uint16_t* w_sp = nullptr;   // pack_gemm_w_split image of w: when set (with a_sp), the split-fp16 path runs
    uint16_t* a_sp = nullptr;         // workspace for the split A operand: [m-tiles][k-tiles][2][128][32] fp16
    int M, N, K;       // N, K as stored (W is [N][K] row-major, K % 4 == 0 guaranteed by packing)
    int lda, ldw, ldc;
    int B, T, ys;      // GEMM_A_CONV: batch, frames per clip, time stride
    EvPair ev;
};
void launch_gemm(const GemmLaunch& p, hipStream_t s);
std::vector<uint16_t> pack_gemm_w_split(const float* w, int N, int K, int ldw);

// rnn_step.hip: one time step of both directions of one recurrent layer.
struct RnnGeom {
    int kind;      // DSMI_RNN_*
    int G;         // gates per unit: 3 / 4 / 1
    int H;
    int U;         // hidden units per workgroup (8: its h granules are whole 16-byte groups of the packed state)
    int nwg;       // ceil(H / U) workgroups per direction
    int Kp;        // H rounded up to 8
    int nq;        // Kp / 8 k-blocks
    int D;         // directions
    int Np;        // D * nwg * G * U : permuted + padded gate columns of the x-projection
};
RnnGeom make_rnn_geom(int kind, int H, int D);
// Row r of the packed x-projection weight / bias  <->  (dir, gate, unit) of torch's [G*H][I].
// returns -1 for padding rows.
int rnn_src_row(const RnnGeom& g, int packed_col, int* dir_out);
// w_hh [G*H][H] (torch layout) for one direction -> packed MFMA operand stream.
std::vector<float> pack_whh(const RnnGeom& g, const float* w_hh);
struct RnnStepLaunch {
    RnnGeom g;
    const float* whh_packed[2];  // per direction
    const float* bhh[2];         // per direction, torch layout [G*H]
    const float* xp;             // [T][B][Np]
    float* out[2];               // per direction [T][B][H]
    float* cstate[2];            // LSTM cell state [B][H] per direction (else null)
    const int32_t* lens_dev;     // [B] output lengths
    float* hpack;                // [2][D][ceil(B/32)][nq][64][4] packed state (double-buffered by step parity)
    int B, T, step;
    const float* hcarry = nullptr;   // unidirectional streaming: [B][Hs] state of the previous chunk (step 0 continues from it)
    int pbase = 0;                   // parity offset of hpack, so that a chunk's step 0 reads what the previous chunk's last step wrote
    EvPair ev;
    unsigned long long* dbg = nullptr;   // diagnostics: per-wave timestamps [D*nwg][8 waves][8]
};
void launch_rnn_step(const RnnStepLaunch& p, hipStream_t s);

// Polls of one hand-off wait before a persistent workgroup gives up (~1 us each: seconds, against a wait of
// microseconds when every workgroup is resident).  DSMI_DEBUG_SPIN_LIMIT overrides it per handle (tests).
constexpr unsigned kPersistSpinLimit = 1u << 22;

// rnn_persist.hip: all T steps of one layer in one launch (weights resident in registers,
// counter-based hand-off of h between workgroups).  Needs every workgroup co-resident.
struct RnnPersistLaunch {
    RnnGeom g;
    const uint16_t* whh_sp[2];     // pack_whh_split output per direction
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens_dev; uint16_t* hpack_sp;   // [2][D*ceil(B/32)][npair][2][64][8] fp16
    unsigned* counters;          // [D * ceil(B/32)][T], zeroed before the launch
    unsigned* err;               // one word, set on a wait timeout
    int B, T;
    int d0 = 0, ny = 1;          // this launch covers directions d0 .. d0+ny-1 (grid = nwg x ny workgroups, all co-resident)
    unsigned spin_limit = kPersistSpinLimit;
    int drop_wg = -1, drop_step = -1;    // test hook: see DSMI_DEBUG_DROP_SIGNAL in api.hip
    EvPair ev;
    unsigned long long* dbg = nullptr;   // diagnostics: accumulated per-wave phase times
};
bool rnn_persist_eligible(const RnnGeom& g, int B, int n_cus);
std::vector<uint16_t> pack_whh_split(const RnnGeom& g, const float* w_hh);
bool launch_rnn_persist(const RnnPersistLaunch& p, hipStream_t s);

// rnn_persist16.hip: second-generation persistent layer (16 units per workgroup, 16-clip batch tiles as separate
// chains side by side); needs the x-projection in the U = 16 geometry.
RnnGeom make_rnn_geom_u(int kind, int H, int D, int U);
struct RnnPersist16Launch {
    RnnGeom g;                   // make_rnn_geom_u(kind, H, D, 16)
    const uint16_t* whh16[2];    // pack_whh16 output per direction
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens_dev; uint16_t* hpack16;     // rnn_persist16_state_halfs(g, B) fp16 values
    unsigned* counters;          // [D * ceil(B/16)][T][kPersist16CntWords], zeroed before the launch
    unsigned* err;
    int B, T, pgroups;           // pgroups from rnn_persist16_eligible / rnn_persist16_half_eligible
    int waves = 8;               // 8: one workgroup per CU; 4: the half-CU variant (two batches in flight share every CU)
    int pair0 = 0, npairs = 0;   // paired-tile kernel: the window of tile pairs this launch carries (npairs 0: all of them)
    int tile0 = 0, ntw = 0, nwin = 1;   // ring kernel: nwin windows of ntw tiles each, from tile0, side by side (ntw 0: all tiles in one window)
    unsigned* tickets = nullptr;        // four-wave ring kernel: [nwin][2] words zeroed before the launch -> directions by XCD half (null: by blockIdx)
    unsigned spin_limit = kPersistSpinLimit;
    int drop_wg = -1, drop_step = -1;    // test hook: see DSMI_DEBUG_DROP_SIGNAL in api.hip
    EvPair ev;
    unsigned long long* dbg = nullptr;   // diagnostics: accumulated per-wave phase times
};
constexpr int kPersist16Shards = 4;                        // shards of the hand-off counter of a (chain, step) ...
constexpr int kPersist16CntWords = kPersist16Shards * 64;   // ... each on its own 256-byte line
bool rnn_persist16_eligible(const RnnGeom& g16, int B, int n_cus, int* pgroups_out);
bool rnn_persist16_half_eligible(const RnnGeom& g16, int B, int n_cus, int* pgroups_out);
std::vector<uint16_t> pack_whh16(const RnnGeom& g16, const float* w_hh);
size_t rnn_persist16_state_halfs(const RnnGeom& g16, int B);
bool launch_rnn_persist16(const RnnPersist16Launch& p, hipStream_t s);

// rnn_persist_duo.hip: one workgroup carries the two 16-clip tiles of a batch in a fixed four-slot pipeline (a 32-clip
// batch of cfgA on 100 CUs); same packed weights, x-projection order and state layout as rnn_persist16.hip.
bool rnn_persist_duo_eligible(const RnnGeom& g16, int B, int n_cus);
int rnn_persist_duo_pairs(const RnnGeom& g16, int B, int n_cus);     // tile pairs one launch can carry on n_cus CUs (0: not this shape)
bool launch_rnn_persist_duo(const RnnPersist16Launch& p, hipStream_t s);

// rnn_persist_ring.hip: a workgroup = 32 units of one direction (two adjacent 16-unit groups), walking every tile of its
// window with the tiles' states staged through an LDS ring (a 64-clip layer of cfgA on 50 CUs); same packed weights,
// x-projection order, state layout and counters as rnn_persist16.hip.
int rnn_persist_ring_tiles(const RnnGeom& g16, int B, int n_cus);    // tiles one window can walk (0: not this shape)
int rnn_persist_ring_cus(const RnnGeom& g16);                        // CUs one window occupies
bool launch_rnn_persist_ring(const RnnPersist16Launch& p, hipStream_t s);
// rnn_persist_ring4.hip: the same window on FOUR waves, one per SIMD on the whole register file (W_hh of a 16-unit group's K half
// per wave, in AccVGPRs), the cell of an item in the shadows of the next item's MFMAs.  Same CUs per window as the eight-wave form.
int rnn_persist_ring4_tiles(const RnnGeom& g16, int B, int n_cus);   // tiles one window can walk (0: not this shape)
bool launch_rnn_persist_ring4(const RnnPersist16Launch& p, hipStream_t s);

// head.hip
//   lookahead: y[t][b][h] = clip(sum_k w[h][k] * x[t+k][b][h], 0, 20)
void launch_lookahead(const float* x, const float* w, float* y, int T, int B, int H, int context, hipStream_t s);
//   FC head: probs[b][t][c] = softmax_c( W[c][:] . ((x1[t][b][:] + x2[t][b][:]) * a + b) )
struct HeadLaunch {
    const float* x1; const float* x2; const float* bn_a; const float* bn_b;
    const float* w_packed; int H, C, T, B; float* probs;
    EvPair ev;
};
std::vector<float> pack_fc(const float* w, int C, int H);
void launch_head(const HeadLaunch& p, hipStream_t s);
//   greedy: argmax + CTC collapse, one workgroup per utterance
//   raw: [B][T] scratch for the per-frame argmax
void launch_greedy(const float* probs, const int32_t* sizes_dev, int B, int T, int C, int blank,
                   int32_t* raw, int32_t* ids, int32_t* offsets, int32_t* n_out, hipStream_t s);
//   y[r][h] = a[r][h] (+ b[r][h]) for h < H, inputs with row stride Hs (stage-level API only)
void launch_add2(const float* a, const float* b, float* y, size_t rows, int H, int Hs, hipStream_t s);
//   y[r][0..Is) = x[r][0..I) zero-padded
void launch_pad_rows(const float* x, float* y, size_t rows, int I, int Is, hipStream_t s);

}  // namespace dsmi
