// The opaque handle behind dsmi_model* (include/dsmi.h).
#pragma once
#include "common.h"

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct ConvW { float *wp = nullptr, *bias = nullptr, *bn_a = nullptr, *bn_b = nullptr; uint16_t* wp_sp = nullptr; };

struct RnnW {
    float* wih = nullptr;   // [Np][ldw] gate-permuted (see rnn_src_row), zero padded
    uint16_t* wih_sp = nullptr;  // the same as tiled two-term fp16 split (pack_gemm_w_split)
    float* bih = nullptr;   // [Np]
    float* whh[2] = {nullptr, nullptr};  // packed MFMA operand stream per direction
    uint16_t* whh_sp[2] = {nullptr, nullptr};  // two-term fp16 split of the same, persistent kernel
    float* bhh[2] = {nullptr, nullptr};  // torch layout [G*H]
    // the same x-projection weights permuted for the 16-unit geometry of rnn_persist16.hip (H % 16 == 0 only)
    uint16_t* wih16_sp = nullptr; float* bih16 = nullptr; uint16_t* whh16_sp[2] = {nullptr, nullptr};
    float* bn_a = nullptr;  // [Hs] BatchNorm1d in front of layers >= 1
    float* bn_b = nullptr;
    int K = 0, ldw = 0;
};

// Per-kernel dispatch timing (profiling level 2): sampled launches carry an event pair stamped
// with the dispatch's own begin/end timestamps; pairs are resolved lazily after a sync.
enum KernelKind { KK_STFT = 0, KK_CONV1, KK_CONV2, KK_CONV3, KK_GEMM0, KK_GEMM, KK_STEP, KK_HEAD, KK_GREEDY, KK_BEAM, KK_PERSIST, KK_COUNT };
struct KernelTimer {
    std::vector<hipEvent_t> free_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[KK_COUNT];
    double sum_us[KK_COUNT] = {0};
    int64_t samples[KK_COUNT] = {0};
    int64_t launches[KK_COUNT] = {0};
    double flops[KK_COUNT] = {0};     // algorithmic FLOPs summed over ALL launches of the kind
    double bytes[KK_COUNT] = {0};     // algorithmic bytes summed over all launches
};

struct dsmi_model {
    dsmi_model_desc desc{};
    int device = 0;
    bool finalized = false;
    std::string err;
    std::map<std::string, HostTensor> tensors;

    // geometry
    int n_fft = 0, n_freq = 0;
    int conv_fi[3] = {0, 0, 0}, conv_fo[3] = {0, 0, 0};
    int I0 = 0, Hs = 0;
    dsmi::RnnGeom geom{};
    dsmi::RnnGeom geom16{};       // U = 16 geometry of the second-generation persistent kernel
    bool have16 = false;          // geom16 weights were packed (H % 16 == 0)
    int ring_windows = 0;         // dsmi_model_set_ring_windows: ring windows per recurrent layer side by side (0: by inflight)
    int inflight = 1;             // dsmi_model_set_inflight: batches the caller keeps in flight on this device (2: throughput variant)

    // weights (device)
    ConvW conv[3];
    std::vector<RnnW> rnn;
    float* look_w = nullptr;
    float *fc_a = nullptr, *fc_b = nullptr, *fc_wp = nullptr;
    std::vector<void*> owned;

    // workspaces (device), sized by dsmi_reserve
    int cap_B = 0, cap_T = 0;
    std::vector<void*> ws;
    float* conv_buf[2] = {nullptr, nullptr};
    uint16_t* conv_buf_sp[2] = {nullptr, nullptr};   // split channels-last intermediates between conv layers
    float* xp = nullptr;
    float* hbuf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float* cst[2] = {nullptr, nullptr};
    float* look_buf = nullptr;
    float* hpack = nullptr;
    uint16_t* hpack_sp = nullptr;
    uint16_t* hpack16 = nullptr;   // packed split state of rnn_persist16.hip
    uint16_t* a_sp = nullptr;       // split A operand of the x-projection GEMM
    unsigned* pcnt = nullptr;     // persistent-kernel step counters [layers][D*ceil(B/32)][T]
    unsigned* perr = nullptr;     // persistent-kernel timeout word (device)
    // ---- a forward whose persistent kernel timed out is recomputed by dsmi_forward_status (api.hip).  Forwards are
    // asynchronous, so a small ring remembers the ones whose status has not been collected yet (oldest first).
    struct FwdSlot {
        const float* feat = nullptr; float* probs = nullptr; std::vector<int32_t> lens; int B = 0, T = 0; void* stream = nullptr;
        hipEvent_t done = nullptr;     // recorded behind the error word's copy at the end of the forward
        unsigned* err_host = nullptr;  // pinned mirror of the error word as that forward left it
    };
    static constexpr int kFwdRing = 4;
    FwdSlot fwd[kFwdRing];
    int fwd_head = 0, fwd_count = 0;   // oldest uncollected slot, number of uncollected forwards
    int recomputed = 0;            // forwards recomputed on the per-step path so far
    unsigned spin_limit = dsmi::kPersistSpinLimit;   // DSMI_DEBUG_SPIN_LIMIT
    int drop_layer = -1, drop_wg = -1, drop_step = -1;   // DSMI_DEBUG_DROP_SIGNAL=layer:workgroup:step (tests: force a timeout)
    int lanes = 2, lane = 0;       // persistent kernels of a handle whose caller keeps two batches in flight take half of the CUs, on this lane
    int persist_lock_fd = -1;      // this process holds the device's persistent-kernel lock file
    // pinned staging of the per-batch lengths (pageable memory must not back an async copy)
    static constexpr int kStage = 4;
    int32_t* lens_stage = nullptr; int stage_cap = 0, stage_next = 0;
    hipEvent_t stage_ev[kStage] = {nullptr, nullptr, nullptr, nullptr}; bool stage_used[kStage] = {false, false, false, false};
    int n_cus = 0;
    bool conv1_split = true;      // first conv layer on the split-fp16 MFMA (conv1_split.hip); DSMI_DENSE_MODE=f32: conv.hip
    int conv_mode = 1;            // 1: split-fp16 conv for the 32-input-channel layers, 0: fp32 MFMA conv
    int gemm_mode = 1;            // 1: split-fp16 GEMM, 0: fp32 MFMA GEMM
    int rnn_mode = 1;             // 1: persistent layer kernel when eligible, 0: one launch per step
    bool ring4 = false;           // DSMI_RNN_KERNEL=ring4: the four-wave ring kernel also for windows of one or two tiles
    bool ring8 = false;           // DSMI_RNN_KERNEL=ring8: the eight-wave ring kernel where the four-wave one would run
    int rnn_kernel = 0;           // DSMI_RNN_KERNEL: 0 auto, 1 "duo" (never the ring kernel), 2 "ring" (also for a lone batch <= 32 clips)
    int persist_gen = 2;          // 2: rnn_persist16.hip where eligible (DSMI_RNN_MODE=persist8 selects the first generation)
    float* xin = nullptr;
    std::vector<int32_t> host_out_lens;   // output lengths of the batch being processed
    int32_t *lens_dev = nullptr, *sizes_dev = nullptr, *raw_ids = nullptr, *ids = nullptr, *offs = nullptr, *nout = nullptr;

    // profiling
    int profiling = 0;        // 0 off, 1 stage events, 2 + sampled per-kernel dispatch timestamps
    KernelTimer kt;
    hipEvent_t ev[8];
    double stage_us[5] = {0, 0, 0, 0, 0};
    int64_t n_step_launches = 0;
    double step_flops = 0, total_flops = 0;
};

dsmi::EvPair timer_arm(dsmi_model* m, int kind, bool sample, double flops, double bytes);
