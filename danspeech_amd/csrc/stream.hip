// Chunked unidirectional inference on the GPU (gfx950): the handle behind dsmi_stream*.
//
// Replaces DeepSpeech(streaming_inference_model=True).streaming_forward of the reference
// (danspeech/deepspeech/model.py:517-537) together with the state its modules carry between calls:
//   MaskConvStream   (model.py:156-201)  the last 10 input frames of each conv layer (after the chunk's
//                                        own first/last padding), glued in front of the next chunk;
//   BatchRNNStream   (model.py:204-238)  h (and c) of every unidirectional layer;
//   LookaheadStream  (model.py:241-283)  the frames whose right context has not arrived yet; the first
//                                        pass only fills this buffer and yields no output.
// The arithmetic runs on the same kernels as the offline path in their fp32-MFMA form: conv_kernel<L>
// on explicitly assembled inputs (full-length mask), gemm_f32 for the x-projection, rnn_step_kernel with
// its carried-state entry (hcarry / parity offset), lookahead_kernel and head_kernel.  One utterance
// (B = 1) per stream; a model can serve any number of streams, each owns its buffers.
// Only 2-conv models: the reference's streaming_init sizes the first RNN layer for two conv layers
// whatever conv_layers says (model.py:476-484) and builds a non-streaming MaskConv for one.
#include "model.h"

#include <algorithm>

using namespace dsmi;

struct dsmi_stream {
    dsmi_model* m = nullptr;
    std::string err;
    // carried state
    float* left[2] = {nullptr, nullptr};     // [ci*fi][10] per conv layer
    bool has_left = false;
    std::vector<float*> hpack;               // per layer [2][nq][64][4]
    std::vector<float*> hcarry, ccarry;      // per layer [Hs]
    std::vector<int> pbase;
    bool has_hidden = false;
    float* la_buf = nullptr;                 // [la_cap][Hs]
    int la_rows = 0, la_cap = 0;
    bool la_init = false;
    int32_t* lens_dev = nullptr;             // [2]: full-length masks of the two conv launches; [2] = INT_MAX for the RNN
    // per-call workspaces (grow only)
    int cap_T = 0;
    float *xin1 = nullptr, *y1 = nullptr, *xin2 = nullptr, *y2 = nullptr, *xp = nullptr, *hb[2] = {nullptr, nullptr};
    float *cat = nullptr, *la_out = nullptr;
    int cat_cap = 0;
};

static thread_local std::string g_stream_error;

#define S_HIP(st, expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { (st)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return DSMI_ERR_HIP; } \
    } while (0)

static int sfail(dsmi_stream* st, int code, const char* msg) { st->err = msg; return code; }

static void free_p(float*& p) { if (p) (void)hipFree(p); p = nullptr; }

extern "C" int dsmi_stream_create(dsmi_model* m, dsmi_stream** out) {
    if (!m || !out) { g_stream_error = "null argument"; return DSMI_ERR_INVALID; }
    if (!m->finalized) { g_stream_error = "dsmi_model_finalize has not been called"; return DSMI_ERR_NOT_READY; }
    if (m->desc.bidirectional || m->desc.conv_layers != 2) {
        g_stream_error = "streaming needs a unidirectional 2-conv model (reference model.py:427-494)";
        return DSMI_ERR_INVALID;
    }
    if (hipSetDevice(m->device) != hipSuccess) { g_stream_error = "hipSetDevice failed"; return DSMI_ERR_HIP; }
    dsmi_stream* st = new dsmi_stream();
    st->m = m;
    const int L = m->desc.rnn_layers;
    st->hpack.assign(L, nullptr); st->hcarry.assign(L, nullptr); st->ccarry.assign(L, nullptr); st->pbase.assign(L, 0);
    bool ok = true;
    for (int l = 0; l < 2 && ok; ++l)
        ok = hipMalloc((void**)&st->left[l], sizeof(float) * kConvSpecs[l].ci * m->conv_fi[l] * 10) == hipSuccess;
    for (int l = 0; l < L && ok; ++l) {
        const size_t np = (size_t)2 * m->geom.nq * 256;
        ok = hipMalloc((void**)&st->hpack[l], sizeof(float) * np) == hipSuccess &&
             hipMemset(st->hpack[l], 0, sizeof(float) * np) == hipSuccess &&
             hipMalloc((void**)&st->hcarry[l], sizeof(float) * m->Hs) == hipSuccess &&
             hipMalloc((void**)&st->ccarry[l], sizeof(float) * m->Hs) == hipSuccess &&
             hipMemset(st->ccarry[l], 0, sizeof(float) * m->Hs) == hipSuccess;
    }
    ok = ok && hipMalloc((void**)&st->lens_dev, sizeof(int32_t) * 4) == hipSuccess;
    if (!ok) { g_stream_error = "stream: HIP allocation failed"; dsmi_stream_destroy(st); return DSMI_ERR_NOMEM; }
    *out = st;
    return DSMI_OK;
}

extern "C" void dsmi_stream_destroy(dsmi_stream* st) {
    if (!st) return;
    (void)hipSetDevice(st->m->device);
    (void)hipDeviceSynchronize();
    for (int l = 0; l < 2; ++l) free_p(st->left[l]);
    for (float*& p : st->hpack) free_p(p);
    for (float*& p : st->hcarry) free_p(p);
    for (float*& p : st->ccarry) free_p(p);
    free_p(st->la_buf);
    if (st->lens_dev) (void)hipFree(st->lens_dev);
    for (float** p : {&st->xin1, &st->y1, &st->xin2, &st->y2, &st->xp, &st->hb[0], &st->hb[1], &st->cat, &st->la_out}) free_p(*p);
    delete st;
}

extern "C" const char* dsmi_stream_last_error(const dsmi_stream* st) { return st ? st->err.c_str() : g_stream_error.c_str(); }

// What is_last does to every module (model.py:234-236, 279-281; MaskConvStream simply stops storing).
extern "C" int dsmi_stream_reset(dsmi_stream* st) {
    if (!st) return DSMI_ERR_INVALID;
    st->has_left = false; st->has_hidden = false; st->la_init = false; st->la_rows = 0;
    std::fill(st->pbase.begin(), st->pbase.end(), 0);
    return DSMI_OK;
}

static int conv_t1(int tin) { return (tin + 2 * 5 - 11) / 2 + 1; }     // conv1: k_t 11, stride 2, pad 5 (model.py:455)

static int ensure(dsmi_stream* st, int T) {
    if (T <= st->cap_T) return DSMI_OK;
    dsmi_model* m = st->m;
    S_HIP(st, hipDeviceSynchronize());
    for (float** p : {&st->xin1, &st->y1, &st->xin2, &st->y2, &st->xp, &st->hb[0], &st->hb[1]}) free_p(*p);
    const int cap = std::max(T, 64) * 2;
    const int tin1 = cap + 15, to1 = conv_t1(tin1), tin2 = to1 + 15, to2 = tin2;
    const size_t f = m->n_freq;
    S_HIP(st, hipMalloc((void**)&st->xin1, sizeof(float) * f * tin1));
    S_HIP(st, hipMalloc((void**)&st->y1, sizeof(float) * 32 * m->conv_fo[0] * round_up(to1, 4)));
    S_HIP(st, hipMalloc((void**)&st->xin2, sizeof(float) * 32 * m->conv_fo[0] * tin2));
    S_HIP(st, hipMalloc((void**)&st->y2, sizeof(float) * 32 * m->conv_fo[1] * round_up(to2, 4)));
    S_HIP(st, hipMalloc((void**)&st->xp, sizeof(float) * (size_t)to2 * m->geom.Np));
    for (int i = 0; i < 2; ++i) S_HIP(st, hipMalloc((void**)&st->hb[i], sizeof(float) * (size_t)to2 * m->Hs));
    st->cap_T = cap;
    return DSMI_OK;
}

// [rows][w] sub-matrix copy between row strides
static hipError_t copy2d(float* dst, int dst_stride, const float* src, int src_stride, int rows, int w, hipStream_t s) {
    if (w <= 0 || rows <= 0) return hipSuccess;
    return hipMemcpy2DAsync(dst, sizeof(float) * dst_stride, src, sizeof(float) * src_stride, sizeof(float) * w, rows,
                            hipMemcpyDeviceToDevice, s);
}

extern "C" int dsmi_stream_forward(dsmi_stream* st, const float* feat, int T, int is_first, int is_last, float* probs,
                                   int T_out_cap, int32_t* T_out, void* stream) {
    if (!st) return DSMI_ERR_INVALID;
    if (!feat || !T_out || T < 1) return sfail(st, DSMI_ERR_INVALID, "bad stream arguments");
    dsmi_model* m = st->m;
    const dsmi_model_desc& d = m->desc;
    S_HIP(st, hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    *T_out = 0;
    if (!is_first && !st->has_left)
        return sfail(st, DSMI_ERR_INVALID, "the first chunk of an utterance must be passed with is_first (MaskConvStream has no left context)");
    int rc = ensure(st, T);
    if (rc) return rc;

    // ---- MaskConvStream (model.py:171-201): per conv layer, pad / glue the left context, remember the tail
    const int padl = is_first ? 5 : 0, padr = (!is_first && is_last) ? 5 : 0, ctxl = is_first ? 0 : 10;
    const int tin1 = ctxl + padl + T + padr, to1 = conv_t1(tin1);
    const int tin2 = ctxl + padl + to1 + padr, to2 = tin2;
    if ((!is_last && (tin1 < 10 || tin2 < 10)) || to1 < 1) return sfail(st, DSMI_ERR_INVALID, "chunk too short for the conv context");
    const int ys1 = round_up(to1, 4), ys2 = round_up(to2, 4);
    // everything that can be refused is refused here, before any carried state changes
    const bool buffering = !st->la_init || is_first;
    const int ncat = st->la_rows + to2;
    const int nout = buffering ? 0 : (is_last ? ncat : ncat - (d.context - 1));
    if (!buffering && nout < 1) return sfail(st, DSMI_ERR_INVALID, "lookahead: fewer buffered frames than the context (torch raises here too)");
    if (!buffering && (!probs || T_out_cap < nout)) return sfail(st, DSMI_ERR_CAPACITY, "probs buffer smaller than the frames this pass yields");
    const int32_t lens_host[4] = {to1, to2, 0x7fffffff, 0};
    S_HIP(st, hipMemcpyAsync(st->lens_dev, lens_host, sizeof(lens_host), hipMemcpyHostToDevice, s));
    const float* src[2] = {feat, st->y1};
    const int src_stride[2] = {T, ys1}, src_w[2] = {T, to1}, tin[2] = {tin1, tin2}, tout[2] = {to1, to2}, ysv[2] = {ys1, ys2};
    float* xin[2] = {st->xin1, st->xin2};
    float* yv[2] = {st->y1, st->y2};
    for (int l = 0; l < 2; ++l) {
        const ConvSpec& sp = kConvSpecs[l];
        const int rows = sp.ci * m->conv_fi[l];
        if (padl || padr) S_HIP(st, hipMemsetAsync(xin[l], 0, sizeof(float) * (size_t)rows * tin[l], s));
        if (ctxl) S_HIP(st, copy2d(xin[l], tin[l], st->left[l], 10, rows, 10, s));
        S_HIP(st, copy2d(xin[l] + ctxl + padl, tin[l], src[l], src_stride[l], rows, src_w[l], s));
        if (!is_last) S_HIP(st, copy2d(st->left[l], 10, xin[l] + tin[l] - 10, tin[l], rows, 10, s));
        ConvLaunch c;
        c.x = xin[l]; c.y = yv[l]; c.wp = m->conv[l].wp; c.bias = m->conv[l].bias; c.bn_a = m->conv[l].bn_a; c.bn_b = m->conv[l].bn_b;
        c.out_lens_dev = st->lens_dev + l;
        c.B = 1; c.ci = sp.ci; c.co = sp.co; c.fi = m->conv_fi[l]; c.fo = m->conv_fo[l];
        c.ti = tin[l]; c.to = tout[l]; c.xs = tin[l]; c.ys = ysv[l]; c.layer = l; c.y_sp = nullptr;
        launch_conv(c, s);
    }
    st->has_left = !is_last;

    // ---- BatchRNNStream x layers (model.py:219-238): x-projection of the chunk, then Tc steps from the carried state
    const int Tc = to2;
    for (int l = 0; l < d.rnn_layers; ++l) {
        const RnnW& r = m->rnn[l];
        GemmLaunch gl{};
        gl.w = r.wih; gl.bias = r.bih; gl.c = st->xp; gl.w_sp = nullptr; gl.a_sp = nullptr;
        gl.M = Tc; gl.N = m->geom.Np; gl.K = r.K; gl.ldw = r.ldw; gl.ldc = m->geom.Np; gl.B = 1; gl.T = Tc;
        if (l == 0) { gl.mode = GEMM_A_CONV; gl.a = st->y2; gl.ys = ys2; }
        else { gl.mode = GEMM_A_SUM_BN; gl.a = st->hb[(l - 1) & 1]; gl.a2 = nullptr; gl.alpha = r.bn_a; gl.beta = r.bn_b; gl.lda = m->Hs; }
        launch_gemm(gl, s);
        float* out = st->hb[l & 1];
        if (m->Hs != d.rnn_hidden_size) S_HIP(st, hipMemsetAsync(out, 0, sizeof(float) * (size_t)Tc * m->Hs, s));
        RnnStepLaunch sl;
        sl.g = m->geom;
        sl.whh_packed[0] = r.whh[0]; sl.whh_packed[1] = nullptr; sl.bhh[0] = r.bhh[0]; sl.bhh[1] = nullptr;
        sl.out[0] = out; sl.out[1] = nullptr; sl.cstate[0] = st->ccarry[l]; sl.cstate[1] = nullptr;
        sl.xp = st->xp; sl.lens_dev = st->lens_dev + 2; sl.B = 1; sl.T = Tc; sl.hpack = st->hpack[l];
        sl.hcarry = st->has_hidden ? st->hcarry[l] : nullptr; sl.pbase = st->pbase[l];
        for (int step = 0; step < Tc; ++step) { sl.step = step; launch_rnn_step(sl, s); }
        S_HIP(st, hipMemcpyAsync(st->hcarry[l], out + (size_t)(Tc - 1) * m->Hs, sizeof(float) * m->Hs, hipMemcpyDeviceToDevice, s));
        st->pbase[l] = (Tc + st->pbase[l]) & 1;
    }
    st->has_hidden = !is_last;
    if (is_last) std::fill(st->pbase.begin(), st->pbase.end(), 0);
    const float* x = st->hb[(d.rnn_layers - 1) & 1];

    // ---- LookaheadStream (model.py:256-283)
    const int ctx = d.context, Hs = m->Hs;
    auto grow_la = [&](int rows) -> int {
        if (rows <= st->la_cap) return DSMI_OK;
        float* nb = nullptr;
        S_HIP(st, hipMalloc((void**)&nb, sizeof(float) * (size_t)rows * 2 * Hs));
        if (st->la_buf) {
            S_HIP(st, hipMemcpyAsync(nb, st->la_buf, sizeof(float) * (size_t)st->la_rows * Hs, hipMemcpyDeviceToDevice, s));
            S_HIP(st, hipStreamSynchronize(s));
            (void)hipFree(st->la_buf);
        }
        st->la_buf = nb; st->la_cap = rows * 2;
        return DSMI_OK;
    };
    if (buffering) {                             // buffer the whole first chunk, no output yet
        if ((rc = grow_la(Tc))) return rc;
        S_HIP(st, hipMemcpyAsync(st->la_buf, x, sizeof(float) * (size_t)Tc * Hs, hipMemcpyDeviceToDevice, s));
        st->la_rows = Tc; st->la_init = true;
        S_HIP(st, hipGetLastError());
        return DSMI_OK;
    }
    if (ncat > st->cat_cap) {
        S_HIP(st, hipStreamSynchronize(s));
        free_p(st->cat); free_p(st->la_out);
        st->cat_cap = ncat * 2;
        S_HIP(st, hipMalloc((void**)&st->cat, sizeof(float) * (size_t)st->cat_cap * Hs));
        S_HIP(st, hipMalloc((void**)&st->la_out, sizeof(float) * (size_t)st->cat_cap * Hs));
    }
    S_HIP(st, hipMemcpyAsync(st->cat, st->la_buf, sizeof(float) * (size_t)st->la_rows * Hs, hipMemcpyDeviceToDevice, s));
    S_HIP(st, hipMemcpyAsync(st->cat + (size_t)st->la_rows * Hs, x, sizeof(float) * (size_t)Tc * Hs, hipMemcpyDeviceToDevice, s));
    const int keep = std::min(Tc, ctx - 1);      // x[-(context-1):]
    if ((rc = grow_la(keep))) return rc;
    S_HIP(st, hipMemcpyAsync(st->la_buf, x + (size_t)(Tc - keep) * Hs, sizeof(float) * (size_t)keep * Hs, hipMemcpyDeviceToDevice, s));
    st->la_rows = keep;
    // rows past ncat count as zeros in the kernel: the is_last right padding; otherwise only the first nout rows are kept
    launch_lookahead(st->cat, m->look_w, st->la_out, ncat, 1, d.rnn_hidden_size, ctx, s);
    if (is_last) { st->la_init = false; st->la_rows = 0; }

    // ---- fc + softmax (model.py:531-536)
    HeadLaunch h;
    h.bn_a = m->fc_a; h.bn_b = m->fc_b; h.w_packed = m->fc_wp; h.H = d.rnn_hidden_size; h.C = d.n_labels;
    h.T = nout; h.B = 1; h.probs = probs; h.x1 = st->la_out; h.x2 = nullptr;
    launch_head(h, s);
    S_HIP(st, hipGetLastError());
    *T_out = nout;
    return DSMI_OK;
}
