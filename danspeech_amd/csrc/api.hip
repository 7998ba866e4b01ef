// C ABI of libdsmi.so (see include/dsmi.h): handle lifecycle, weight repacking, workspace
// management and the orchestration of the forward pass.  Host code only; kernels live in
// conv.hip / gemm.hip / rnn_step.hip / head.hip / features.hip / beam.hip.
#include "common.h"
#include "model.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

using namespace dsmi;

static thread_local std::string g_create_error;

#define HIP_OK(m, expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (m)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            return DSMI_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

// Count one launch of `kind`; when profiling level 2 is on and `sample` is set, hand out an
// event pair for hipExtLaunchKernelGGL.
EvPair timer_arm(dsmi_model* m, int kind, bool sample, double flops, double bytes) {
    EvPair ev;
    if (m->profiling < 2) return ev;
    KernelTimer& t = m->kt;
    t.launches[kind] += 1;
    t.flops[kind] += flops;
    t.bytes[kind] += bytes;
    if (!sample) return ev;
    // Every launch of the recurrent kernels is stamped (bench.py's roofline kernel: its mean duration is over ALL its launches of the timed
    // region); of the other kinds every FIFTH -- a stamped launch is a hipExtLaunchKernelGGL with two events, and stamping all sixteen
    // launches of every forward costs the four-lane pipeline 2 - 4 % of a 20-batch call.  Five, not four: a kind's launches per
    // forward (layer GEMM: 4 for five layers, 6 for seven, 8 for nine) share no factor with it, so the stamped launch walks through
    // the layers instead of always being the same one; bench.py weights a kind's share by launches / samples.
    // (DSMI_DEBUG_SAMPLE_EVERY: experiments)
    static const int every = [] { const char* e = exp_env("DSMI_DEBUG_SAMPLE_EVERY"); const int v = e ? std::atoi(e) : 5; return v < 1 ? 1 : v; }();
    if (kind != KK_PERSIST && kind != KK_STEP && (t.launches[kind] - 1) % every != 0) return ev;
    static const int ring_every = [] { const char* e = exp_env("DSMI_DEBUG_SAMPLE_RING_EVERY"); const int v = e ? std::atoi(e) : 1; return v < 1 ? 1 : v; }();
    if (kind == KK_PERSIST && (t.launches[kind] - 1) % ring_every != 0) return ev;
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
        if (!t.free_events.empty()) { e[i] = t.free_events.back(); t.free_events.pop_back(); }
        else if (hipEventCreate(&e[i]) != hipSuccess) return EvPair();
    }
    ev.start = e[0]; ev.stop = e[1];
    t.pending[kind].push_back({e[0], e[1]});
    return ev;
}

static void timer_resolve(dsmi_model* m) {
    KernelTimer& t = m->kt;
    for (int k = 0; k < KK_COUNT; ++k) {
        for (auto& pr : t.pending[k]) {
            float ms = 0.f;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                t.sum_us[k] += ms * 1e3;
                t.samples[k] += 1;
            }
            t.free_events.push_back(pr.first);
            t.free_events.push_back(pr.second);
        }
        t.pending[k].clear();
    }
}

// Persistent kernels need every workgroup co-resident, so two of them must never share the device.
//  * Within the process: one gate per device.  wait(gate) -> launch -> record(gate) runs under the gate's mutex, so
//    two host threads with two handles / streams always chain their launches one behind the other.
//  * Across processes: the first handle of a process takes an exclusive flock on a per-device lock file (named by
//    the device's PCI bus id, so HIP_VISIBLE_DEVICES remapping does not matter); a process that cannot get it runs
//    its recurrent layers on the per-step path.  DSMI_PERSIST_SHARED=1 skips the lock (one process per GPU is
//    guaranteed by the caller).
//  * The device is four gate slots of a quarter of the CUs each (DSMI_PERSIST_LANES=1: one).  A persistent kernel whose grid
//    fits a quarter takes ONE slot (handle-affine: four batches in flight on four handles and streams run their recurrent
//    layers side by side), one that fits half takes a PAIR of slots (two batches in flight), anything else all four.
constexpr int kMaxLanes = 4;
//  * Launches that take the WHOLE device (the tile-walking kernel of H > 896, the paired-tile windows, the first generation) of
//    several handles take turns through a lock word IN DEVICE MEMORY, not through the events: an event wait orders a launch behind
//    whatever was recorded when the wait was ENQUEUED, and a forward is enqueued whole -- forward B's first recurrent layer then
//    waits for forward A's LAST one, the forwards' recurrent layers run strictly one forward after the other and B's x-projection
//    GEMMs have nothing to run beside (config 4, round 5: 40.2 ms per batch = the sum of a forward's kernels).  With the lock a
//    stream holds acquire (one wave spinning on an atomic compare-and-swap) -> the persistent launch -> release, so the turn goes to
//    whichever forward's layer is READY: A's layer l + 1 waits for its own GEMM while B's layer l runs.  Events still order the
//    whole-device launches against the slot-sized ones (two models of different widths in one process).  Two forwards in flight
//    is the count that pays: with three, every recurrent launch runs beside the dense kernels of two others and is slower for it
//    (config 4: 4.1 ms per layer against 3.85 beside one and 3.55 alone; 35.0 ms per batch against 33.7 -- also when the third
//    forward is kept out of the turns until one of the two has finished its layers: 35.3; profiles/r06_config4.txt).
// The ring kernel's windows (rnn_persist_ring.hip: H / 32 workgroups per direction, 50 CUs for cfgA) have slots of their own: as many as
// fit the device side by side, at most kRingSlots; a ring launch is ordered behind every launch of the other kernels and vice versa
// (the two families never share the device: the other kernels' grids are sized for halves and quarters of it).
constexpr int kRingSlots = 5;
struct PersistGate { std::mutex mu; hipEvent_t ev[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t ring_ev[kRingSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr};
                     hipEvent_t full_ev = nullptr;      // the whole-device launch recorded last (the slot-sized launches wait for it)
                     unsigned* turn = nullptr;           // device word: 0 free, 1 a whole-device persistent launch is running
                     int ring_cus[kRingSlots] = {0, 0, 0, 0, 0};     // CUs of the window last recorded on each ring slot (it may still be running)
                     int lock_fd = -1; bool lock_tried = false; int next_lane = 0; };

// acquire: one wave spins until it has swapped the word from 0 to 1.  (Bounded: after a second or two it goes on regardless -- two persistent
// kernels that then share the device time out at their hand-offs and their batches are recomputed on the per-step path.)
// (Measured and not kept, profiles/r06_config4.txt: two forwards that alternate at the lock can fall into step -- both in their conv
// layers at the same time, with no recurrent launch to run beside.  Holding a forward's first acquire back until the other is half
// way through its layers keeps them apart, and the stream of batches takes the same time: 35.3 against 35.8 ms per batch over four
// runs each, inside their spread.)
__global__ void turn_acquire_kernel(unsigned* turn) {
    if (threadIdx.x != 0) return;
    unsigned spins = 0;
    while (atomicCAS(turn, 0u, 1u) != 0u && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(32);
}
__global__ void turn_release_kernel(unsigned* turn) {
    if (threadIdx.x == 0) __hip_atomic_store(turn, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE forward at a time runs a dense kernel (the conv stack, an x-projection GEMM) when forwards are in flight (round 6).  Four
// lanes whose GEMMs share the chip fairly all get a quarter of what the ring windows leave, finish together and start their ring
// windows together: the lanes fall into step, and in step the chip alternates between four GEMMs crawling and four windows holding
// 200 CUs at half their MFMA rate with 56 idle.  With a token (a counter in device memory beside the turn lock: a one-wave kernel
// spins until it has taken it, a second gives it back behind the GEMM) a GEMM has everything the windows leave, is done in a
// quarter of the time, and the next lane's follows: the windows start one after the other and stay out of step.  cfgA, 64-clip
// forwards on four lanes: 5.30 -> 5.01 ms per 32-clip batch in steady state, 5.73 -> 5.45 over a 20-batch call; two forwards at
// a time 5.22, three 5.27 (profiles/r06_dense_token.txt).  On only where the caller has given the runtime a hardware queue per
// stream (GPU_MAX_HW_QUEUES >= 8 in the environment): on a shared queue a lane's give-back could stand behind
// another lane's spinning take until that gives up.  DSMI_DENSE_TOKENS=0 turns it off (A/B runs).
__global__ void dense_enter_kernel(unsigned* sem, unsigned limit) {
    if (threadIdx.x != 0) return;
    for (unsigned spins = 0; spins < (1u << 18); ++spins) {
        const unsigned c = __hip_atomic_load(sem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c < limit && atomicCAS(sem, c, c + 1u) == c) return;
        __builtin_amdgcn_s_sleep(8);
    }
    atomicAdd(sem, 1u);      // (gave up waiting: goes on; counted, so that its give-back does not free somebody else's place)
}
__global__ void dense_leave_kernel(unsigned* sem) {
    if (threadIdx.x == 0) atomicSub(sem, 1u);
}
static int dense_tokens() {
    static const int k = [] {
        const char* e = std::getenv("DSMI_DENSE_TOKENS");
        if (e) return std::atoi(e);
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        return q && std::atoi(q) >= 8 ? 1 : 0;
    }();
    return k;
}
static unsigned* dense_sem(int device);
// (experiments: what runs behind the token -- 0 the conv stack as one and every GEMM, 1 the GEMMs only, 2 every conv layer by itself and every GEMM)
static int dense_scope() {
    static const int k = [] { const char* e = exp_env("DSMI_DEBUG_DENSE_SCOPE"); return e ? std::atoi(e) : 0; }();
    return k;
}

// slots a launch of `width` (1, 2 or kMaxLanes) takes for a handle whose home slot is `lane`: [first, first + width)
static int gate_first(int lane, int width) { return width >= kMaxLanes ? 0 : (width == 2 ? 2 * (lane & 1) : (lane % kMaxLanes)); }
// Under g->mu: make stream `s` wait for the slots this launch needs ...
static void gate_wait(PersistGate* g, hipStream_t s, int lane, int width) {
    for (int i = gate_first(lane, width); i < gate_first(lane, width) + width && i < kMaxLanes; ++i)
        if (g->ev[i]) (void)hipStreamWaitEvent(s, g->ev[i], 0);
    for (int i = 0; i < kRingSlots; ++i)
        if (g->ring_ev[i]) (void)hipStreamWaitEvent(s, g->ring_ev[i], 0);
    if (width >= kMaxLanes && g->turn) {      // whole device: behind the slot-sized launches by events, among themselves by the lock
        hipLaunchKernelGGL(turn_acquire_kernel, dim3(1), dim3(64), 0, s, g->turn);
        return;
    }
    if (g->full_ev) (void)hipStreamWaitEvent(s, g->full_ev, 0);
}
// The same for `n` windows of the ring kernel, `cus` CUs each, on ring slots [first, first + n).  Slots are counted per model
// (n_cus / cus of ITS geometry), but they index one set of events per device: two models of different widths in one process
// could otherwise be admitted side by side beyond the device (H = 800: 50 CUs on slots 0..4, H = 896: 56 CUs on slots 0..3 --
// slot 4 at 50 CUs beside four windows at 56 is 274 CUs), and windows that are not all resident spin to their timeout.  So a
// launch also waits for as many OTHER slots as it takes for the windows that may still run plus its own to fit the device.
static void ring_gate_wait(PersistGate* g, hipStream_t s, int first, int n, int cus, int n_cus) {
    int others = 0;
    for (int i = 0; i < kRingSlots; ++i) {
        const bool mine = i >= first && i < first + n;
        if (mine) { if (g->ring_ev[i]) (void)hipStreamWaitEvent(s, g->ring_ev[i], 0); }
        else others += g->ring_cus[i];
    }
    for (int i = 0; i < kRingSlots && others + n * cus > n_cus; ++i) {
        if ((i >= first && i < first + n) || !g->ring_cus[i]) continue;
        if (g->ring_ev[i]) (void)hipStreamWaitEvent(s, g->ring_ev[i], 0);
        others -= g->ring_cus[i];
    }
    for (int i = 0; i < kMaxLanes; ++i)
        if (g->ev[i]) (void)hipStreamWaitEvent(s, g->ev[i], 0);
    if (g->full_ev) (void)hipStreamWaitEvent(s, g->full_ev, 0);
}
static void ring_gate_record(PersistGate* g, hipStream_t s, int first, int n, int cus) {
    for (int i = first; i < first + n && i < kRingSlots; ++i) {
        if (g->ring_ev[i]) (void)hipEventRecord(g->ring_ev[i], s);
        g->ring_cus[i] = cus;
    }
}
// ... and publish the launch on them.
static void gate_record(PersistGate* g, hipStream_t s, int lane, int width) {
    if (width >= kMaxLanes && g->turn) {
        hipLaunchKernelGGL(turn_release_kernel, dim3(1), dim3(64), 0, s, g->turn);
        if (g->full_ev) (void)hipEventRecord(g->full_ev, s);
        return;
    }
    for (int i = gate_first(lane, width); i < gate_first(lane, width) + width && i < kMaxLanes; ++i)
        if (g->ev[i]) (void)hipEventRecord(g->ev[i], s);
}
static PersistGate* persist_gate(int device) {
    static std::mutex mu;
    static std::map<int, PersistGate*> gates;
    std::lock_guard<std::mutex> lk(mu);
    auto it = gates.find(device);
    if (it != gates.end()) return it->second;
    PersistGate* g = new PersistGate();
    for (int i = 0; i < kMaxLanes; ++i)
        if (hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming) != hipSuccess) g->ev[i] = nullptr;
    for (int i = 0; i < kRingSlots; ++i)
        if (hipEventCreateWithFlags(&g->ring_ev[i], hipEventDisableTiming) != hipSuccess) g->ring_ev[i] = nullptr;
    if (hipEventCreateWithFlags(&g->full_ev, hipEventDisableTiming) != hipSuccess) g->full_ev = nullptr;
    // (without the word, or with DSMI_PERSIST_TURNS=events, the whole-device launches chain through the events as before: A/B runs)
    const char* turns = std::getenv("DSMI_PERSIST_TURNS");
    if (!(turns && std::string(turns) == "events") && g->full_ev) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&g->turn, 2 * sizeof(unsigned)) != hipSuccess ||          // [1]: the dense kernels' token
            hipMemset(g->turn, 0, 2 * sizeof(unsigned)) != hipSuccess) g->turn = nullptr;
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
    gates[device] = g;
    return g;
}

static unsigned* dense_sem(int device) {
    PersistGate* g = persist_gate(device);
    return g->turn ? g->turn + 1 : nullptr;
}

// true when this process may run persistent kernels on `device`
static bool persist_process_lock(int device) {
    const char* shared = std::getenv("DSMI_PERSIST_SHARED");
    if (shared && shared[0] == '1') return true;
    PersistGate* g = persist_gate(device);
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->lock_tried) return g->lock_fd >= 0 || g->lock_fd == -2;
    g->lock_tried = true;
    char bus[64] = "unknown";
    (void)hipDeviceGetPCIBusId(bus, sizeof(bus), device);
    for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
    const char* tmp = std::getenv("TMPDIR");
    const std::string path = std::string(tmp && tmp[0] ? tmp : "/tmp") + "/dsmi-persist-" + bus + ".lock";
    const int fd = open(path.c_str(), O_CREAT | O_RDWR, 0666);
    if (fd < 0) { g->lock_fd = -2; return true; }          // no lock directory: nothing to arbitrate with
    if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); g->lock_fd = -1; return false; }
    g->lock_fd = fd;                                        // held until the process exits
    return true;
}

constexpr float kF16Safe = 60000.f;    // below fp16's 65504 with room for rounding

static int fail(dsmi_model* m, int code, const std::string& msg) {
    m->err = msg;
    return code;
}

// ------------------------------------------------------------------------------------------
extern "C" int dsmi_model_create(const dsmi_model_desc* d, int device, dsmi_model** out) {
    if (!d || !out) { g_create_error = "null argument"; return DSMI_ERR_INVALID; }
    // reference model.py:344-348
    if (d->conv_layers == 0) { g_create_error = "0 convolutional layers configuration not supported by DanSpeech"; return DSMI_ERR_CONV; }
    if (d->conv_layers > 3 || d->conv_layers < 0) { g_create_error = "Maximum amount of convolutional layers supported by DanSpeech is 3"; return DSMI_ERR_CONV; }
    if (d->rnn_type < 0 || d->rnn_type > 2 || d->rnn_hidden_size < 1 || d->rnn_layers < 1 || d->n_labels < 1 ||
        d->n_labels > 128 || (!d->bidirectional && d->context < 1)) {
        g_create_error = "invalid model description";
        return DSMI_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "no such HIP device";
        return DSMI_ERR_HIP;
    }
    dsmi_model* m = new dsmi_model();
    m->desc = *d;
    m->device = device;
    m->n_fft = (int)(d->sample_rate * d->window_size);      // model.py:354: int(floor(rate * size / 2) + 1) below
    m->n_freq = m->n_fft / 2 + 1;                            // model.py:354
    int f = m->n_freq;
    for (int l = 0; l < d->conv_layers; ++l) {
        const ConvSpec& s = kConvSpecs[l];
        m->conv_fi[l] = f;
        f = (f + 2 * s.pf - s.kf) / s.sf + 1;
        m->conv_fo[l] = f;
    }
    m->I0 = kConvSpecs[d->conv_layers - 1].co * f;           // model.py:365,379,396
    m->geom = make_rnn_geom(d->rnn_type, d->rnn_hidden_size, d->bidirectional ? 2 : 1);
    m->Hs = m->geom.Kp;
    m->geom16 = make_rnn_geom_u(d->rnn_type, d->rnn_hidden_size, d->bidirectional ? 2 : 1, 16);
    m->have16 = d->rnn_hidden_size % 16 == 0;
    m->rnn.resize(d->rnn_layers);
    {
        hipDeviceProp_t prop;
        m->n_cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 0;
        const char* mode = std::getenv("DSMI_RNN_MODE");      // "steps": one launch per time step; "persist8": first-generation persistent kernel
        m->rnn_mode = (mode && std::string(mode) == "steps") ? 0 : 1;
        m->persist_gen = (mode && std::string(mode) == "persist8") ? 1 : 2;
        // DSMI_RNN_KERNEL=duo: the paired-tile / tile-walking kernels where the ring kernel would run (A/B measurements and the
        // parity tests of those kernels); =ring: the ring kernel also for a lone batch of up to 32 clips
        const char* rk = std::getenv("DSMI_RNN_KERNEL");
        m->rnn_kernel = (rk && std::string(rk) == "duo") ? 1 : ((rk && std::string(rk) == "ring") ? 2 : 0);
        m->ring8 = rk && std::string(rk) == "ring8";      // the eight-wave form of the ring kernel everywhere (A/B runs)
        m->ring4 = rk && std::string(rk) == "ring4";      // the four-wave form everywhere (also for windows of one or two tiles)
        // DSMI_DENSE_MODE=f32: GEMM and conv layers on the plain fp32-MFMA kernels (the round-1 path, and where a model whose
        // weights leave fp16's range ends up by itself); with DSMI_RNN_MODE=steps the whole forward is the second, independent
        // implementation the parity tests compare the default one with
        const char* dm = std::getenv("DSMI_DENSE_MODE");
        const bool dense_f32 = dm && std::string(dm) == "f32";
        m->gemm_mode = dense_f32 ? 0 : 1;
        m->conv_mode = dense_f32 ? 0 : 1;
        m->conv1_split = !dense_f32;
        // test hooks for the hand-off timeout path (tests/test_gpu_timeout.py)
        if (const char* sl = std::getenv("DSMI_DEBUG_SPIN_LIMIT")) m->spin_limit = (unsigned)std::max(1L, std::atol(sl));
        if (const char* ds = std::getenv("DSMI_DEBUG_DROP_SIGNAL"))
            if (std::sscanf(ds, "%d:%d:%d", &m->drop_layer, &m->drop_wg, &m->drop_step) != 3) m->drop_layer = -1;
        m->lanes = 2;
        {
            PersistGate* g = persist_gate(device);
            std::lock_guard<std::mutex> lk(g->mu);
            m->lane = g->next_lane++;
        }
        if (m->rnn_mode == 1 && !persist_process_lock(device)) {
            m->rnn_mode = 0;
            m->err = "another process holds this GPU's persistent-kernel lock: recurrent layers run one launch per step";
        }
    }
    *out = m;
    return DSMI_OK;
}

extern "C" const char* dsmi_last_error(const dsmi_model* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

extern "C" int dsmi_model_info(const dsmi_model* m, dsmi_model_desc* desc, int* device) {
    if (!m) return DSMI_ERR_INVALID;
    if (desc) *desc = m->desc;
    if (device) *device = m->device;
    return DSMI_OK;
}

extern "C" int dsmi_model_load_tensor(dsmi_model* m, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!m || !name || !data || ndim < 0 || ndim > 4) return m ? fail(m, DSMI_ERR_INVALID, "bad tensor argument") : DSMI_ERR_INVALID;
    if (m->finalized) return fail(m, DSMI_ERR_INVALID, "model already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    // LookaheadStream is a direct attribute of a streaming model (model.py:490), the first module of a
    // Sequential otherwise (model.py:407-411): one weight, two names
    const std::string key = std::string(name) == "lookahead.conv.weight" ? "lookahead.0.conv.weight" : name;
    m->tensors[key] = std::move(t);
    return DSMI_OK;
}

static const HostTensor* need(dsmi_model* m, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = m->tensors.find(name);
    if (it == m->tensors.end()) { m->err = "missing tensor " + name; return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) { m->err = "bad shape for tensor " + name; return nullptr; }
    return &it->second;
}

template <typename T>
static int upload(dsmi_model* m, const std::vector<T>& h, T** dev) {
    HIP_OK(m, hipMalloc((void**)dev, std::max<size_t>(h.size(), 1) * sizeof(T)));
    m->owned.push_back(*dev);
    if (!h.empty()) HIP_OK(m, hipMemcpy(*dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return DSMI_OK;
}

// eval-mode BatchNorm as y = x*a + b, computed as ATen's CPU kernel does:
// invstd = 1/sqrt(var+eps); a = weight*invstd; b = bias - mean*a.
static bool bn_affine(dsmi_model* m, const std::string& prefix, int n, int n_pad, std::vector<float>& a, std::vector<float>& b) {
    const HostTensor *w = need(m, prefix + ".weight", {n}), *bi = need(m, prefix + ".bias", {n}),
                     *mu = need(m, prefix + ".running_mean", {n}), *var = need(m, prefix + ".running_var", {n});
    if (!w || !bi || !mu || !var) return false;
    a.assign(n_pad, 0.f); b.assign(n_pad, 0.f);
    for (int i = 0; i < n; ++i) {
        const float invstd = 1.f / std::sqrt(var->data[i] + 1e-5f);
        a[i] = w->data[i] * invstd;
        b[i] = bi->data[i] - mu->data[i] * a[i];
    }
    return true;
}

extern "C" int dsmi_model_finalize(dsmi_model* m) {
    if (!m) return DSMI_ERR_INVALID;
    if (m->finalized) return DSMI_OK;
    HIP_OK(m, hipSetDevice(m->device));
    const dsmi_model_desc& d = m->desc;
    const RnnGeom& g = m->geom;
    const int H = d.rnn_hidden_size, G = g.G;
    // ---- conv stack
    for (int l = 0; l < d.conv_layers; ++l) {
        const ConvSpec& s = kConvSpecs[l];
        const std::string p = "conv.seq_module." + std::to_string(3 * l);
        const HostTensor* w = need(m, p + ".weight", {s.co, s.ci, s.kf, s.kt});
        const HostTensor* b = need(m, p + ".bias", {s.co});
        std::vector<float> a, bb;
        if (!w || !b || !bn_affine(m, "conv.seq_module." + std::to_string(3 * l + 1), s.co, s.co, a, bb)) return DSMI_ERR_NOT_READY;
        int rc;
        if ((rc = upload(m, pack_conv_weights(w->data.data(), l), &m->conv[l].wp))) return rc;
        for (float v : w->data) if (!(std::fabs(v) < kF16Safe / 64.f)) m->conv_mode = 0;    // split-fp16 operand range (packed times 2^6: conv_split.hip)
        if (l > 0) {
            if ((rc = upload(m, pack_conv_w_split(w->data.data(), s.co), &m->conv[l].wp_sp))) return rc;
        } else {
            if ((rc = upload(m, pack_conv1_w_split(w->data.data()), &m->conv[l].wp_sp))) return rc;
        }
        if ((rc = upload(m, b->data, &m->conv[l].bias))) return rc;
        if ((rc = upload(m, a, &m->conv[l].bn_a))) return rc;
        if ((rc = upload(m, bb, &m->conv[l].bn_b))) return rc;
    }
    // ---- recurrent layers
    for (int l = 0; l < d.rnn_layers; ++l) {
        RnnW& r = m->rnn[l];
        const int I = l == 0 ? m->I0 : H;
        r.K = l == 0 ? I : m->Hs;
        r.ldw = round_up(r.K, 4);
        std::vector<float> wih((size_t)g.Np * r.ldw, 0.f), bih(g.Np, 0.f);
        const std::string p = "rnns." + std::to_string(l) + ".rnn.";
        const HostTensor *wi[2], *wh[2], *bi[2], *bh[2];
        for (int dd = 0; dd < g.D; ++dd) {
            const std::string sfx = dd ? "_reverse" : "";
            wi[dd] = need(m, p + "weight_ih_l0" + sfx, {G * H, I});
            wh[dd] = need(m, p + "weight_hh_l0" + sfx, {G * H, H});
            bi[dd] = need(m, p + "bias_ih_l0" + sfx, {G * H});
            bh[dd] = need(m, p + "bias_hh_l0" + sfx, {G * H});
            if (!wi[dd] || !wh[dd] || !bi[dd] || !bh[dd]) return DSMI_ERR_NOT_READY;
        }
        for (int col = 0; col < g.Np; ++col) {
            int dd;
            const int src = rnn_src_row(g, col, &dd);
            if (src < 0) continue;
            std::memcpy(&wih[(size_t)col * r.ldw], &wi[dd]->data[(size_t)src * I], sizeof(float) * I);
            bih[col] = bi[dd]->data[src];
        }
        int rc;
        for (float v : wih) if (!(std::fabs(v) < kF16Safe / 64.f)) m->gemm_mode = 0;     // split-fp16 operand range (packed times 2^6: gemm.hip)
        if ((rc = upload(m, wih, &r.wih))) return rc;
        if ((rc = upload(m, pack_gemm_w_split(wih.data(), g.Np, r.K, r.ldw), &r.wih_sp))) return rc;
        if ((rc = upload(m, bih, &r.bih))) return rc;
        for (int dd = 0; dd < g.D; ++dd) {
            // the split-fp16 operands hold |x| < 65504 only: a model beyond that stays on the fp32 kernels
            for (float v : wh[dd]->data) if (!(std::fabs(v) < kF16Safe)) m->rnn_mode = 0;
            if ((rc = upload(m, pack_whh(g, wh[dd]->data.data()), &r.whh[dd]))) return rc;
            if ((rc = upload(m, pack_whh_split(g, wh[dd]->data.data()), &r.whh_sp[dd]))) return rc;
            if ((rc = upload(m, bh[dd]->data, &r.bhh[dd]))) return rc;
        }
        if (m->have16) {
            const RnnGeom& g16 = m->geom16;
            std::vector<float> w16((size_t)g16.Np * r.ldw, 0.f), b16(g16.Np, 0.f);
            for (int col = 0; col < g16.Np; ++col) {
                int dd;
                const int src = rnn_src_row(g16, col, &dd);
                if (src < 0) continue;
                std::memcpy(&w16[(size_t)col * r.ldw], &wi[dd]->data[(size_t)src * I], sizeof(float) * I);
                b16[col] = bi[dd]->data[src];
            }
            if ((rc = upload(m, pack_gemm_w_split(w16.data(), g16.Np, r.K, r.ldw), &r.wih16_sp))) return rc;
            if ((rc = upload(m, b16, &r.bih16))) return rc;
            for (int dd = 0; dd < g.D; ++dd)
                if ((rc = upload(m, pack_whh16(g16, wh[dd]->data.data()), &r.whh16_sp[dd]))) return rc;
        }
        if (l > 0) {  // model.py:403-404: BatchNorm1d(H) in front of layers >= 1
            std::vector<float> a, b;
            if (!bn_affine(m, "rnns." + std::to_string(l) + ".batch_norm.module", H, m->Hs, a, b)) return DSMI_ERR_NOT_READY;
            // the GEMM's A operand is (h_fwd + h_bwd) * a + b with |h| <= 1: bounded by 2|a| + |b|
            for (size_t k = 0; k < a.size(); ++k) if (!(2.f * std::fabs(a[k]) + std::fabs(b[k]) < kF16Safe)) m->gemm_mode = 0;
            if ((rc = upload(m, a, &r.bn_a))) return rc;
            if ((rc = upload(m, b, &r.bn_b))) return rc;
        }
    }
    int rc;
    if (!d.bidirectional) {
        const HostTensor* lw = need(m, "lookahead.0.conv.weight", {H, 1, d.context});
        if (!lw) return DSMI_ERR_NOT_READY;
        if ((rc = upload(m, lw->data, &m->look_w))) return rc;
    }
    {
        std::vector<float> a, b;
        const HostTensor* fw = need(m, "fc.0.module.1.weight", {d.n_labels, H});
        if (!fw || !bn_affine(m, "fc.0.module.0", H, m->Hs, a, b)) return DSMI_ERR_NOT_READY;
        if ((rc = upload(m, a, &m->fc_a))) return rc;
        if ((rc = upload(m, b, &m->fc_b))) return rc;
        if ((rc = upload(m, pack_fc(fw->data.data(), d.n_labels, H), &m->fc_wp))) return rc;
    }
    m->tensors.clear();
    for (int i = 0; i < 8; ++i) HIP_OK(m, hipEventCreate(&m->ev[i]));
    // every slot of the forward-status ring now, not on the first four forwards (a pinned allocation can take ~100 ms)
    for (auto& f : m->fwd) {
        if (f.done) continue;
        HIP_OK(m, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
        HIP_OK(m, hipHostMalloc((void**)&f.err_host, sizeof(unsigned), hipHostMallocDefault));
        *f.err_host = 0;
    }
    m->finalized = true;
    return DSMI_OK;
}

static int seq_len(const dsmi_model* m, int L) {  // model.py:540-551
    for (int l = 0; l < m->desc.conv_layers; ++l) {
        const ConvSpec& s = kConvSpecs[l];
        L = (L + 2 * s.pt - (s.kt - 1) - 1) / s.st + 1;
    }
    return L;
}

extern "C" int dsmi_seq_lens(const dsmi_model* m, const int32_t* lens, int n, int32_t* out) {
    if (!m || !lens || !out) return DSMI_ERR_INVALID;
    for (int i = 0; i < n; ++i) out[i] = seq_len(m, lens[i]);
    return DSMI_OK;
}

static void free_ws(dsmi_model* m) {
    for (void* p : m->ws) (void)hipFree(p);
    m->ws.clear();
    m->cap_B = m->cap_T = 0;
}

template <typename T>
static int ws_alloc(dsmi_model* m, T** p, size_t n) {
    HIP_OK(m, hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    m->ws.push_back(*p);
    return DSMI_OK;
}

extern "C" int dsmi_reserve(dsmi_model* m, int max_B, int max_T) {
    if (!m || max_B < 1 || max_T < 1) return m ? fail(m, DSMI_ERR_INVALID, "bad reserve size") : DSMI_ERR_INVALID;
    if (!m->finalized) return fail(m, DSMI_ERR_NOT_READY, "dsmi_model_finalize has not been called");
    if (max_B <= m->cap_B && max_T <= m->cap_T) return DSMI_OK;
    HIP_OK(m, hipSetDevice(m->device));
    HIP_OK(m, hipDeviceSynchronize());
    max_B = std::max(max_B, m->cap_B);
    max_T = std::max(max_T, m->cap_T);
    free_ws(m);
    const dsmi_model_desc& d = m->desc;
    const int To = seq_len(m, max_T);
    const int ys = round_up(std::max(To, 1), 4);
    size_t conv_max = 0;
    for (int l = 0; l < d.conv_layers; ++l)
        conv_max = std::max(conv_max, (size_t)max_B * kConvSpecs[l].co * m->conv_fo[l] * ys);
    int rc;
    if ((rc = ws_alloc(m, &m->conv_buf[0], conv_max))) return rc;
    if ((rc = ws_alloc(m, &m->conv_buf[1], d.conv_layers > 1 ? conv_max : 1))) return rc;
    for (int i = 0; i < 2; ++i) {   // split intermediates: layer 0 -> buf3[0], layer 1 -> buf3[1] (3-conv models)
        const size_t n = i < d.conv_layers - 1 ? (size_t)max_B * m->conv_fo[i] * 2 * std::max(To, 1) * 32 : 1;
        if ((rc = ws_alloc(m, &m->conv_buf_sp[i], n))) return rc;
    }
    const size_t rows = (size_t)To * max_B;
    if ((rc = ws_alloc(m, &m->xp, rows * std::max(m->geom.Np, m->have16 ? m->geom16.Np : 0)))) return rc;
    for (int i = 0; i < 2; ++i)
        for (int dd = 0; dd < 2; ++dd) {
            m->hbuf[i][dd] = nullptr;
            if (dd < m->geom.D && (rc = ws_alloc(m, &m->hbuf[i][dd], rows * m->Hs))) return rc;
        }
    for (int dd = 0; dd < 2; ++dd) {
        m->cst[dd] = nullptr;
        if (d.rnn_type == DSMI_RNN_LSTM && dd < m->geom.D && (rc = ws_alloc(m, &m->cst[dd], (size_t)max_B * m->Hs))) return rc;
    }
    {
        // zero once: operand slots of padding units (k in [H, Hs)) are never written and must stay finite
        const size_t n = (size_t)2 * m->geom.D * ceil_div(max_B, 32) * m->geom.nq * 256;
        if ((rc = ws_alloc(m, &m->hpack, n))) return rc;
        HIP_OK(m, hipMemset(m->hpack, 0, n * sizeof(float)));
    }
    {
        const size_t n = (size_t)2 * m->geom.D * ceil_div(max_B, 32) * ceil_div(m->geom.nq, 2) * 2 * 64 * 8;
        if ((rc = ws_alloc(m, &m->hpack_sp, n))) return rc;
        HIP_OK(m, hipMemset(m->hpack_sp, 0, n * sizeof(uint16_t)));
    }
    {
        const size_t mt = (size_t)max_B * ceil_div(std::max(To, 1), 128) + ceil_div((int)rows, 128) + 1;
        const size_t kt = (size_t)ceil_div(std::max(m->I0, m->Hs), 32);
        if ((rc = ws_alloc(m, &m->a_sp, mt * kt * 2 * 4096))) return rc;
    }
    if (m->have16) {
        const size_t n = rnn_persist16_state_halfs(m->geom16, max_B);
        if ((rc = ws_alloc(m, &m->hpack16, n))) return rc;
        HIP_OK(m, hipMemset(m->hpack16, 0, n * sizeof(uint16_t)));
    }
    // (+ the ring kernel's direction tickets behind the counters: two words per window, zeroed by the same memset)
    if ((rc = ws_alloc(m, &m->pcnt, (size_t)m->geom.D * ceil_div(max_B, 16) * std::max(To, 1) * kPersist16CntWords + 2 * ceil_div(max_B, 16) + 2))) return rc;
    if ((rc = ws_alloc(m, &m->perr, (size_t)4))) return rc;
    HIP_OK(m, hipMemset(m->perr, 0, 4 * sizeof(unsigned)));
    m->look_buf = nullptr;
    if (!d.bidirectional && (rc = ws_alloc(m, &m->look_buf, rows * m->Hs))) return rc;
    if ((rc = ws_alloc(m, &m->xin, rows * round_up(std::max(m->I0, m->Hs), 4)))) return rc;
    if ((rc = ws_alloc(m, &m->lens_dev, (size_t)max_B))) return rc;
    if ((rc = ws_alloc(m, &m->sizes_dev, (size_t)max_B))) return rc;
    if ((rc = ws_alloc(m, &m->raw_ids, rows))) return rc;
    if ((rc = ws_alloc(m, &m->ids, rows))) return rc;
    if ((rc = ws_alloc(m, &m->offs, rows))) return rc;
    if ((rc = ws_alloc(m, &m->nout, (size_t)max_B))) return rc;
    m->cap_B = max_B;
    m->cap_T = max_T;
    return DSMI_OK;
}

extern "C" void dsmi_model_destroy(dsmi_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    free_ws(m);
    for (void* p : m->owned) (void)hipFree(p);
    if (m->finalized)
        for (int i = 0; i < 8; ++i) (void)hipEventDestroy(m->ev[i]);
    timer_resolve(m);
    for (auto& f : m->fwd) {
        if (f.err_host) (void)hipHostFree(f.err_host);
        if (f.done) (void)hipEventDestroy(f.done);
    }
    if (m->lens_stage) (void)hipHostFree(m->lens_stage);
    for (hipEvent_t e : m->stage_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->kt.free_events) (void)hipEventDestroy(e);
    delete m;
}

// ------------------------------------------------------------------------------------------
static int check_batch(dsmi_model* m, const int32_t* lens, int B, int T) {
    if (!m->finalized) return fail(m, DSMI_ERR_NOT_READY, "dsmi_model_finalize has not been called");
    if (!lens || B < 1 || T < 1) return fail(m, DSMI_ERR_INVALID, "bad batch arguments");
    for (int i = 0; i < B; ++i) {
        if (lens[i] < 1 || lens[i] > T) return fail(m, DSMI_ERR_INVALID, "length outside 1..T");
        // pack_padded_sequence(enforce_sorted=True), model.py:117
        if (i && lens[i] > lens[i - 1]) return fail(m, DSMI_ERR_UNSORTED, "`lengths` array must be sorted in decreasing order");
    }
    return DSMI_OK;
}

static int run_conv(dsmi_model* m, const float* feat, int B, int T, int To, int ys, hipStream_t s, const float** out) {
    const float* x = feat;
    const uint16_t* x_sp = nullptr;
    int ti = T, xs = T;
    const int L = m->desc.conv_layers;
    for (int l = 0; l < L; ++l) {
        const ConvSpec& sp = kConvSpecs[l];
        double fl = 0;
        for (int i = 0; i < B; ++i) fl += 2.0 * sp.co * m->conv_fo[l] * (double)m->host_out_lens[i] * sp.ci * sp.kf * sp.kt;
        const double by = 4.0 * B * ((double)sp.ci * m->conv_fi[l] * ti + (double)sp.co * m->conv_fo[l] * To);
        const bool next_split = m->conv_mode == 1 && l + 1 < L;       // the consumer is a split-fp16 conv layer
        unsigned* lsem = (dense_tokens() > 0 && m->inflight >= 2 && dense_scope() == 2) ? dense_sem(m->device) : nullptr;
        if (lsem) hipLaunchKernelGGL(dense_enter_kernel, dim3(1), dim3(64), 0, s, lsem, (unsigned)dense_tokens());
        if (x_sp) {
            ConvSplitLaunch c;
            c.x_sp = x_sp; c.wp_sp = m->conv[l].wp_sp; c.bias = m->conv[l].bias; c.bn_a = m->conv[l].bn_a; c.bn_b = m->conv[l].bn_b;
            c.out_lens_dev = m->lens_dev; c.y = next_split ? nullptr : m->conv_buf[l & 1]; c.y_sp = next_split ? m->conv_buf_sp[l] : nullptr;
            c.B = B; c.co = sp.co; c.fi = m->conv_fi[l]; c.fo = m->conv_fo[l]; c.ti = ti; c.to = To; c.ys = ys;
            c.ev = timer_arm(m, KK_CONV1 + l, true, fl, by);
            launch_conv_split(c, s);
            x = c.y; x_sp = c.y_sp;
        } else {
            ConvLaunch c;
            c.x = x; c.y = m->conv_buf[l & 1]; c.wp = m->conv[l].wp; c.bias = m->conv[l].bias;
            c.bn_a = m->conv[l].bn_a; c.bn_b = m->conv[l].bn_b; c.out_lens_dev = m->lens_dev;
            c.B = B; c.ci = sp.ci; c.co = sp.co; c.fi = m->conv_fi[l]; c.fo = m->conv_fo[l];
            c.ti = ti; c.to = To; c.xs = xs; c.ys = ys; c.layer = l;
            c.y_sp = next_split ? m->conv_buf_sp[l] : nullptr;
            c.ev = timer_arm(m, KK_CONV1 + l, true, fl, by);
            // layer 0 on the split-fp16 MFMA (features are z-normalised log magnitudes: far inside fp16's range)
            if (l == 0 && m->conv_mode == 1 && m->conv1_split) launch_conv1_split(c, m->conv[0].wp_sp, s);
            else launch_conv(c, s);
            x = c.y; x_sp = c.y_sp;
        }
        if (lsem) hipLaunchKernelGGL(dense_leave_kernel, dim3(1), dim3(64), 0, s, lsem);
        ti = To; xs = ys;
    }
    *out = x;
    return DSMI_OK;
}

// One BatchRNN layer on the internal buffers: x-projection GEMM + To recurrent step launches.
static void run_rnn_layer(dsmi_model* m, int l, GemmLaunch gl, int B, int To, int dst, hipStream_t s) {
    double sumlen = 0;
    for (int i = 0; i < B; ++i) sumlen += m->host_out_lens[i];
    const double GH = (double)m->geom.G * m->desc.rnn_hidden_size, Dd = m->geom.D;
    int pgroups = 0, waves = 8;
    bool use16 = m->rnn_mode == 1 && m->persist_gen == 2 && m->gemm_mode == 1 && m->have16 && gl.w_sp;
    // Which 16-unit kernel.  The caller says how many batches it keeps in flight (dsmi_model_set_inflight):
    //   1 -> whole-CU workgroups on the whole device: the shortest step for a lone batch (2.7 us for cfgA at B = 32);
    //   2 -> the paired-tile pipeline on ONE gate lane's CUs when the batch fits there (B = 17..32 for cfgA: 100 CUs), so that
    //        the second batch's recurrent layer runs on the other half of the chip; failing that, half-CU workgroups (one lane,
    //        the two batches share every CU); failing that, whole-CU workgroups (both lanes: the two batches take turns).
    bool duo = false, duo_lane = false;
    if (use16 && m->inflight >= 2) {
        duo_lane = rnn_persist_duo_eligible(m->geom16, B, m->n_cus / 2);
        duo = duo_lane || rnn_persist_duo_eligible(m->geom16, B, m->n_cus);
    }
    // Batches of more than one tile pair (B > 32): the paired-tile kernel in WINDOWS of as many tile pairs as the device holds,
    // one launch after the other (batches in flight take turns): 3.8 us per step and window against 2.3-2.9 us per 32 clips for
    // the kernel that walks the tiles.  (One pair per launch on the handle's own lane, the other batch's windows beside it, was
    // measured and is worse -- config 5: 169 against 128 ms per batch: four times as many persistent launches, each of which
    // waits for whole free CUs behind the other batch's small dense workgroups.)
    int duo_window = 0;
    if (use16 && !duo_lane && B > 32) {
        duo_window = rnn_persist_duo_pairs(m->geom16, B, m->n_cus);
        duo = duo_window >= 1;
    }
    // The ring kernel (rnn_persist_ring.hip): a window = every tile of up to 4..8 on H / 32 workgroups per direction (cfgA: 50 CUs,
    // ONE gate slot).  With batches in flight a handle's layer is one window on its own slot -- the other slots and the rest of
    // the chip belong to the other batches; a lone batch of more than 32 clips spreads its tiles over as many windows side by
    // side as the device holds.  (A lone batch of up to 32 clips keeps the whole-device kernel: the shortest step.)
    int ring_ntw = 0, ring_nwin = 0, ring_slots = 0;
    bool ring_only8 = m->ring8;      // the eight-wave form on every window: asked for, or a shape the four-wave form does not take
    if (use16 && m->rnn_kernel != 1 && (m->inflight >= 2 || B > 32 || m->rnn_kernel == 2)) {
        const int rcus = rnn_persist_ring_cus(m->geom16);
        ring_slots = rcus > 0 ? std::min(kRingSlots, m->n_cus / rcus) : 0;        // windows the device holds side by side
        static const int slot_cap = [] { const char* e = exp_env("DSMI_DEBUG_RING_SLOTS"); return e ? std::atoi(e) : 0; }();      // (experiments)
        if (slot_cap >= 2 && ring_slots > slot_cap) ring_slots = slot_cap;
        const int cap4 = m->ring8 ? 0 : rnn_persist_ring4_tiles(m->geom16, B, rcus);
        ring_only8 = ring_only8 || cap4 == 0;
        const int cap = ring_slots >= 2 ? (ring_only8 ? rnn_persist_ring_tiles(m->geom16, B, rcus) : cap4) : 0;
        if (cap > 0) {
            const int ntiles = ceil_div(B, 16);
            // windows side by side: one with batches in flight, up to four for a lone batch -- or what the caller said
            // (dsmi_model_set_ring_windows: two where only two forwards will share the chip)
            const int slots = m->ring_windows > 0 ? std::min(std::min(m->ring_windows, ring_slots), kMaxLanes)
                                                  : (m->inflight >= 2 ? 1 : std::min(ring_slots, kMaxLanes));
            ring_ntw = std::min(std::max(ceil_div(ntiles, slots), 1), cap);
            if (m->inflight < 2 && m->ring_windows <= 0) ring_ntw = std::max(ring_ntw, std::min(ntiles, 2));
            ring_nwin = std::min(ceil_div(ntiles, ring_ntw), slots);
            duo = false;
        }
    }
    if (use16 && !duo && !ring_ntw) {
        if (m->inflight >= 2 && rnn_persist16_half_eligible(m->geom16, B, m->n_cus, &pgroups)) waves = 4;
        else use16 = rnn_persist16_eligible(m->geom16, B, m->n_cus, &pgroups);
    }
    if (use16) {      // the second-generation kernel reads the x-projection in its own column order
        gl.w_sp = m->rnn[l].wih16_sp; gl.bias = m->rnn[l].bih16; gl.N = m->geom16.Np; gl.ldc = m->geom16.Np;
    }
    gl.ev = timer_arm(m, gl.mode == GEMM_A_CONV ? KK_GEMM0 : KK_GEMM, true, 2.0 * Dd * GH * gl.K * sumlen,
                      4.0 * ((double)gl.M * gl.K * (gl.a2 ? 2 : 1) + (double)gl.N * gl.K + (double)gl.M * gl.N));
    unsigned* sem = (dense_tokens() > 0 && m->inflight >= 2) ? dense_sem(m->device) : nullptr;      // one forward's dense kernel at a time
    if (sem) hipLaunchKernelGGL(dense_enter_kernel, dim3(1), dim3(64), 0, s, sem, (unsigned)dense_tokens());
    launch_gemm(gl, s);
    if (sem) hipLaunchKernelGGL(dense_leave_kernel, dim3(1), dim3(64), 0, s, sem);
    if (use16) {
        RnnPersist16Launch pl;
        pl.g = m->geom16;
        for (int dd = 0; dd < 2; ++dd) {
            pl.whh16[dd] = m->rnn[l].whh16_sp[dd];
            pl.bhh[dd] = m->rnn[l].bhh[dd]; pl.out[dd] = m->hbuf[dst][dd];
        }
        pl.xp = m->xp; pl.lens_dev = m->lens_dev; pl.hpack16 = m->hpack16; pl.counters = m->pcnt; pl.err = m->perr;
        pl.B = B; pl.T = To; pl.pgroups = pgroups; pl.waves = waves;
        pl.spin_limit = m->spin_limit;
        if (m->drop_layer == l) { pl.drop_wg = m->drop_wg; pl.drop_step = m->drop_step; }
        const size_t cnt_words = (size_t)m->geom.D * ceil_div(B, 16) * To * kPersist16CntWords;
        (void)hipMemsetAsync(m->pcnt, 0, sizeof(unsigned) * (cnt_words + 2 * ceil_div(B, 16) + 2), s);
        // the four-wave ring kernel's directions by XCD half (its prologue; profiles/r06_ring_experiments.txt: 2.29 -> 1.56 GB fetched per
        // launch); DSMI_DEBUG_RING_XCD=0 in the experiments build: by blockIdx
        static const bool ring_xcd = [] { const char* e = exp_env("DSMI_DEBUG_RING_XCD"); return !(e && e[0] == '0'); }();
        const int total_pairs = (ceil_div(B, 16) + 1) / 2;
        const int window = duo && duo_window > 0 ? duo_window : total_pairs;
        bool ok = true;
        const int ntiles = ceil_div(B, 16);
        for (int t0 = 0; ring_ntw && t0 < ntiles && ok; t0 += ring_ntw * ring_nwin) {
            const int nw = std::min(ring_nwin, ceil_div(ntiles - t0, ring_ntw));
            const double part = (double)std::min(ring_ntw * nw, ntiles - t0) / ntiles;
            pl.tile0 = t0; pl.ntw = ring_ntw; pl.nwin = nw;
            pl.tickets = ring_xcd ? m->pcnt + cnt_words + 2 * (t0 / ring_ntw) : nullptr;
            pl.ev = timer_arm(m, KK_PERSIST, true, part * 2.0 * Dd * GH * m->desc.rnn_hidden_size * sumlen,
                              part * 4.0 * Dd * (GH * m->desc.rnn_hidden_size + (double)To * B * (GH + 2.0 * m->desc.rnn_hidden_size)));
            PersistGate* gate = persist_gate(m->device);
            std::lock_guard<std::mutex> lk(gate->mu);
            // a handle's own slot; a PAIR of windows with batches in flight: the handle's own pair of slots (consecutive forwards run on
            // consecutive handles: their pairs differ); a lone batch's windows: from slot 0
            const int first = nw == 1 ? m->lane % ring_slots : ((nw == 2 && m->inflight >= 2 && ring_slots >= 4) ? 2 * (m->lane & 1) : 0);
            const int rcus = rnn_persist_ring_cus(m->geom16);
            ring_gate_wait(gate, s, first, nw, rcus, m->n_cus);
            // Which form of the ring kernel.  Four waves (one per SIMD, the cell in the MFMAs' shadows) where a window walks three
            // tiles or more: 6.4 against 7.2 us per step of four tiles (cfgA, alone on the chip).  The four-wave form multiplies
            // phantom tiles like real ones (its phase has no branch), the eight-wave form skips them: a window of one or two tiles
            // -- a lone 32-clip batch at the end of a stream -- is 4.3 / 5.4 us per step there against 5.8 / 6.0 (round 5,
            // tools/exp/ring_layer_time.py).  DSMI_RNN_KERNEL=ring8 / ring4: one form everywhere (A/B runs, the forms' own tests).
            const bool eight = ring_only8 || (!m->ring4 && std::min(ring_ntw, ntiles - t0) <= 2 && rnn_persist_ring_tiles(m->geom16, B, rcus) > 0);
            if (eight) pl.ntw = std::min(ring_ntw, rnn_persist_ring_tiles(m->geom16, B, rcus));      // (its own cap: at most two real tiles are left)
            ok = eight ? launch_rnn_persist_ring(pl, s) : launch_rnn_persist_ring4(pl, s);
            ring_gate_record(gate, s, first, nw, rcus);
        }
        if (ring_ntw && ok) return;
        for (int p0 = 0; !ring_ntw && p0 < (duo ? total_pairs : 1) && ok; p0 += window) {
            const double part = duo ? (double)std::min(window, total_pairs - p0) / total_pairs : 1.0;
            if (duo) { pl.pair0 = p0; pl.npairs = std::min(window, total_pairs - p0); }
            pl.ev = timer_arm(m, KK_PERSIST, true, part * 2.0 * Dd * GH * m->desc.rnn_hidden_size * sumlen,
                              part * 4.0 * Dd * (GH * m->desc.rnn_hidden_size + (double)To * B * (GH + 2.0 * m->desc.rnn_hidden_size)));
            PersistGate* gate = persist_gate(m->device);
            std::lock_guard<std::mutex> lk(gate->mu);       // wait -> launch -> record is atomic against other host threads
            // a half-CU / half-chip kernel takes one lane (a pair of gate slots), anything else the device
            const int width = ((waves == 4 && !duo) || duo_lane) ? 2 : kMaxLanes;
            gate_wait(gate, s, m->lane, width);
            ok = duo ? launch_rnn_persist_duo(pl, s) : launch_rnn_persist16(pl, s);
            gate_record(gate, s, m->lane, width);
        }
        if (ok) return;
        // (not reachable for eligible shapes; the x-projection is in the other column order, so redo it)
        gl.w_sp = m->rnn[l].wih_sp; gl.bias = m->rnn[l].bih; gl.N = m->geom.Np; gl.ldc = m->geom.Np; gl.ev = EvPair{};
        launch_gemm(gl, s);
    }
    if (m->rnn_mode == 1 && rnn_persist_eligible(m->geom, B, m->n_cus)) {
        // whole layer in one launch; counters are single-use per step, zeroed right before
        RnnPersistLaunch pl;
        pl.g = m->geom;
        for (int dd = 0; dd < 2; ++dd) { pl.whh_sp[dd] = m->rnn[l].whh_sp[dd]; pl.bhh[dd] = m->rnn[l].bhh[dd]; pl.out[dd] = m->hbuf[dst][dd]; }
        pl.xp = m->xp; pl.lens_dev = m->lens_dev; pl.hpack_sp = m->hpack_sp; pl.counters = m->pcnt; pl.err = m->perr; pl.B = B; pl.T = To;
        pl.spin_limit = m->spin_limit;
        if (m->drop_layer == l) { pl.drop_wg = m->drop_wg; pl.drop_step = m->drop_step; }
        (void)hipMemsetAsync(m->pcnt, 0, sizeof(unsigned) * (size_t)m->geom.D * ceil_div(B, 32) * To, s);
        // Two persistent kernels must never share the device (see persist_gate): chain them through the per-device
        // event.  A layer too wide for both directions at once runs them one after the other.
        const int ny = m->geom.nwg * m->geom.D <= m->n_cus ? m->geom.D : 1;
        PersistGate* gate = persist_gate(m->device);
        std::lock_guard<std::mutex> lk(gate->mu);
        gate_wait(gate, s, 0, kMaxLanes);              // the first-generation kernel is sized for the whole device
        bool ok = true;
        for (int d0 = 0; d0 < m->geom.D && ok; d0 += ny) {
            const double part = (double)ny;
            pl.d0 = d0; pl.ny = ny;
            pl.ev = timer_arm(m, KK_PERSIST, true, 2.0 * part * GH * m->desc.rnn_hidden_size * sumlen,
                              4.0 * part * (GH * m->desc.rnn_hidden_size + (double)To * B * (GH + 2.0 * m->desc.rnn_hidden_size)));
            ok = launch_rnn_persist(pl, s);
        }
        gate_record(gate, s, 0, kMaxLanes);
        if (ok) return;
    }
    RnnStepLaunch st;
    st.g = m->geom;
    for (int dd = 0; dd < 2; ++dd) {
        st.whh_packed[dd] = m->rnn[l].whh[dd]; st.bhh[dd] = m->rnn[l].bhh[dd];
        st.out[dd] = m->hbuf[dst][dd]; st.cstate[dd] = m->cst[dd];
    }
    st.xp = m->xp; st.lens_dev = m->lens_dev; st.B = B; st.T = To; st.hpack = m->hpack;
    for (int step = 0; step < To; ++step) {
        st.step = step;
        // algorithmic FLOPs of this launch: clips still running at this step (both directions)
        int act = 0;
        for (int i = 0; i < B; ++i) act += step < m->host_out_lens[i] ? 1 : 0;
        st.ev = timer_arm(m, KK_STEP, (step & 7) == 3, 2.0 * Dd * GH * m->desc.rnn_hidden_size * act,
                          4.0 * Dd * (GH * m->desc.rnn_hidden_size + (double)B * (GH + 2.0 * m->desc.rnn_hidden_size)));
        launch_rnn_step(st, s);
    }
}

static GemmLaunch xproj_gemm(dsmi_model* m, int l, int B, int To) {
    GemmLaunch gl{};
    const RnnW& r = m->rnn[l];
    gl.w = r.wih; gl.bias = r.bih; gl.c = m->xp;
    gl.w_sp = m->gemm_mode == 1 ? r.wih_sp : nullptr;
    gl.a_sp = m->a_sp;
    gl.M = To * B; gl.N = m->geom.Np; gl.K = r.K; gl.ldw = r.ldw; gl.ldc = m->geom.Np;
    gl.B = B; gl.T = To;
    return gl;
}

// host_out_lens -> lens_dev through a small ring of pinned staging slots (an async copy must not read pageable
// memory that the next call overwrites); a slot is reused only after the copy that read it has completed.
static int stage_lens(dsmi_model* m, int B, hipStream_t s) {
    if (B > m->stage_cap) {
        HIP_OK(m, hipDeviceSynchronize());
        if (m->lens_stage) (void)hipHostFree(m->lens_stage);
        m->lens_stage = nullptr;
        const int cap = std::max(B, 64);
        HIP_OK(m, hipHostMalloc((void**)&m->lens_stage, sizeof(int32_t) * (size_t)cap * dsmi_model::kStage, hipHostMallocDefault));
        m->stage_cap = cap;
        for (int i = 0; i < dsmi_model::kStage; ++i) {
            if (!m->stage_ev[i]) HIP_OK(m, hipEventCreateWithFlags(&m->stage_ev[i], hipEventDisableTiming));
            m->stage_used[i] = false;
        }
    }
    const int slot = m->stage_next++ % dsmi_model::kStage;
    if (m->stage_used[slot]) HIP_OK(m, hipEventSynchronize(m->stage_ev[slot]));
    int32_t* h = m->lens_stage + (size_t)slot * m->stage_cap;
    std::memcpy(h, m->host_out_lens.data(), sizeof(int32_t) * B);
    HIP_OK(m, hipMemcpyAsync(m->lens_dev, h, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
    HIP_OK(m, hipEventRecord(m->stage_ev[slot], s));
    m->stage_used[slot] = true;
    return DSMI_OK;
}

static int forward_enqueue(dsmi_model* m, const float* feat, int B, int T, float* probs, hipStream_t s);

// After a hand-off timeout: clear the error word and keep this handle on the per-step path from now on.
static int persist_give_up(dsmi_model* m, hipStream_t s) {
    m->rnn_mode = 0;
    HIP_OK(m, hipMemsetAsync(m->perr, 0, sizeof(unsigned), s));
    return DSMI_OK;
}

// Collect the oldest uncollected forward (its event has completed or `wait`): DSMI_OK, DSMI_RECOMPUTED or < 0.
static int collect_oldest(dsmi_model* m, bool recompute) {
    dsmi_model::FwdSlot& f = m->fwd[m->fwd_head];
    HIP_OK(m, hipEventSynchronize(f.done));
    m->fwd_head = (m->fwd_head + 1) % dsmi_model::kFwdRing;
    m->fwd_count -= 1;
    if (*f.err_host == 0) return DSMI_OK;
    *f.err_host = 0;
    hipStream_t s = (hipStream_t)f.stream;
    int rc;
    if ((rc = persist_give_up(m, s))) return rc;
    if (!recompute)
        return fail(m, DSMI_ERR_TIMEOUT, "the persistent recurrent kernel timed out in an earlier forward whose status was never "
                                         "collected with dsmi_forward_status: those results were invalid; this handle now runs "
                                         "one launch per step");
    m->host_out_lens.resize(f.B);
    for (int i = 0; i < f.B; ++i) m->host_out_lens[i] = seq_len(m, f.lens[i]);
    if ((rc = forward_enqueue(m, f.feat, f.B, f.T, f.probs, s))) return rc;
    HIP_OK(m, hipStreamSynchronize(s));
    m->recomputed += 1;
    m->err = "a hand-off wait of the persistent recurrent kernel timed out; the batch was recomputed with one launch per step";
    return DSMI_RECOMPUTED;
}

extern "C" int dsmi_forward(dsmi_model* m, const float* feat, const int32_t* lens, int B, int T, float* probs,
                            int32_t* out_lens, void* stream) {
    if (!m) return DSMI_ERR_INVALID;
    int rc = check_batch(m, lens, B, T);
    if (rc) return rc;
    if (!feat || !probs || !out_lens) return fail(m, DSMI_ERR_INVALID, "null buffer");
    if ((rc = dsmi_reserve(m, B, T))) return rc;
    HIP_OK(m, hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    // Forwards whose status the caller has not collected: drop the finished good ones, fail loudly on a finished bad one
    // (its results were invalid and may have been consumed), and never keep more than the ring holds in flight.
    while (m->fwd_count > 0) {
        const bool full = m->fwd_count == dsmi_model::kFwdRing;
        if (!full && hipEventQuery(m->fwd[m->fwd_head].done) != hipSuccess) break;
        if ((rc = collect_oldest(m, false)) < 0) return rc;
    }
    for (int i = 0; i < B; ++i) out_lens[i] = seq_len(m, lens[i]);
    m->host_out_lens.assign(out_lens, out_lens + B);
    if ((rc = forward_enqueue(m, feat, B, T, probs, s))) return rc;
    dsmi_model::FwdSlot& f = m->fwd[(m->fwd_head + m->fwd_count) % dsmi_model::kFwdRing];
    f.feat = feat; f.probs = probs; f.lens.assign(lens, lens + B); f.B = B; f.T = T; f.stream = stream;
    if (!f.done) {
        HIP_OK(m, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
        HIP_OK(m, hipHostMalloc((void**)&f.err_host, sizeof(unsigned), hipHostMallocDefault));
        *f.err_host = 0;
    }
    HIP_OK(m, hipMemcpyAsync(f.err_host, m->perr, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIP_OK(m, hipEventRecord(f.done, s));
    m->fwd_count += 1;
    return DSMI_OK;
}

// See include/dsmi.h.  Collects the OLDEST dsmi_forward of this handle whose status has not been collected yet.
extern "C" int dsmi_forward_status(dsmi_model* m) {
    if (!m) return DSMI_ERR_INVALID;
    if (m->fwd_count == 0) return DSMI_OK;
    HIP_OK(m, hipSetDevice(m->device));
    return collect_oldest(m, true);
}

// See include/dsmi.h: has the handle's oldest uncollected forward finished?  Never blocks.
extern "C" int dsmi_forward_ready(dsmi_model* m) {
    if (!m) return DSMI_ERR_INVALID;
    if (m->fwd_count == 0) return 1;
    (void)hipSetDevice(m->device);
    const hipError_t e = hipEventQuery(m->fwd[m->fwd_head].done);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    m->err = std::string("hipEventQuery: ") + hipGetErrorString(e);
    return DSMI_ERR_HIP;
}

static int forward_enqueue(dsmi_model* m, const float* feat, int B, int T, float* probs, hipStream_t s) {
    int rc;
    const dsmi_model_desc& d = m->desc;
    const int To = seq_len(m, T), ys = round_up(To, 4);
    const int32_t* out_lens = m->host_out_lens.data();
    if ((rc = stage_lens(m, B, s))) return rc;
    if (m->Hs != d.rnn_hidden_size)
        for (int i = 0; i < 2; ++i)
            for (int dd = 0; dd < m->geom.D; ++dd)
                HIP_OK(m, hipMemsetAsync(m->hbuf[i][dd], 0, sizeof(float) * (size_t)To * B * m->Hs, s));

    if (m->profiling) HIP_OK(m, hipEventRecord(m->ev[0], s));
    const float* cx;
    unsigned* csem = (dense_tokens() > 0 && m->inflight >= 2 && dense_scope() == 0) ? dense_sem(m->device) : nullptr;     // (see dense_enter_kernel)
    if (csem) hipLaunchKernelGGL(dense_enter_kernel, dim3(1), dim3(64), 0, s, csem, (unsigned)dense_tokens());
    run_conv(m, feat, B, T, To, ys, s, &cx);
    if (csem) hipLaunchKernelGGL(dense_leave_kernel, dim3(1), dim3(64), 0, s, csem);
    if (m->profiling) HIP_OK(m, hipEventRecord(m->ev[1], s));

    for (int l = 0; l < d.rnn_layers; ++l) {
        GemmLaunch gl = xproj_gemm(m, l, B, To);
        if (l == 0) {
            gl.mode = GEMM_A_CONV; gl.a = cx; gl.ys = ys;
        } else {
            gl.mode = GEMM_A_SUM_BN;
            gl.a = m->hbuf[(l - 1) & 1][0]; gl.a2 = m->geom.D == 2 ? m->hbuf[(l - 1) & 1][1] : nullptr;
            gl.alpha = m->rnn[l].bn_a; gl.beta = m->rnn[l].bn_b; gl.lda = m->Hs;
        }
        run_rnn_layer(m, l, gl, B, To, l & 1, s);
    }
    if (m->profiling) HIP_OK(m, hipEventRecord(m->ev[2], s));
    const int last = (d.rnn_layers - 1) & 1;
    HeadLaunch h;
    h.bn_a = m->fc_a; h.bn_b = m->fc_b; h.w_packed = m->fc_wp; h.H = d.rnn_hidden_size; h.C = d.n_labels;
    h.T = To; h.B = B; h.probs = probs;
    if (!d.bidirectional) {   // model.py:508-509
        launch_lookahead(m->hbuf[last][0], m->look_w, m->look_buf, To, B, d.rnn_hidden_size, d.context, s);
        h.x1 = m->look_buf; h.x2 = nullptr;
    } else {
        h.x1 = m->hbuf[last][0]; h.x2 = m->hbuf[last][1];
    }
    {
        double sumlen = 0;
        for (int i = 0; i < B; ++i) sumlen += out_lens[i];
        h.ev = timer_arm(m, KK_HEAD, true, 2.0 * sumlen * d.rnn_hidden_size * d.n_labels,
                         4.0 * To * B * ((d.bidirectional ? 2.0 : 1.0) * m->Hs + d.n_labels));
    }
    launch_head(h, s);
    if (m->profiling) HIP_OK(m, hipEventRecord(m->ev[3], s));
    HIP_OK(m, hipGetLastError());

    // bookkeeping for roofline maths: recurrent and total algorithmic FLOPs
    m->n_step_launches = (int64_t)To * d.rnn_layers;
    double rec = 0, tot = 0;
    for (int i = 0; i < B; ++i) {
        const double t = out_lens[i];
        double conv = 0;
        for (int l = 0; l < d.conv_layers; ++l) {
            const ConvSpec& sp = kConvSpecs[l];
            conv += (double)sp.co * m->conv_fo[l] * t * sp.ci * sp.kf * sp.kt;
        }
        const double H = d.rnn_hidden_size, G = m->geom.G, D = m->geom.D;
        double r = 0, ip = 0;
        for (int l = 0; l < d.rnn_layers; ++l) {
            r += D * t * G * H * H;
            ip += D * t * G * H * (l == 0 ? m->I0 : H);
        }
        rec += 2 * r;
        tot += 2 * (conv + r + ip + t * H * d.n_labels);
    }
    m->step_flops = rec;
    m->total_flops = tot;
    if (m->profiling == 1) {
        HIP_OK(m, hipStreamSynchronize(s));
        float ms;
        HIP_OK(m, hipEventElapsedTime(&ms, m->ev[0], m->ev[1])); m->stage_us[0] = ms * 1e3;
        HIP_OK(m, hipEventElapsedTime(&ms, m->ev[1], m->ev[2])); m->stage_us[2] = ms * 1e3;  // GEMMs + steps
        HIP_OK(m, hipEventElapsedTime(&ms, m->ev[2], m->ev[3])); m->stage_us[3] = ms * 1e3;
        HIP_OK(m, hipEventElapsedTime(&ms, m->ev[0], m->ev[3])); m->stage_us[4] = ms * 1e3;
        m->stage_us[1] = 0;
    }
    return DSMI_OK;
}

extern "C" int dsmi_conv_stack(dsmi_model* m, const float* feat, const int32_t* lens, int B, int T, float* out, void* stream) {
    if (!m) return DSMI_ERR_INVALID;
    int rc = check_batch(m, lens, B, T);
    if (rc) return rc;
    if ((rc = dsmi_reserve(m, B, T))) return rc;
    HIP_OK(m, hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int To = seq_len(m, T), ys = round_up(To, 4);
    m->host_out_lens.resize(B);
    for (int i = 0; i < B; ++i) m->host_out_lens[i] = seq_len(m, lens[i]);
    if ((rc = stage_lens(m, B, s))) return rc;
    const float* cx;
    run_conv(m, feat, B, T, To, ys, s, &cx);
    // strip the time-stride padding: [B][C*F][ys] -> [B][C*F][To]
    const int cl = m->desc.conv_layers - 1;
    const size_t rows = (size_t)B * kConvSpecs[cl].co * m->conv_fo[cl];
    HIP_OK(m, hipMemcpy2DAsync(out, sizeof(float) * To, cx, sizeof(float) * ys, sizeof(float) * To, rows, hipMemcpyDeviceToDevice, s));
    HIP_OK(m, hipStreamSynchronize(s));
    HIP_OK(m, hipGetLastError());
    return DSMI_OK;
}

extern "C" int dsmi_rnn_layer(dsmi_model* m, int layer, const float* x, const int32_t* out_lens, int B, int To, float* y, void* stream) {
    if (!m) return DSMI_ERR_INVALID;
    if (!m->finalized) return fail(m, DSMI_ERR_NOT_READY, "dsmi_model_finalize has not been called");
    if (layer < 0 || layer >= m->desc.rnn_layers || !x || !y || !out_lens || B < 1 || To < 1) return fail(m, DSMI_ERR_INVALID, "bad rnn_layer arguments");
    for (int i = 0; i < B; ++i) {
        if (out_lens[i] < 1 || out_lens[i] > To) return fail(m, DSMI_ERR_INVALID, "length outside 1..T");
        if (i && out_lens[i] > out_lens[i - 1]) return fail(m, DSMI_ERR_UNSORTED, "`lengths` array must be sorted in decreasing order");
    }
    // workspaces are sized by input frames; find a T whose seq_len covers To
    int Tin = To;
    while (seq_len(m, Tin) < To) Tin += 1;
    int rc;
    if ((rc = dsmi_reserve(m, B, Tin))) return rc;
    HIP_OK(m, hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int H = m->desc.rnn_hidden_size;
    m->host_out_lens.assign(out_lens, out_lens + B);
    if ((rc = stage_lens(m, B, s))) return rc;
    const RnnW& r = m->rnn[layer];
    const int I = layer == 0 ? m->I0 : H;
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (int dd = 0; dd < m->geom.D; ++dd)
            HIP_OK(m, hipMemsetAsync(m->hbuf[0][dd], 0, sizeof(float) * (size_t)To * B * m->Hs, s));
        launch_pad_rows(x, m->xin, (size_t)To * B, I, r.ldw, s);
        GemmLaunch gl = xproj_gemm(m, layer, B, To);
        if (layer == 0) {
            gl.mode = GEMM_A_ROWMAJOR; gl.a = m->xin; gl.lda = r.ldw;
        } else {
            gl.mode = GEMM_A_SUM_BN; gl.a = m->xin; gl.a2 = nullptr; gl.alpha = r.bn_a; gl.beta = r.bn_b; gl.lda = r.ldw;
        }
        run_rnn_layer(m, layer, gl, B, To, 0, s);
        launch_add2(m->hbuf[0][0], m->geom.D == 2 ? m->hbuf[0][1] : nullptr, y, (size_t)To * B, H, m->Hs, s);
        HIP_OK(m, hipStreamSynchronize(s));
        HIP_OK(m, hipGetLastError());
        // a hand-off timeout of the persistent kernel: recompute this layer with one launch per step, in this call
        unsigned e = 0;
        HIP_OK(m, hipMemcpy(&e, m->perr, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (!e) return DSMI_OK;
        if (attempt == 1) return fail(m, DSMI_ERR_TIMEOUT, "recurrent layer timed out on the per-step path");
        HIP_OK(m, hipMemset(m->perr, 0, sizeof(unsigned)));
        m->rnn_mode = 0;
        m->recomputed += 1;
    }
    return DSMI_OK;
}

extern "C" int dsmi_recompute_count(const dsmi_model* m) { return m ? m->recomputed : DSMI_ERR_INVALID; }

extern "C" int dsmi_model_set_inflight(dsmi_model* m, int batches) {
    if (!m || batches < 1) return DSMI_ERR_INVALID;
    m->inflight = batches;
    return DSMI_OK;
}

extern "C" int dsmi_model_set_ring_windows(dsmi_model* m, int windows) {
    if (!m || windows < 0) return DSMI_ERR_INVALID;
    m->ring_windows = windows;
    return DSMI_OK;
}

extern "C" int dsmi_set_profiling(dsmi_model* m, int level) {
    if (!m) return DSMI_ERR_INVALID;
    m->profiling = level < 0 ? 0 : (level > 2 ? 2 : level);
    // the events of the stamped launches are made HERE, not at the launches: an event's first creation is tens of microseconds of
    // the caller's thread, and a region that is being timed would pay for a hundred of them (bench.py: 20 steps = 80 stamped launches)
    if (m->profiling == 2 && hipSetDevice(m->device) == hipSuccess)
        while (m->kt.free_events.size() < 256) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            m->kt.free_events.push_back(e);
        }
    return DSMI_OK;
}

extern "C" int dsmi_kernel_stats(dsmi_model* m, int kind, int64_t* launches, int64_t* samples, double* avg_us,
                                 double* flops_per_launch, double* bytes_per_launch) {
    if (!m || kind < 0 || kind >= KK_COUNT) return DSMI_ERR_INVALID;
    (void)hipSetDevice(m->device);
    timer_resolve(m);
    const KernelTimer& t = m->kt;
    if (launches) *launches = t.launches[kind];
    if (samples) *samples = t.samples[kind];
    if (avg_us) *avg_us = t.samples[kind] ? t.sum_us[kind] / t.samples[kind] : 0.0;
    if (flops_per_launch) *flops_per_launch = t.launches[kind] ? t.flops[kind] / t.launches[kind] : 0.0;
    if (bytes_per_launch) *bytes_per_launch = t.launches[kind] ? t.bytes[kind] / t.launches[kind] : 0.0;
    return DSMI_OK;
}

extern "C" int dsmi_reset_kernel_stats(dsmi_model* m) {
    if (!m) return DSMI_ERR_INVALID;
    timer_resolve(m);
    for (int k = 0; k < KK_COUNT; ++k) { m->kt.sum_us[k] = 0; m->kt.samples[k] = 0; m->kt.launches[k] = 0; m->kt.flops[k] = 0; m->kt.bytes[k] = 0; }
    return DSMI_OK;
}

extern "C" double dsmi_stage_time_us(const dsmi_model* m, int stage) {
    if (!m || stage < 0 || stage > 4) return -1.0;
    return m->stage_us[stage];
}

extern "C" int dsmi_last_forward_stats(const dsmi_model* m, int64_t* n_step, double* step_flops, double* total_flops) {
    if (!m) return DSMI_ERR_INVALID;
    if (n_step) *n_step = m->n_step_launches;
    if (step_flops) *step_flops = m->step_flops;
    if (total_flops) *total_flops = m->total_flops;
    return DSMI_OK;
}

// ---- diagnostics: per-wave phase timestamps (s_memrealtime, 100 MHz) of ONE recurrent step launch.
// Runs steps 0..step of `layer` on whatever the workspaces hold (timing only) and returns
// stamps[D*nwg][8 waves][8] for the last one.  GRU, B <= 32.
extern "C" int dsmi_debug_step_stamps(dsmi_model* m, int layer, int B, int To, int step, uint64_t* stamps_host, int64_t n_words) {
    if (!m || !m->finalized || m->desc.rnn_type != DSMI_RNN_GRU || B > 32 || layer < 0 || layer >= m->desc.rnn_layers) return DSMI_ERR_INVALID;
    int Tin = To;
    while (seq_len(m, Tin) < To) Tin += 1;
    int rc;
    if ((rc = dsmi_reserve(m, B, Tin))) return rc;
    HIP_OK(m, hipSetDevice(m->device));
    const int64_t need = (int64_t)m->geom.D * m->geom.nwg * 8 * 8;
    if (n_words < need) return fail(m, DSMI_ERR_INVALID, "stamp buffer too small");
    unsigned long long* dbg;
    HIP_OK(m, hipMalloc((void**)&dbg, sizeof(unsigned long long) * need));
    HIP_OK(m, hipMemset(dbg, 0, sizeof(unsigned long long) * need));
    std::vector<int32_t> lens(B, To);
    HIP_OK(m, hipMemcpy(m->lens_dev, lens.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    HIP_OK(m, hipMemset(m->xp, 0, sizeof(float) * (size_t)To * B * m->geom.Np));
    for (int dd = 0; dd < m->geom.D; ++dd) HIP_OK(m, hipMemset(m->hbuf[0][dd], 0, sizeof(float) * (size_t)To * B * m->Hs));
    RnnStepLaunch st;
    st.g = m->geom;
    for (int dd = 0; dd < 2; ++dd) {
        st.whh_packed[dd] = m->rnn[layer].whh[dd]; st.bhh[dd] = m->rnn[layer].bhh[dd];
        st.out[dd] = m->hbuf[0][dd]; st.cstate[dd] = m->cst[dd];
    }
    st.xp = m->xp; st.lens_dev = m->lens_dev; st.B = B; st.T = To; st.hpack = m->hpack;
    for (int s2 = 0; s2 <= step; ++s2) {
        st.step = s2;
        st.dbg = s2 == step ? dbg : nullptr;
        launch_rnn_step(st, nullptr);
    }
    HIP_OK(m, hipDeviceSynchronize());
    HIP_OK(m, hipMemcpy(stamps_host, dbg, sizeof(unsigned long long) * need, hipMemcpyDeviceToHost));
    (void)hipFree(dbg);
    return DSMI_OK;
}

// ---- diagnostics: accumulated per-wave phase times (100 MHz ticks) of one persistent layer launch;
// stamps_host[workgroups][8 waves][8]: 0 loop head, 1 wait, 2 h load + MFMA, 3 LDS + barrier, 4 cell (+ publish stores),
// 5 drain + signal.  Returns the number of workgroups stamped (> 0) or a DSMI_ERR_* code (< 0).
// DSMI_STAMP_RING=1: the ring kernel (one window of every tile of B <= 64 clips; <= 128 with DSMI_RING_TILES=8): the four-wave form stamps[workgroup][4 waves][8] (Ring4Args::dbg) = phase work,
// wait for the wave's requests, poll spin, barrier ([7] phases); DSMI_RNN_KERNEL=ring8, the eight-wave form: stamps[workgroup][8 waves][16] (RingArgs::dbg) = M work, M-end waits,
// C work, barrier behind M, barrier behind C, poll spin (100 MHz ticks), shader cycles in M work, slots.
static int ring_stamps(dsmi_model* m, int layer, int B, int To, uint64_t* stamps_host, int64_t n_words);

extern "C" int dsmi_debug_persist_stamps(dsmi_model* m, int layer, int B, int To, uint64_t* stamps_host, int64_t n_words) {
    if (m && m->finalized && std::getenv("DSMI_STAMP_RING")) return ring_stamps(m, layer, B, To, stamps_host, n_words);
    if (!m || !m->finalized || B > 32 || layer < 0 || layer >= m->desc.rnn_layers || !rnn_persist_eligible(m->geom, B, m->n_cus) ||
        m->geom.nwg * m->geom.D > m->n_cus) return DSMI_ERR_INVALID;
    int Tin = To;
    while (seq_len(m, Tin) < To) Tin += 1;
    int rc;
    if ((rc = dsmi_reserve(m, B, Tin))) return rc;
    HIP_OK(m, hipSetDevice(m->device));
    // diagnostics launch persistent kernels too: same rules as the product path -- one process per GPU, and the per-device gate
    // held from here to the end (every launch below is followed by a device synchronise before the lock is released)
    if (!persist_process_lock(m->device)) return fail(m, DSMI_ERR_INVALID, "another process holds this GPU's persistent-kernel lock");
    PersistGate* stamp_gate = persist_gate(m->device);
    std::lock_guard<std::mutex> stamp_lk(stamp_gate->mu);
    HIP_OK(m, hipDeviceSynchronize());
    int pgroups = 0;
    const bool duo = std::getenv("DSMI_STAMP_DUO") && m->have16 && rnn_persist_duo_eligible(m->geom16, B, m->n_cus);
    if (duo) {          // the paired-tile kernel: stamps[workgroup][8 waves][8] = time in slots 0..3 and at the barrier behind each
        const int64_t needd = (int64_t)m->geom16.D * ceil_div(ceil_div(B, 16), 2) * m->geom16.nwg * 8 * 8;
        if (n_words < needd) return fail(m, DSMI_ERR_INVALID, "stamp buffer too small");
        unsigned long long* dbg;
        HIP_OK(m, hipMalloc((void**)&dbg, sizeof(unsigned long long) * needd));
        HIP_OK(m, hipMemset(dbg, 0, sizeof(unsigned long long) * needd));
        std::vector<int32_t> lens(B, To);
        HIP_OK(m, hipMemcpy(m->lens_dev, lens.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
        HIP_OK(m, hipMemset(m->xp, 0, sizeof(float) * (size_t)To * B * m->geom16.Np));
        RnnPersist16Launch pl;
        pl.g = m->geom16;
        for (int dd = 0; dd < 2; ++dd) { pl.whh16[dd] = m->rnn[layer].whh16_sp[dd]; pl.bhh[dd] = m->rnn[layer].bhh[dd]; pl.out[dd] = m->hbuf[0][dd]; }
        pl.xp = m->xp; pl.lens_dev = m->lens_dev; pl.hpack16 = m->hpack16; pl.counters = m->pcnt; pl.err = m->perr;
        pl.B = B; pl.T = To;
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(m, hipMemset(m->pcnt, 0, sizeof(unsigned) * (size_t)m->geom.D * ceil_div(B, 16) * To * kPersist16CntWords));
            pl.dbg = rep ? dbg : nullptr;
            launch_rnn_persist_duo(pl, nullptr);
            HIP_OK(m, hipDeviceSynchronize());
        }
        HIP_OK(m, hipMemcpy(stamps_host, dbg, sizeof(unsigned long long) * needd, hipMemcpyDeviceToHost));
        (void)hipFree(dbg);
        return (int)(needd / 64);
    }
    const bool use16 = m->persist_gen == 2 && m->have16 && rnn_persist16_eligible(m->geom16, B, m->n_cus, &pgroups) &&
                       ceil_div(B, 16) <= pgroups;        // one tile per workgroup: the plain single-tile path is what is stamped
    const int64_t need = use16 ? (int64_t)m->geom16.D * pgroups * m->geom16.nwg * 8 * 8 : (int64_t)m->geom.D * m->geom.nwg * 8 * 8;
    if (n_words < need) return fail(m, DSMI_ERR_INVALID, "stamp buffer too small");
    unsigned long long* dbg;
    HIP_OK(m, hipMalloc((void**)&dbg, sizeof(unsigned long long) * need));
    HIP_OK(m, hipMemset(dbg, 0, sizeof(unsigned long long) * need));
    std::vector<int32_t> lens(B, To);
    HIP_OK(m, hipMemcpy(m->lens_dev, lens.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    HIP_OK(m, hipMemset(m->xp, 0, sizeof(float) * (size_t)To * B * std::max(m->geom.Np, m->have16 ? m->geom16.Np : 0)));
    if (use16) {
        RnnPersist16Launch pl;
        pl.g = m->geom16;
        for (int dd = 0; dd < 2; ++dd) { pl.whh16[dd] = m->rnn[layer].whh16_sp[dd]; pl.bhh[dd] = m->rnn[layer].bhh[dd]; pl.out[dd] = m->hbuf[0][dd]; }
        pl.xp = m->xp; pl.lens_dev = m->lens_dev; pl.hpack16 = m->hpack16; pl.counters = m->pcnt; pl.err = m->perr;
        pl.B = B; pl.T = To; pl.pgroups = pgroups;
        for (int rep = 0; rep < 2; ++rep) {     // first pass warms up, second is stamped
            HIP_OK(m, hipMemset(m->pcnt, 0, sizeof(unsigned) * (size_t)m->geom.D * ceil_div(B, 16) * To * kPersist16CntWords));
            pl.dbg = rep ? dbg : nullptr;
            launch_rnn_persist16(pl, nullptr);
            HIP_OK(m, hipDeviceSynchronize());
        }
        HIP_OK(m, hipMemcpy(stamps_host, dbg, sizeof(unsigned long long) * need, hipMemcpyDeviceToHost));
        (void)hipFree(dbg);
        return (int)(need / 64);      // number of workgroups stamped
    }
    RnnPersistLaunch pl;
    pl.g = m->geom;
    for (int dd = 0; dd < 2; ++dd) { pl.whh_sp[dd] = m->rnn[layer].whh_sp[dd]; pl.bhh[dd] = m->rnn[layer].bhh[dd]; pl.out[dd] = m->hbuf[0][dd]; }
    pl.xp = m->xp; pl.lens_dev = m->lens_dev; pl.hpack_sp = m->hpack_sp; pl.counters = m->pcnt; pl.err = m->perr; pl.B = B; pl.T = To;
    pl.d0 = 0; pl.ny = m->geom.D;
    for (int rep = 0; rep < 2; ++rep) {     // first pass warms up, second is stamped
        HIP_OK(m, hipMemset(m->pcnt, 0, sizeof(unsigned) * (size_t)m->geom.D * To));
        pl.dbg = rep ? dbg : nullptr;
        launch_rnn_persist(pl, nullptr);
        HIP_OK(m, hipDeviceSynchronize());
    }
    HIP_OK(m, hipMemcpy(stamps_host, dbg, sizeof(unsigned long long) * need, hipMemcpyDeviceToHost));
    (void)hipFree(dbg);
    return (int)(need / 64);          // number of workgroups stamped
}

static int ring_stamps(dsmi_model* m, int layer, int B, int To, uint64_t* stamps_host, int64_t n_words) {
    if (B < 1 || B > 128 || layer < 0 || layer >= m->desc.rnn_layers || !m->have16) return DSMI_ERR_INVALID;
    const int cap = m->ring8 ? rnn_persist_ring_tiles(m->geom16, B, m->n_cus) : rnn_persist_ring4_tiles(m->geom16, B, m->n_cus);
    if (cap < ceil_div(B, 16)) return DSMI_ERR_INVALID;
    int Tin = To;
    while (seq_len(m, Tin) < To) Tin += 1;
    int rc;
    if ((rc = dsmi_reserve(m, B, Tin))) return rc;
    HIP_OK(m, hipSetDevice(m->device));
    if (!persist_process_lock(m->device)) return fail(m, DSMI_ERR_INVALID, "another process holds this GPU's persistent-kernel lock");
    PersistGate* stamp_gate = persist_gate(m->device);
    std::lock_guard<std::mutex> stamp_lk(stamp_gate->mu);      // (no other launch of this process can start; the device is drained below)
    HIP_OK(m, hipDeviceSynchronize());
    // per workgroup: the eight-wave form 8 waves x 16 words, the four-wave form 4 waves x 8 words (in the first 32 of the 128)
    const int64_t need = (int64_t)rnn_persist_ring_cus(m->geom16) * 8 * 16;
    if (n_words < need) return fail(m, DSMI_ERR_INVALID, "stamp buffer too small");
    struct DevBuf { unsigned long long* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } } dbg;      // freed on every return
    HIP_OK(m, hipMalloc((void**)&dbg.p, sizeof(unsigned long long) * need));
    HIP_OK(m, hipMemset(dbg.p, 0, sizeof(unsigned long long) * need));
    std::vector<int32_t> lens(B, To);
    HIP_OK(m, hipMemcpy(m->lens_dev, lens.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    HIP_OK(m, hipMemset(m->xp, 0, sizeof(float) * (size_t)To * B * m->geom16.Np));
    RnnPersist16Launch pl;
    pl.g = m->geom16;
    for (int dd = 0; dd < 2; ++dd) { pl.whh16[dd] = m->rnn[layer].whh16_sp[dd]; pl.bhh[dd] = m->rnn[layer].bhh[dd]; pl.out[dd] = m->hbuf[0][dd]; }
    pl.xp = m->xp; pl.lens_dev = m->lens_dev; pl.hpack16 = m->hpack16; pl.counters = m->pcnt; pl.err = m->perr;
    pl.B = B; pl.T = To; pl.tile0 = 0; pl.ntw = ceil_div(B, 16); pl.nwin = 1;
    bool ok = true;
    for (int rep = 0; rep < 2 && ok; ++rep) {
        HIP_OK(m, hipMemset(m->pcnt, 0, sizeof(unsigned) * (size_t)m->geom.D * ceil_div(B, 16) * To * kPersist16CntWords));
        pl.dbg = rep ? dbg.p : nullptr;
        ok = m->ring8 ? launch_rnn_persist_ring(pl, nullptr) : launch_rnn_persist_ring4(pl, nullptr);
        HIP_OK(m, hipDeviceSynchronize());
    }
    HIP_OK(m, hipMemcpy(stamps_host, dbg.p, sizeof(unsigned long long) * need, hipMemcpyDeviceToHost));
    return ok ? (int)(need / 128) : DSMI_ERR_INVALID;
}
