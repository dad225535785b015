// FQL step engine for MI355X (gfx950): host side + C ABI (include/fql_amd.h).
//
// One stateful engine owns every device buffer of the step (params, Adam state, target network,
// activations, dataset).  A training step is written ONCE as a straight-line program of tile tasks
// with explicit read/write sets; a list scheduler levels the program (RAW/WAR/WAW on buffers), all
// tasks of one kernel type in one level become ONE launch driven by a task table in HBM, and the
// level sequence is captured into a hipGraph.  The kernel boundary is the only grid-wide barrier used
// (cheapest on gfx950: ~1.3 us vs >4 us for an in-kernel grid barrier).
//
// Reference behaviour being replaced: agents/fql.py:22-171, utils/networks.py:34-61,153-235,
// utils/flax_utils.py:90-159, utils/datasets.py:64-100,435-495 (zhouzypaul/fql).
#include "../../include/fql_amd.h"
#include "fql_kernels.h"
#include "fql_conv.h"
#include "fql_chain.h"
#include "fql_aux.h"
#include "fql_xchain.h"
#include <hip/hip_ext.h>
#include "fql_aql.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

#define FQL_LANES 4

namespace {

thread_local std::string g_create_error;

struct HipError {
    std::string msg;
};
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            char buf_[512];                                                                          \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw HipError{buf_};                                                                    \
        }                                                                                            \
    } while (0)

struct Invalid {
    std::string msg;
};
[[noreturn]] void invalid(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Invalid{buf};
}

inline int pad16(int x) { return (x + 15) & ~15; }

// ------------------------------------------------------------------------------------------------
// networks in the parameter arena
// ------------------------------------------------------------------------------------------------
struct Layer {
    int in, out, in_p, out_p;
    size_t w, b;      // arena offsets (floats)
    bool ln;          // LayerNorm after this layer's GELU
    size_t g, be;     // LN scale / bias offsets
};
struct Net {
    std::vector<Layer> layers;
    size_t off = 0, size = 0;
    int in() const { return layers.front().in; }
    int in_p() const { return layers.front().in_p; }
    int nl() const { return (int)layers.size(); }
};
enum { NET_C0 = 0, NET_C1, NET_BC, NET_OS, NET_T0, NET_T1, NUM_NETS };

// IMPALA encoder in the arena (utils/encoders.py:61-100): per stack 1 + 2 num_blocks convolutions, then Dense + GELU.
// A convolution leaf is stored exactly as flax lays it out, [3][3][cin][cout] = [tap][cin][cout], bias [cout].
struct ConvL {
    size_t w, b;
    int cin, cout;
    float *Wf = nullptr, *Wb = nullptr;   // LDS-layout copies refreshed by fql_conv_wprep_kernel at the start of each pass
    bool split = false;                   // precision = 2: the copies are bf16 hi / lo planes (fql_conv3x3_split_kernel)
    bool pool_fused = false;              // a stack's first convolution that runs fused with the stack's max-pool (fp32 forward copy in both precisions)
};
struct EncStack {
    std::vector<ConvL> conv;  // conv[0] at the stack's input resolution, the rest after the 2x max-pool
    int H, W;                 // input resolution of the stack
};
struct EncNet {
    std::vector<EncStack> stacks;
    Layer dense;              // flat -> enc_dim
    int flat = 0;
    size_t off = 0, size = 0;
};
enum { ENC_C = 0, ENC_BC, ENC_OS, ENC_T, NUM_ENC };
inline int pad16c(int c) { return (c + 15) & ~15; }

struct Segment {  // one contiguous padded block of a leaf
    size_t off;
    int rows, cols, rows_p, cols_p;  // logical and padded 2-D shape (vectors: rows = 1)
};
struct Leaf {
    std::string name;
    int ndim;
    int64_t shape[4];
    std::vector<Segment> segs;  // 1 (actors) or 2 (ensemble members)
    bool trainable;
    int train_id;  // index among trainable leaves or -1
};

// ------------------------------------------------------------------------------------------------
// program = ordered ops with read/write sets -> levels -> launches
// ------------------------------------------------------------------------------------------------
enum OpType { OP_GEMM, OP_GEMM64, OP_WGRAD, OP_LNBWD, OP_PREP, OP_POSTOS, OP_EULER_FIN, OP_PEC, OP_LOSS_CRITIC, OP_LOSS_Q, OP_LOSS_BC,
              OP_LOSS_ACTOR, OP_CONV_WPREP, OP_CONV, OP_CONV_U8, OP_POOL, OP_POOL_BWD, OP_CONV_WGRAD, OP_CONV_WRED, OP_ENC_DZ, OP_CHAIN, OP_WFRAG, OP_XCHAIN, OP_HEAD_DGRAD, OP_DGRAD0, OP_ADAM,
              OP_FINALIZE };

struct Op {
    OpType type;
    std::vector<const void*> reads, writes;
    GemmTask gemm;
    WgradTask wgrad;
    LnBwdTask ln;
    PrepArgs prep;
    PostOsArgs postos;
    LossCriticArgs lc;
    LossQArgs lq;
    LossBcArgs lb;
    LossActorArgs la;
    EulerFinishArgs ef;
    PecArgs pec;
    ConvArgs conv;
    PoolArgs pool;
    PoolBwdArgs poolb;
    ConvWgradArgs cw;
    ConvWredArgs cwr;
    EncDzArgs edz;
    ChainArgs chain;
    XChainArgs xchain;
    const ConvWprepTask* wprep_tasks = nullptr;
    int wprep_n = 0;
    int cw_grid = 0;
    int adam_c0 = 0, adam_n = -1;  // chunk range of an Adam op (-1: all chunks)
    int adam_c1 = 0, adam_n1 = 0;  // second chunk range of the same launch (two modules merged), adam_n1 = 0: none
    int fin_mode = 0;
    int level = 0;
    int lane = 0;            // 0 = critical lane (Euler chain, one-step backward), 1 = side lane
    std::vector<int> deps;   // indices of earlier ops this op must follow (RAW / WAW / WAR)
};

struct Launch {
    OpType type;
    int grid = 0, ntasks = 0;
    size_t lds = 0;
    void* table = nullptr;  // device task table (GEMM/WGRAD/LNBWD)
    Op op;                  // arg-struct kernels
    bool pool_fused = false; // OP_CONV_U8: the stack's max-pool runs inside the convolution kernel (fql_conv3x3_pool_kernel)
    int lane = 0;
    bool tmt2 = false, kbig = false, euler = false;
    double macs = 0.0;           // algorithmic multiply-accumulates of the GEMM-shaped tasks of this launch (roofline accounting)
    bool side = false;           // merged gemm64 + wgrad + lnbwd launch
    void *table_w = nullptr, *table_l = nullptr, *table_m = nullptr;
    int n_w = 0, n_l = 0, tile_w = 0, tile_l = 0, tile_m = 0;
    std::vector<GemmTask> dg_tasks;   // OP_DGRAD0: the (<= 4) tasks of this launch, handed over as the kernel argument
    std::vector<int> waits;      // launches of the OTHER lane that must have completed
    bool record_after = false;   // some launch of the other lane waits on this one
    hipEvent_t ev = nullptr;
};

struct Program {
    std::vector<Op> ops;
    std::vector<Launch> launches;
    bool two_lanes = false;  // more than one lane in use
    bool lane_used[FQL_LANES] = {};
    hipEvent_t ev_fork = nullptr, ev_join[FQL_LANES] = {};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int64_t macs = 0;
    // threaded eager issue (run_threaded): sequence number of the run whose event record of launch i has been enqueued
    std::unique_ptr<std::atomic<uint64_t>[]> rec;
    size_t rec_n = 0;
    uint64_t run_seq = 0;
};

// One host thread per extra lane for the threaded eager executor: spins for a short while after a job (back-to-back updates find it
// hot), then sleeps on a condition variable.
struct LaneWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> job{0}, done{0};
    std::atomic<bool> stop{false};
    Program* pr = nullptr;
    hipStream_t s = nullptr;
    int lane = 0;
    std::string err;     // set by the worker when a HIP call failed during the job
};

struct PassBuf {
    int net = 0, M = 0;
    const void* id = nullptr;          // dependency id base
    float* x0 = nullptr;               // [M, in_p]
    std::vector<float*> g, xn, z, stats;  // per hidden layer
    std::vector<float*> lnpart;        // LN layers: per-row partial sums per 32 columns (LDS-tiled kernels)
    float* out = nullptr;              // [M, out_p]
    std::vector<float*> dz;            // per layer gradient wrt pre-activation / output
    std::vector<float*> dy;            // LN nets: gradient wrt LN output
    float* dx0 = nullptr;
};

// activations of one encoder pass over n images (kept for the backward pass)
struct EncBuf {
    int enc = 0, n = 0;
    const unsigned char* img = nullptr;
    struct St { float *c0 = nullptr, *pool = nullptr; unsigned char* arg = nullptr; std::vector<float*> c1, y; };
    std::vector<St> st;
    float *frelu = nullptr, *z = nullptr, *E = nullptr;   // [n, flat], [n, enc_dim] x 2
    // backward scratch (allocated only for differentiated passes)
    float *dz = nullptr, *dA = nullptr, *dB = nullptr, *dC = nullptr;
    std::map<const void*, float*> wpart;   // per convolution (keyed by its arena weight pointer): weight-gradient partials, so the
                                            // wgrads of a pass are independent and their folds share launches
};

}  // namespace

struct fql_engine {
    fql_config cfg{};
    uint64_t seed = 0;
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr, stream4 = nullptr;
    bool allow64 = true;
    int emit_lane = 0;
    std::vector<Op>* defer_wgrads = nullptr;   // when set, emit_backward collects weight-gradient ops here instead of emitting them in place
    std::string err;

    Net nets[NUM_NETS];
    EncNet encs[NUM_ENC];
    ConvWprepTask* enc_wprep[NUM_ENC] = {nullptr, nullptr, nullptr, nullptr};   // device task tables of fql_conv_wprep_kernel
    int enc_nconv[NUM_ENC] = {0, 0, 0, 0};
    std::vector<void*> enc_allocs;
    bool visual = false;
    int enc_dim = 0;
    unsigned char* img_all = nullptr;   // [2B] images: obs batch, then next_obs batch (fixed address: graphs bake it in)
    EncBuf eb_c, eb_t, eb_bc, eb_os;
    size_t n_train = 0, n_total = 0, critic_size = 0;
    float *P = nullptr, *G = nullptr, *Mu = nullptr, *Nu = nullptr;
    std::vector<Leaf> leaves;
    int n_train_leaves = 0;
    AdamChunk* d_chunks = nullptr;
    float* d_partials = nullptr;
    int* d_leaf_range = nullptr;
    int n_chunks = 0;

    DevState* d_state = nullptr;
    SrcDesc* d_src = nullptr;
    SrcDesc h_src_shadow{};
    bool src_valid = false;
    SrcDesc* h_src_ring = nullptr;
    int src_ring_pos = 0;

    // workspace (per batch size)
    int B = 0;
    std::vector<void*> ws_allocs;
    float *in_obs = nullptr, *in_act = nullptr, *in_rew = nullptr, *in_mask = nullptr, *in_nobs = nullptr;
    float* in_noise[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t* in_idx = nullptr;
    float *X_os = nullptr, *X_bc = nullptr, *X_eu = nullptr, *X_c1 = nullptr, *X_c2 = nullptr, *X_ct = nullptr;
    float *vel = nullptr, *w_rew = nullptr, *w_mask = nullptr, *w_act = nullptr, *tgt = nullptr;
    float *X_e0 = nullptr, *C0 = nullptr, *Abuf[2] = {nullptr, nullptr}, *Vpart = nullptr;  // fused Euler chain
    bool fused_euler = false, use_pec = false;
    // Euler-chain kernel (fql_chain.h): fragment-major copies of the BC flow's hidden kernels (layers 1..nh-1), of the 16 rows of
    // W0 that start at the action block and of the head kernel, refreshed behind every Adam step of that module
    bool use_chain = false;
    int fill_lane_full = 1;          // lane of the critic-loss / BC passes (and of their Adam launches) in the single-GPU programs
    std::vector<float*> wf_bc;       // per layer 1..nh-1: [H/4][H][4]
    float *wf_w0 = nullptr, *wf_w4 = nullptr;
    WfragTask* d_wfrag = nullptr;
    int wfrag_n = 0, wfrag_grid = 0;
    std::vector<void*> chain_allocs;
    bool split_build = false;   // building the data-parallel program: lane 1 must not depend on lane 0's backward
    bool wide_tiles = false;    // three-lane programs: 32 x 64 side tiles everywhere (the other lanes fill the CUs a launch leaves idle; +1 %)
    bool split_ok = false;
    int vp_tiles = 0;
    int pec_teams = 0;               // teams of the persistent chain (H/32 workgroups each, one workgroup per CU)
    unsigned* pec_epoch = nullptr;   // [teams] launch epochs of the persistent chain + 1 error word at the end
    fql_u64 *pec_g[2] = {nullptr, nullptr}, *pec_vg = nullptr;  // granule buffers: activations [B][H] x 2, head partials [T][B][16]
    int num_cus = 0;
    PassBuf p_os, p_os_bwd, p_bc, p_eu, p_c1[2], p_c2[2], p_ct[2];
    Program prog_fwdbwd, prog_opt, prog_loss;
    Program prog_opt_split;   // the optimizer half with the critic's / BC flow's Adam on lane 1: fql_update_end_split issues that lane on the stream that carried bucket 0's all-reduce
    Program prog_full;   // single-GPU update in ONE graph: per-module Adam launches start as soon as that module's gradients exist
    int mod_chunk0[3] = {0, 0, 0}, mod_chunkn[3] = {0, 0, 0};  // chunk ranges of bc_flow, onestep, critic (leaf order)
    // data-parallel variant: the same update as three single-lane graphs (lane 0 before / after it needs lane 1, lane 1)
    // so the gradient bucket of lane 1 can be all-reduced while lane 0 is still running
    Program prog_split;
    hipGraphExec_t split_exec[4] = {nullptr, nullptr, nullptr, nullptr};  // A0 (prep), B (lane 1), A1, A2
    hipGraph_t split_graph[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    bool split_begun = false;   // the pending update was begun by launch_split (ev_c marks the end of lane 0's pre-join part)
    bool began = false;
    // the single-GPU update as AQL packets on the engine's own HSA queues (fql_aql.h); everything else stays on HIP streams
    AqlRuntime aql;
    AqlProgram aql_full;
    std::vector<AqlDispatch>* aql_rec = nullptr;   // issue() records instead of launching while this is set
    bool last_update_aql = false;
    bool aql_tried = false;                        // build_aql has run for the current programs (lazily: data-parallel and graph-only users never start the HSA side)
    bool hip_dirty = true;                         // HIP work may be pending on the engine's stream: drain it before the next AQL submit
    void aql_drain() {                             // every AQL update submitted so far has finished (host wait)
        if (!aql.up) return;
        try { aql.drain(); } catch (const AqlError& e) { throw HipError{e.msg}; }
    }

    // dataset
    float *ds_obs = nullptr, *ds_act = nullptr, *ds_rew = nullptr, *ds_mask = nullptr, *ds_nobs = nullptr;
    float* ds_row = nullptr;
    int64_t ds_size = 0, ds_cap = 0, ds_ptr = 0;
    // frames dataset (visual agents): uint8 frames + per-row episode start, frame stacking and crop done by the gather
    unsigned char *ds_frames = nullptr, *ds_next_frames = nullptr;
    int64_t* ds_init = nullptr;
    int ds_fs = 0;
    float ds_paug = 0.f;
    // replay ring beside the training dataset (balanced sampling, main.py:106-109,255-259): same row layout, starts empty
    float *rb_obs = nullptr, *rb_act = nullptr, *rb_rew = nullptr, *rb_mask = nullptr, *rb_nobs = nullptr;
    unsigned char *rb_frames = nullptr, *rb_next_frames = nullptr;
    int64_t* rb_init = nullptr;
    int64_t rb_size = 0, rb_cap = 0, rb_ptr = 0;
    int64_t* in_init = nullptr;   // [B] workspace
    int *in_crop = nullptr, *in_crop_user = nullptr;  // [B][2]

    // eval (sample_actions / flow_actions) workspaces keyed by padded row count
    struct Eval {
        int n_pad = 0;
        std::vector<void*> allocs;
        float *X = nullptr, *Xf = nullptr, *tgt = nullptr;
        PassBuf p_os, p_eu;
        Program prog_os, prog_flow;
        float *st_obs = nullptr, *st_noise = nullptr, *st_out = nullptr;
        // visual agents: the rows are images; they pass through the module's encoder first (agents/fql.py:162-163, networks.py:221)
        unsigned char* st_img = nullptr;
        EncBuf eb_os, eb_bc;
        Program prog_enc_os, prog_enc_bc;
    };
    std::map<int, std::unique_ptr<Eval>> evals;

    // XCD-resident Euler chain (fql_xchain.h): the whole chain as one persistent launch on the critical lane
    bool use_xchain = false;
    unsigned* xsync = nullptr;      // [17][32] words: arrival flags, tickets (zeroed by the prep launch), sticky error word
    float* xvp = nullptr;           // head partials [32][B][16]
    size_t x_lds = 0;
    unsigned long long* x_stamps = nullptr;

    int64_t launches_per_update = 0;
    // lazily read infos (the reference returns device scalars that are only read at log time, main.py:276): a ring of pinned host
    // slots, one asynchronous 52-byte copy + event per ticket
    static constexpr int kInfoRing = 64;
    float (*h_info_ring)[16] = nullptr;
    hipEvent_t info_ev[kInfoRing] = {};
    uint64_t info_seq = 0;

    // ---------------------------------------------------------------------------------------
    float* dalloc(std::vector<void*>& owner, size_t nfloats) {
        void* p = nullptr;
        const size_t bytes = std::max<size_t>(nfloats, 4) * sizeof(float);
        HIP_CHECK(hipMalloc(&p, bytes));
        HIP_CHECK(hipMemset(p, 0, bytes));
        owner.push_back(p);
        return (float*)p;
    }

    void build_nets() {
        const int od = cfg.obs_dim, ad = cfg.act_dim;
        auto make = [&](int in, const int32_t* hid, int nh, int out, bool ln) {
            Net n;
            int prev = in;
            for (int i = 0; i <= nh; ++i) {
                Layer L{};
                L.in = prev;
                L.out = (i < nh) ? hid[i] : out;
                L.in_p = (i == 0) ? ((L.in + 63) & ~63) : pad16(L.in);
                L.out_p = pad16(L.out);
                L.ln = ln && i < nh;
                n.layers.push_back(L);
                prev = L.out;
            }
            return n;
        };
        nets[NET_C0] = make(od + ad, cfg.value_hidden, cfg.num_value_hidden, 1, cfg.layer_norm != 0);
        nets[NET_C1] = nets[NET_C0];
        nets[NET_BC] = make(od + ad + 1, cfg.actor_hidden, cfg.num_actor_hidden, ad, cfg.actor_layer_norm != 0);
        nets[NET_OS] = make(od + ad, cfg.actor_hidden, cfg.num_actor_hidden, ad, cfg.actor_layer_norm != 0);
        nets[NET_T0] = nets[NET_C0];
        nets[NET_T1] = nets[NET_C0];
        size_t off = 0;
        auto place_enc = [&](int ei) {
            if (!visual) return;
            EncNet& en = encs[ei];
            en = EncNet{};
            en.off = off;
            const int sizes[3] = {16, 32, 32};  // impala_small: stack_sizes (16, 32, 32), num_blocks 1 (utils/encoders.py:66-67,106)
            int H = cfg.img_h, W = cfg.img_w, cin = cfg.img_c;
            for (int s = 0; s < 3; ++s) {
                EncStack st;
                st.H = H; st.W = W;
                const int nconv = 1 + 2 * (cfg.encoder == 2 ? 2 : 1);   // impala: num_blocks 2, impala_small: 1 (utils/encoders.py:67,106)
                for (int j = 0; j < nconv; ++j) {
                    ConvL c{};
                    c.cin = cin; c.cout = sizes[s];
                    c.w = off; off += (size_t)9 * c.cin * c.cout;
                    c.b = off; off += c.cout;
                    st.conv.push_back(c);
                    cin = sizes[s];
                }
                en.stacks.push_back(st);
                H /= 2; W /= 2;
            }
            en.flat = H * W * cin;
            Layer& D = en.dense;
            D = Layer{};
            D.in = D.in_p = en.flat; D.out = D.out_p = enc_dim; D.ln = false;
            D.w = off; off += (size_t)D.in_p * D.out_p;
            D.b = off; off += D.out_p;
            en.size = off - en.off;
        };
        for (int ni = 0; ni < NUM_NETS; ++ni) {
            Net& n = nets[ni];
            n.off = off;
            for (Layer& L : n.layers) {
                L.w = off; off += (size_t)L.in_p * L.out_p;
                L.b = off; off += L.out_p;
                if (L.ln) {
                    L.g = off; off += L.out_p;
                    L.be = off; off += L.out_p;
                }
            }
            n.size = off - n.off;
            // every module's encoder follows its MLP(s); the critic's sits inside the Polyak-averaged region
            if (ni == NET_C1) { place_enc(ENC_C); critic_size = off; }
            if (ni == NET_BC) place_enc(ENC_BC);
            if (ni == NET_OS) { place_enc(ENC_OS); n_train = off; }
            if (ni == NET_T1) place_enc(ENC_T);
        }
        n_total = off;
        for (int ni = 0; ni < NUM_NETS; ++ni)
            for (const Layer& L : nets[ni].layers)
                if (L.in_p > 1024 || L.out_p > 1024) invalid("layer widths above 1024 are not supported (got %d -> %d)", L.in, L.out);
        if (nets[NET_T0].off - n_train != 0 || nets[NET_T0].size + nets[NET_T1].size + (visual ? encs[ENC_T].size : 0) != critic_size)
            invalid("internal: target arena layout mismatch");
    }

    // can the first convolution of a stack run fused with the stack's max-pool (fql_conv3x3_pool_kernel)?  first_stack: it reads the uint8 images
    // Measured (visual update, one box): the uint8 layer fused +2.8 % (fp32) and more in bf16x3; the float layers of stacks 1 / 2 fused too: bf16x3 370.5 -> 373.9, fp32
    // 311.7 -> 308.3 (their 5-row tiles redo a quarter of the rows, which fp32 MFMAs pay for) - so those are fused under precision = 2 only (FQL_FUSE_POOL_FLOAT=0/1 overrides).
    bool pool_fusable(const EncStack& st, const ConvL& c, bool first_stack) const {
        constexpr bool on = true;
        constexpr int float_env = -1;
        const bool on_float = float_env >= 0 ? float_env != 0 : cfg.precision == 2;
        const int Ci = pad16c(c.cin);
        if (!on || st.W % 16 || st.W < 16 || st.W > 128 || st.H % 2 || (c.cout != 16 && c.cout != 32)) return false;
        if ((size_t)FQL_CONV_POOL_LDS_FLOATS(st.W, Ci, c.cout) * sizeof(float) > 65536) return false;
        if (first_stack) return c.cout == 16 && Ci == 16 && 7 * st.W * c.cin <= 4 * 4 * FQL_THREADS && (st.W * c.cin) % 4 == 0;   // seven uint8 rows: <= 4 dwords per thread
        return on_float && (Ci == 16 || Ci == 32) && Ci == c.cin && 7 * (st.W + 2) * (Ci / 4) <= 4 * FQL_THREADS;               // seven float rows: <= 4 float4 per thread
    }
    // LDS-layout weight copies of every convolution + the task tables of fql_conv_wprep_kernel (needs P)
    void build_enc_weights() {
        if (!visual) return;
        for (int ei = 0; ei < NUM_ENC; ++ei) {
            std::vector<ConvWprepTask> tasks;
            bool first = true;
            for (EncStack& st : encs[ei].stacks)
                for (ConvL& c : st.conv) {
                    const int Ci = pad16c(c.cin);
                    const bool first_stack = &st == &encs[ei].stacks[0];
                    // precision = 2: the float convolutions read pre-split bf16 hi / lo planes (slightly larger than the fp32 image)
                    const bool split = cfg.precision == 2 && !first && (Ci == 16 || Ci == 32) && (c.cout == 16 || c.cout == 32);
                    // a stack's first convolution runs fused with the stack's max-pool and splits its fragments in registers: its forward copy stays fp32
                    c.pool_fused = (&c == &st.conv[0]) && pool_fusable(st, c, first_stack);
                    c.Wf = dalloc(enc_allocs, std::max((size_t)c.cout * (9 * Ci + 4), (size_t)2 * c.cout * (9 * Ci / 2 + Ci / 4)));
                    c.Wb = first ? nullptr : dalloc(enc_allocs, std::max((size_t)c.cin * (9 * c.cout + 4), (size_t)2 * c.cin * (9 * c.cout / 2 + c.cout / 4)));   // no gradient into the images
                    c.split = split;
                    tasks.push_back(ConvWprepTask{P + c.w, c.Wf, c.Wb, c.cin, c.cout, Ci, split ? (c.pool_fused ? 2 : 1) : 0});
                    first = false;
                }
            enc_nconv[ei] = (int)tasks.size();
            enc_wprep[ei] = (ConvWprepTask*)dalloc(enc_allocs, tasks.size() * sizeof(ConvWprepTask) / sizeof(float) + 4);
            HIP_CHECK(hipMemcpy(enc_wprep[ei], tasks.data(), tasks.size() * sizeof(ConvWprepTask), hipMemcpyHostToDevice));
        }
    }

    // Fragment-major weight copies for fql_chain_kernel (needs P): BC flow with >= 3 equal hidden layers of width 256 / 512
    void build_chain_weights() {
        const Net& n = nets[NET_BC];
        const int nh = n.nl() - 1;
        const int H = n.layers[0].out_p;
        bool ok = !cfg.actor_layer_norm && cfg.act_dim <= 15 && nh >= 3 && (H == 256 || H == 512);
        for (int l = 0; l < nh && ok; ++l) ok = n.layers[l].out == H && n.layers[l].out_p == H;
        use_chain = ok;
        if (!ok) return;
        const int od = cfg.obs_dim, ap = pad16(cfg.act_dim);
        std::vector<WfragTask> tasks;
        int tile = 0;
        const bool split = cfg.precision == 2;   // fql_chain_split_kernel reads pre-split bf16 hi / lo operands (same bytes)
        auto add = [&](const float* src, int K, int N, int ld, int kvalid, int split_mode) {
            float* dst = dalloc(chain_allocs, (size_t)K * N);
            tasks.push_back(WfragTask{src, dst, K, N, ld, kvalid, tile, split ? split_mode : 0});
            tile += ((K / 4) * N + FQL_THREADS - 1) / FQL_THREADS;
            return dst;
        };
        wf_bc.clear();
        for (int l = 1; l < nh; ++l) wf_bc.push_back(add(P + n.layers[l].w, H, H, H, H, 1));
        const Layer& l0 = n.layers[0];
        wf_w0 = add(P + l0.w + (size_t)od * l0.out_p, 16, H, l0.out_p, std::min(16, l0.in_p - od), 2);   // rows of (action block, t, padding)
        wf_w4 = add(P + n.layers[nh].w, H, ap, ap, H, 1);
        wfrag_n = (int)tasks.size(); wfrag_grid = tile;
        d_wfrag = (WfragTask*)dalloc(chain_allocs, tasks.size() * sizeof(WfragTask) / sizeof(float) + 4);
        HIP_CHECK(hipMemcpy(d_wfrag, tasks.data(), tasks.size() * sizeof(WfragTask), hipMemcpyHostToDevice));
    }
    // after any out-of-graph parameter write (init, fql_set_param): bring the copies up to date
    void refresh_chain_weights(hipStream_t s) {
        if (!use_chain) return;
        hipLaunchKernelGGL(fql_wfrag_kernel, dim3(wfrag_grid), dim3(FQL_THREADS), 0, s, (const WfragTask*)d_wfrag, wfrag_n, -1);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s));
    }

    void build_leaves() {
        leaves.clear();
        struct Mod { const char* name; const char* sub; int n0, n1; bool train; int enc; };
        const Mod mods[4] = {{"modules_actor_bc_flow", "mlp", NET_BC, -1, true, ENC_BC},
                             {"modules_actor_onestep_flow", "mlp", NET_OS, -1, true, ENC_OS},
                             {"modules_critic", "value_net", NET_C0, NET_C1, true, ENC_C},
                             {"modules_target_critic", "value_net", NET_T0, NET_T1, false, ENC_T}};
        int tid = 0;
        for (const Mod& m : mods) {
            const Net& n0 = nets[m.n0];
            const bool ens = m.n1 >= 0;
            if (visual) {
                // "<module>/encoder/..." sorts before "mlp" / "value_net"; inside: MLP_0 < stack_blocks_* (flax auto names,
                // utils/encoders.py:72-79,98); the encoder is NOT ensembled (utils/networks.py:186-187: applied before value_net)
                const EncNet& en = encs[m.enc];
                auto add_plain = [&](const std::string& name, std::vector<int64_t> shape, size_t off, int rows, int cols) {
                    Leaf lf;
                    lf.name = name;
                    lf.ndim = (int)shape.size();
                    for (int k = 0; k < 4; ++k) lf.shape[k] = k < lf.ndim ? shape[k] : 1;
                    lf.segs.push_back(Segment{off, rows, cols, rows, cols});
                    lf.trainable = m.train;
                    lf.train_id = m.train ? tid++ : -1;
                    leaves.push_back(lf);
                };
                const std::string eb = std::string(m.name) + "/encoder/";
                add_plain(eb + "MLP_0/Dense_0/bias", {en.dense.out}, en.dense.b, 1, en.dense.out);
                add_plain(eb + "MLP_0/Dense_0/kernel", {en.dense.in, en.dense.out}, en.dense.w, en.dense.in, en.dense.out);
                for (size_t s = 0; s < en.stacks.size(); ++s)
                    for (size_t j = 0; j < en.stacks[s].conv.size(); ++j) {
                        const ConvL& c = en.stacks[s].conv[j];
                        const std::string cb = eb + "stack_blocks_" + std::to_string(s) + "/Conv_" + std::to_string(j);
                        add_plain(cb + "/bias", {c.cout}, c.b, 1, c.cout);
                        add_plain(cb + "/kernel", {3, 3, c.cin, c.cout}, c.w, 9 * c.cin, c.cout);
                    }
            }
            auto add = [&](const std::string& name, bool matrix, int rows, int cols, size_t o0, size_t o1, int rp, int cp) {
                Leaf lf;
                lf.name = name;
                int d = 0;
                if (ens) lf.shape[d++] = 2;
                if (matrix) lf.shape[d++] = rows;
                lf.shape[d++] = cols;
                lf.ndim = d;
                for (int k = d; k < 4; ++k) lf.shape[k] = 1;
                lf.segs.push_back(Segment{o0, matrix ? rows : 1, cols, matrix ? rp : 1, cp});
                if (ens) lf.segs.push_back(Segment{o1, matrix ? rows : 1, cols, matrix ? rp : 1, cp});
                lf.trainable = m.train;
                lf.train_id = m.train ? tid++ : -1;
                leaves.push_back(lf);
            };
            const std::string base = std::string(m.name) + "/" + m.sub + "/";
            // jax dict order: Dense_0..Dense_L (bias, kernel), then LayerNorm_0.. (bias, scale)
            for (int i = 0; i < n0.nl(); ++i) {
                const Layer& a = n0.layers[i];
                const Layer* b = ens ? &nets[m.n1].layers[i] : nullptr;
                add(base + "Dense_" + std::to_string(i) + "/bias", false, 1, a.out, a.b, b ? b->b : 0, 1, a.out_p);
                add(base + "Dense_" + std::to_string(i) + "/kernel", true, a.in, a.out, a.w, b ? b->w : 0, a.in_p, a.out_p);
            }
            for (int i = 0; i < n0.nl(); ++i) {
                const Layer& a = n0.layers[i];
                if (!a.ln) continue;
                const Layer* b = ens ? &nets[m.n1].layers[i] : nullptr;
                add(base + "LayerNorm_" + std::to_string(i) + "/bias", false, 1, a.out, a.be, b ? b->be : 0, 1, a.out_p);
                add(base + "LayerNorm_" + std::to_string(i) + "/scale", false, 1, a.out, a.g, b ? b->g : 0, 1, a.out_p);
            }
        }
        n_train_leaves = tid;
        if (n_train_leaves > 256) invalid("too many trainable leaves (%d)", n_train_leaves);
        // Adam chunks: contiguous pieces of one leaf, <= 4096 elements
        std::vector<AdamChunk> ch;
        for (const Leaf& lf : leaves) {
            if (!lf.trainable) continue;
            for (const Segment& s : lf.segs) {
                const size_t len = (size_t)s.rows_p * s.cols_p;
                for (size_t o = 0; o < len; o += 4096)
                    ch.push_back(AdamChunk{(int)(s.off + o), (int)std::min<size_t>(4096, len - o), lf.train_id});
            }
        }
        n_chunks = (int)ch.size();
        HIP_CHECK(hipMalloc((void**)&d_chunks, ch.size() * sizeof(AdamChunk)));
        HIP_CHECK(hipMemcpy(d_chunks, ch.data(), ch.size() * sizeof(AdamChunk), hipMemcpyHostToDevice));
        std::vector<int> range(n_train_leaves + 1, 0);  // chunks are emitted leaf by leaf: contiguous ranges
        for (const AdamChunk& c : ch) range[c.leaf + 1]++;
        for (int i = 0; i < n_train_leaves; ++i) range[i + 1] += range[i];
        {   // leaves (and so chunks) are ordered module by module: actor_bc_flow, actor_onestep_flow, critic
            const char* mods[3] = {"modules_actor_bc_flow/", "modules_actor_onestep_flow/", "modules_critic/"};
            for (int m = 0; m < 3; ++m) {
                int lo = 1 << 30, hi = -1;
                for (const Leaf& lf : leaves)
                    if (lf.trainable && lf.name.rfind(mods[m], 0) == 0) { lo = std::min(lo, lf.train_id); hi = std::max(hi, lf.train_id); }
                mod_chunk0[m] = range[lo];
                mod_chunkn[m] = range[hi + 1] - range[lo];
            }
        }
        HIP_CHECK(hipMalloc((void**)&d_leaf_range, range.size() * sizeof(int)));
        HIP_CHECK(hipMemcpy(d_leaf_range, range.data(), range.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc((void**)&d_partials, (size_t)n_chunks * 4 * sizeof(float)));
        HIP_CHECK(hipMemset(d_partials, 0, (size_t)n_chunks * 4 * sizeof(float)));
    }

    const Leaf* find_leaf(const char* name) const {
        for (const Leaf& l : leaves)
            if (l.name == name) return &l;
        return nullptr;
    }
    static size_t leaf_count(const Leaf& l) {
        size_t n = 1;
        for (int i = 0; i < l.ndim; ++i) n *= (size_t)l.shape[i];
        return n;
    }
    void leaf_io(const Leaf& lf, float* arena, float* host, bool to_device) {
        HIP_CHECK(hipStreamSynchronize(stream));
        size_t ho = 0;
        for (const Segment& s : lf.segs) {
            HIP_CHECK(hipMemcpy2D(to_device ? (void*)(arena + s.off) : (void*)(host + ho), (to_device ? s.cols_p : s.cols) * sizeof(float),
                                  to_device ? (const void*)(host + ho) : (const void*)(arena + s.off),
                                  (to_device ? s.cols : s.cols_p) * sizeof(float), s.cols * sizeof(float), s.rows,
                                  to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
            ho += (size_t)s.rows * s.cols;
        }
    }

    void init_params() {
        // utils/networks.py:9-11 Glorot-uniform kernels; flax Dense bias 0; LayerNorm scale 1 / bias 0;
        // agents/fql.py:241-242 target := critic.  (Host RNG: JAX's threefry init is not reproducible.)
        std::vector<float> h(n_total, 0.f);
        std::mt19937_64 gen(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull);
        for (int ni = 0; ni <= NET_OS; ++ni)
            for (const Layer& L : nets[ni].layers) {
                const double lim = std::sqrt(6.0 / (double)(L.in + L.out));
                std::uniform_real_distribution<double> U(-lim, lim);
                for (int r = 0; r < L.in; ++r)
                    for (int c = 0; c < L.out; ++c) h[L.w + (size_t)r * L.out_p + c] = (float)U(gen);
                if (L.ln)
                    for (int c = 0; c < L.out; ++c) h[L.g + c] = 1.0f;
            }
        if (visual)
            for (int ei = 0; ei <= ENC_OS; ++ei) {   // xavier_uniform convolutions (utils/encoders.py:18), Glorot Dense, zero biases
                const EncNet& en = encs[ei];
                for (const EncStack& st : en.stacks)
                    for (const ConvL& c : st.conv) {
                        const double lim = std::sqrt(6.0 / (9.0 * c.cin + 9.0 * c.cout));
                        std::uniform_real_distribution<double> U(-lim, lim);
                        for (size_t i = 0; i < (size_t)9 * c.cin * c.cout; ++i) h[c.w + i] = (float)U(gen);
                    }
                const Layer& D = en.dense;
                const double lim = std::sqrt(6.0 / (double)(D.in + D.out));
                std::uniform_real_distribution<double> U(-lim, lim);
                for (size_t i = 0; i < (size_t)D.in * D.out; ++i) h[D.w + i] = (float)U(gen);
            }
        std::memcpy(h.data() + n_train, h.data(), critic_size * sizeof(float));
        HIP_CHECK(hipMemcpy(P, h.data(), n_total * sizeof(float), hipMemcpyHostToDevice));
    }

    // ---------------------------------------------------------------------------------------
    // pass buffers
    // ---------------------------------------------------------------------------------------
    PassBuf make_pass(std::vector<void*>& owner, int net, int M, float* x0, bool bwd, bool input_grad) {
        const Net& n = nets[net];
        PassBuf p;
        p.net = net; p.M = M; p.x0 = x0;
        const int L = n.nl() - 1;
        for (int l = 0; l < L; ++l) {
            const int H = n.layers[l].out_p;
            p.g.push_back(dalloc(owner, (size_t)M * H));
            p.z.push_back(dalloc(owner, (size_t)M * H));
            if (n.layers[l].ln) {
                p.xn.push_back(dalloc(owner, (size_t)M * H));
                p.stats.push_back(dalloc(owner, (size_t)M * 2));
                p.lnpart.push_back(dalloc(owner, (size_t)M * ((2 * ((H + 31) / 32) + 3) & ~3)));   // (sum, sum sq) per 32 columns
            } else {
                p.xn.push_back(p.g.back());
                p.stats.push_back(nullptr);
                p.lnpart.push_back(nullptr);
            }
        }
        p.out = dalloc(owner, (size_t)M * n.layers[L].out_p);
        if (bwd) {
            for (int l = 0; l <= L; ++l) p.dz.push_back(dalloc(owner, (size_t)M * n.layers[l].out_p));
            for (int l = 0; l < L; ++l) p.dy.push_back(n.layers[l].ln ? dalloc(owner, (size_t)M * n.layers[l].out_p) : nullptr);
            if (input_grad) p.dx0 = dalloc(owner, (size_t)M * n.in_p());
        }
        return p;
    }

    // ---------------------------------------------------------------------------------------
    // visual path: encoder passes (utils/encoders.py:61-100) as conv / pool / GEMM ops of the same program
    // ---------------------------------------------------------------------------------------
    EncBuf make_enc_buf(std::vector<void*>& owner, int enc, int n, const unsigned char* img, bool bwd) {
        const EncNet& en = encs[enc];
        EncBuf b;
        b.enc = enc; b.n = n; b.img = img;
        size_t maxel = 0;
        for (const EncStack& st : en.stacks) {
            EncBuf::St s;
            const int C = st.conv[0].cout;
            const size_t full = (size_t)n * st.H * st.W * C, quarter = full / 4;
            maxel = std::max(maxel, full);
            s.c0 = dalloc(owner, full);
            s.pool = dalloc(owner, quarter);
            s.arg = (unsigned char*)dalloc(owner, quarter / 4 + 4);
            for (size_t j = 1; j + 1 < st.conv.size(); j += 2) {
                s.c1.push_back(dalloc(owner, quarter));
                s.y.push_back(dalloc(owner, quarter));
            }
            b.st.push_back(s);
        }
        b.frelu = dalloc(owner, (size_t)n * en.flat);
        b.z = dalloc(owner, (size_t)n * enc_dim);
        b.E = dalloc(owner, (size_t)n * enc_dim);
        if (bwd) {
            b.dz = dalloc(owner, (size_t)n * enc_dim);
            b.dA = dalloc(owner, maxel); b.dB = dalloc(owner, maxel); b.dC = dalloc(owner, maxel);
            for (const EncStack& st : en.stacks)
                for (const ConvL& c : st.conv) b.wpart[P + c.w] = dalloc(owner, (size_t)256 * (9 * pad16c(c.cin) + 1) * c.cout);
        }
        return b;
    }
    // image rows per workgroup: <= 8 MFMA row tiles of 16 pixels, whole rows, LDS within 64 KB
    static int conv_rows(int H, int W, int Ci, int Co, bool wgrad) {
        for (int R = std::min(H, std::max(1, 128 / W)); R >= 1; R >>= 1) {
            if (H % R || (R * W) % 16) continue;
            const size_t fl = wgrad ? (size_t)(R + 2) * (W + 2) * (Ci + 4) + (size_t)R * W * (Co + 4)
                                    : (size_t)(R + 2) * (W + 2) * (Ci + 4) + (size_t)Co * (9 * Ci + 4);
            if (fl * sizeof(float) <= 65536) return R;
        }
        invalid("convolution tile does not fit (H %d, W %d, channels %d -> %d)", H, W, Ci, Co);
    }
    void emit_conv(Program& pr, const void* in, const void* in_id, int in_mode, int n, int H, int W, const ConvL& c, bool transposed,
                   float* out, const void* out_id, float* out_relu, const float* mask, const void* mask_id, const float* add, const void* add_id) {
        Op op{};
        op.type = in_mode == 2 ? OP_CONV_U8 : OP_CONV;
        if (in_mode == 2 && (transposed || c.cout != 16)) invalid("uint8 convolution kernel: 16 output channels expected (got %d)", c.cout);
        ConvArgs& a = op.conv;
        a.in = in; a.Wl = transposed ? c.Wb : c.Wf; a.bias = transposed ? nullptr : P + c.b;
        a.out = out; a.out_relu = out_relu; a.mask = mask; a.add = add;
        a.N = n; a.H = H; a.W = W;
        if (!transposed) { a.Ci = pad16c(c.cin); a.Ci_real = c.cin; a.Co = c.cout; }
        else { a.Ci = c.cout; a.Ci_real = c.cout; a.Co = c.cin; }
        a.in_mode = in_mode; a.transposed = transposed ? 1 : 0;
        a.R = conv_rows(H, W, a.Ci, a.Co, false);
        op.reads = {in_id, a.Wl};
        if (!transposed) op.reads.push_back(P + c.b);
        if (mask) op.reads.push_back(mask_id);
        if (add) op.reads.push_back(add_id);
        op.writes = {out_id};
        if (out_relu) op.writes.push_back(out_relu);
        push(pr, op);
    }
    void emit_encoder_forward(Program& pr, EncBuf& b) {
        const EncNet& en = encs[b.enc];
        const int n = b.n;
        {   // this pass's weights in LDS layout (forward and data-gradient forms), one small launch per encoder pass
            Op op{};
            op.type = OP_CONV_WPREP;
            op.wprep_tasks = enc_wprep[b.enc];
            op.wprep_n = enc_nconv[b.enc];
            for (const EncStack& st : en.stacks)
                for (const ConvL& c : st.conv) {
                    op.reads.push_back(P + c.w);
                    op.writes.push_back(c.Wf);
                    if (c.Wb) op.writes.push_back(c.Wb);
                }
            push(pr, op);
        }
        const float* x = nullptr;
        for (size_t s = 0; s < en.stacks.size(); ++s) {
            const EncStack& st = en.stacks[s];
            EncBuf::St& bs = b.st[s];
            const int C = st.conv[0].cout, H2 = st.H / 2, W2 = st.W / 2;
            // a stack's first convolution and its max-pool as one kernel: the pre-pool tensor (for stack 0 the largest write of the update) is never materialised
            const bool fuse_pool = st.conv[0].pool_fused;
            if (s == 0) emit_conv(pr, b.img, b.img, 2, n, st.H, st.W, st.conv[0], false, bs.c0, bs.c0, nullptr, nullptr, nullptr, nullptr, nullptr);
            else emit_conv(pr, x, x, 0, n, st.H, st.W, st.conv[0], false, bs.c0, bs.c0, nullptr, nullptr, nullptr, nullptr, nullptr);
            if (fuse_pool) {
                Op& co = pr.ops.back();
                co.conv.out = bs.pool; co.conv.parg = bs.arg;
                co.writes = {bs.pool, bs.arg};
            } else
            {
                Op op{};
                op.type = OP_POOL;
                op.pool = PoolArgs{bs.c0, bs.pool, bs.arg, n, st.H, st.W, C};
                op.reads = {bs.c0};
                op.writes = {bs.pool, bs.arg};
                push(pr, op);
            }
            const float* y = bs.pool;
            for (size_t blk = 0; blk < bs.c1.size(); ++blk) {
                const bool last = s + 1 == en.stacks.size() && blk + 1 == bs.c1.size();
                emit_conv(pr, y, y, 1, n, H2, W2, st.conv[1 + 2 * blk], false, bs.c1[blk], bs.c1[blk], nullptr, nullptr, nullptr, nullptr, nullptr);
                emit_conv(pr, bs.c1[blk], bs.c1[blk], 1, n, H2, W2, st.conv[2 + 2 * blk], false, bs.y[blk], bs.y[blk], last ? b.frelu : nullptr,
                          nullptr, nullptr, y, y);
                y = bs.y[blk];
            }
            x = y;
        }
        Op op{};   // MLP((512,), activate_final=True): Dense + GELU (utils/encoders.py:98)
        op.type = OP_GEMM;
        GemmTask& t = op.gemm;
        const Layer& D = en.dense;
        t.A = b.frelu; t.lda = D.in_p;
        t.B = P + D.w; t.ldb = D.out_p; t.bias = P + D.b;
        t.C = b.E; t.ldc = D.out_p; t.Zout = b.z;
        t.M = n; t.N = D.out_p; t.K = D.in_p;
        t.flags = GF_BIAS | GF_GELU | GF_SAVE_Z;
        op.reads = {t.A, t.B};
        op.writes = {t.C, t.Zout};
        // (the LDS-tiled kernel whichever lane this pass is on: K = flat does not fit the 16-row kernel's A tile)
        if (t.M % 64 == 0 && t.N % 64 == 0 && t.K % 64 == 0) op.type = OP_GEMM64;
        else if (t.K > 1024) invalid("encoder Dense with %d inputs needs a batch that is a multiple of 64", t.K);
        push(pr, op);
    }
    void emit_conv_wgrad(Program& pr, EncBuf& b, const void* in, const void* in_id, int in_mode, int n, int H, int W, const ConvL& c,
                         const float* dout, const void* dout_id) {
        Op op{};
        op.type = OP_CONV_WGRAD;
        ConvWgradArgs& a = op.cw;
        float* wp = b.wpart.at(P + c.w);
        a.in = in; a.dout = dout; a.partial = wp;
        a.N = n; a.H = H; a.W = W; a.Ci = pad16c(c.cin); a.Ci_real = c.cin; a.Co = c.cout; a.in_mode = in_mode;
        if (in_mode == 2 && (a.Ci != 16 || a.Co != 16)) invalid("uint8 convolution weight gradient: <= 16 input channels and 16 output channels expected (got %d -> %d)", c.cin, c.cout);
        a.R = conv_rows(H, W, a.Ci, a.Co, true);
        a.nblocks = n * (H / a.R);
        constexpr int cwg_env = 256;   // persistent workgroups per weight-gradient task
        op.cw_grid = std::min(a.nblocks, cwg_env);
        op.reads = {in_id, dout_id};
        op.writes = {wp};
        push(pr, op);
        Op r{};
        r.type = OP_CONV_WRED;
        r.cwr = ConvWredArgs{wp, G + c.w, G + c.b, op.cw_grid, a.Ci, a.Co, c.cin, 0};
        r.reads = {wp};
        r.writes = {G + c.w, G + c.b};
        push(pr, r);
    }
    // backward of images [img0, img0 + n) of pass b; the encoding's gradient = first enc_dim columns of dxa (+ dxb)
    void emit_encoder_backward(Program& pr, EncBuf& b, int img0, int n, const float* dxa, const float* dxb, int ld, const void* align_with = nullptr) {
        const EncNet& en = encs[b.enc];
        const Layer& D = en.dense;
        {
            Op op{};
            op.type = OP_ENC_DZ;
            op.edz = EncDzArgs{dxa, dxb, b.z + (size_t)img0 * enc_dim, b.dz, n, enc_dim, ld};
            op.reads = {dxa, b.z};
            if (dxb) op.reads.push_back(dxb);
            // level alignment: the three encoder backward passes start together so that their per-level convolutions,
            // weight gradients and partial folds share launches (the lane is serial either way)
            if (align_with) op.reads.push_back(align_with);
            op.writes = {b.dz};
            push(pr, op);
        }
        {
            Op op{};
            op.type = OP_WGRAD;
            WgradTask& w = op.wgrad;
            w.X = b.frelu + (size_t)img0 * en.flat; w.ldx = D.in_p;
            w.dZ = b.dz; w.ldz = D.out_p;
            w.dW = G + D.w; w.ldw = D.out_p; w.db = G + D.b;
            w.M = n; w.Kin = D.in_p; w.N = D.out_p;
            op.reads = {b.frelu, b.dz};
            op.writes = {w.dW, w.db};
            push(pr, op);
        }
        const EncStack& lst = en.stacks.back();
        const float* xfinal = b.st.back().y.empty() ? b.st.back().pool : b.st.back().y.back();
        {
            Op op{};   // d(flat) = dz W^T, through the final ReLU (utils/encoders.py:92)
            op.type = OP_GEMM;
            GemmTask& t = op.gemm;
            t.A = b.dz; t.lda = D.out_p;
            t.B = P + D.w; t.ldb = D.out_p;
            t.C = b.dA; t.ldc = D.in_p;
            t.M = n; t.N = D.in_p; t.K = D.out_p;
            t.flags = GF_TRANS_B | GF_RELUGRAD;
            t.Zprev = xfinal + (size_t)img0 * en.flat;
            op.reads = {b.dz, t.B, xfinal};
            op.writes = {b.dA};
            if (t.M % 64 == 0 && t.N % 64 == 0 && t.K % 64 == 0) op.type = OP_GEMM64;
            push(pr, op);
        }
        (void)lst;
        float* dy = b.dA;
        std::vector<float*> freeb = {b.dB, b.dC};
        auto take = [&]() { float* p = freeb.back(); freeb.pop_back(); return p; };
        for (int s = (int)en.stacks.size() - 1; s >= 0; --s) {
            const EncStack& st = en.stacks[s];
            EncBuf::St& bs = b.st[s];
            const int C = st.conv[0].cout, H2 = st.H / 2, W2 = st.W / 2;
            const size_t q_img = (size_t)H2 * W2 * C, f_img = (size_t)st.H * st.W * C;
            for (int blk = (int)bs.c1.size() - 1; blk >= 0; --blk) {
                const float* inp = blk == 0 ? bs.pool : bs.y[blk - 1];
                const float* c1 = bs.c1[blk];
                const float* inp_o = inp + (size_t)img0 * q_img;
                const float* c1_o = c1 + (size_t)img0 * q_img;
                emit_conv_wgrad(pr, b, c1_o, c1, 1, n, H2, W2, st.conv[2 + 2 * blk], dy, dy);
                float* d_c1 = take();
                emit_conv(pr, dy, dy, 0, n, H2, W2, st.conv[2 + 2 * blk], true, d_c1, d_c1, nullptr, c1_o, c1, nullptr, nullptr);
                emit_conv_wgrad(pr, b, inp_o, inp, 1, n, H2, W2, st.conv[1 + 2 * blk], d_c1, d_c1);
                float* dy_new = take();
                emit_conv(pr, d_c1, d_c1, 0, n, H2, W2, st.conv[1 + 2 * blk], true, dy_new, dy_new, nullptr, inp_o, inp, dy, dy);
                freeb.push_back(d_c1); freeb.push_back(dy);
                dy = dy_new;
            }
            float* d_c0 = take();
            {
                Op op{};
                op.type = OP_POOL_BWD;
                op.poolb = PoolBwdArgs{dy, bs.arg + (size_t)img0 * q_img, d_c0, n, st.H, st.W, C};
                op.reads = {dy, bs.arg};
                op.writes = {d_c0};
                push(pr, op);
            }
            if (s == 0) {
                const size_t ib = (size_t)st.H * st.W * st.conv[0].cin;
                emit_conv_wgrad(pr, b, b.img + (size_t)img0 * ib, b.img, 2, n, st.H, st.W, st.conv[0], d_c0, d_c0);
            } else {
                const EncBuf::St& ps = b.st[s - 1];
                const float* xin = ps.y.empty() ? ps.pool : ps.y.back();
                const size_t xi = (size_t)st.H * st.W * st.conv[0].cin;
                emit_conv_wgrad(pr, b, xin + (size_t)img0 * xi, xin, 0, n, st.H, st.W, st.conv[0], d_c0, d_c0);
                float* dprev = take();
                emit_conv(pr, d_c0, d_c0, 0, n, st.H, st.W, st.conv[0], true, dprev, dprev, nullptr, nullptr, nullptr, nullptr, nullptr);
                freeb.push_back(dy);
                dy = dprev;
            }
            freeb.push_back(d_c0);
            (void)f_img;
        }
    }

    // throughput-lane tasks with 64-aligned shapes go to the 64x64 LDS-tiled kernel
    bool want64(int M, int N, int K, int flags) const {
        constexpr bool off = false;
        if (off || emit_lane == 0 || !allow64) return false;
        if (M % 64 || N % 64 || K % 64) return false;
        if (flags & (GF_EULER | GF_CLIP_OUT)) return false;
        return true;
    }

    // per-pass placement: lane and kernel family (FQL_LANE_<pass> overrides the lane: tests/test_gpu_streams.py provokes the refusal of a two-way edge with it)
    void place(const char* pass, int default_lane, bool default64) {
        char key[64];
        snprintf(key, sizeof key, "FQL_LANE_%s", pass);
        const char* e = getenv(key);
        emit_lane = e ? atoi(e) : default_lane;
        allow64 = default64;
    }

    void push(Program& pr, Op& op) {
        op.lane = emit_lane;
        pr.ops.push_back(op);
    }

    // forward of one pass appended to a program.  final_flags: epilogue of the last layer.
    void emit_forward(Program& pr, const PassBuf& p, bool save, int final_flags = 0, float* aux = nullptr,
                      float* aux2 = nullptr, float f0 = 0.f, float f1 = 0.f) {
        const Net& n = nets[p.net];
        const int L = n.nl() - 1;
        for (int l = 0; l <= L; ++l) {
            const Layer& ly = n.layers[l];
            Op op{};
            op.type = OP_GEMM;
            GemmTask& t = op.gemm;
            const bool a_ln = l > 0 && n.layers[l - 1].ln;
            t.A = (l == 0) ? p.x0 : p.g[l - 1];
            t.lda = ly.in_p;
            t.B = P + ly.w; t.ldb = ly.out_p;
            t.bias = P + ly.b;
            t.M = p.M; t.N = ly.out_p; t.K = ly.in_p;
            t.ldc = ly.out_p;
            t.flags = GF_BIAS;
            op.reads = {t.A, t.B};
            if (a_ln) {
                t.flags |= GF_A_LN;
                t.ln_g = P + n.layers[l - 1].g; t.ln_b = P + n.layers[l - 1].be;
                t.ln_width = n.layers[l - 1].out;
                if (save) {
                    t.flags |= GF_LN_WRITE;
                    t.ln_xout = p.xn[l - 1]; t.ln_stats = p.stats[l - 1];
                    op.writes.push_back(t.ln_xout);
                    op.writes.push_back(t.ln_stats);
                }
            }
            if (l < L) {
                t.C = p.g[l];
                t.flags |= GF_GELU;
                if (save) { t.flags |= GF_SAVE_Z; t.Zout = p.z[l]; op.writes.push_back(t.Zout); }
                op.writes.push_back(t.C);
                // LDS-tiled kernel; a LayerNorm'd A operand needs the producing layer's partial sums, i.e. that layer on it too
                const bool prev64 = l > 0 && want64(p.M, n.layers[l - 1].out_p, n.layers[l - 1].in_p, GF_BIAS | GF_GELU);
                if (want64(t.M, t.N, t.K, t.flags) && (!a_ln || prev64)) {
                    op.type = OP_GEMM64;
                    if (a_ln) { t.aux2 = p.lnpart[l - 1]; t.i0 = t.K / 32; op.reads.push_back(t.aux2); }
                    if (ly.ln) { t.flags |= GF_LN_PART; t.aux = p.lnpart[l]; t.i1 = t.N / 32; op.writes.push_back(t.aux); }
                }
            } else {
                t.C = p.out;
                t.flags |= final_flags;
                if (final_flags & GF_EULER) {
                    t.aux = aux; t.aux2 = aux2;
                    t.i0 = n.in_p(); t.i1 = cfg.obs_dim; t.i2 = cfg.act_dim;
                    t.f0 = f0; t.f1 = f1;
                    op.reads.push_back(aux);
                    op.writes.push_back(aux);
                    if (aux2) op.writes.push_back(aux2);
                } else {
                    op.writes.push_back(t.C);
                    if (final_flags & GF_OS_SCATTER) {
                        t.aux = aux; t.aux2 = aux2;
                        t.i0 = nets[NET_C0].in_p(); t.i1 = cfg.obs_dim; t.i2 = cfg.act_dim;
                        op.writes.push_back(aux);
                        op.writes.push_back(aux2);
                    }
                }
            }
            push(pr, op);
        }
    }

    // The whole Euler chain in one persistent launch (fql_euler_persistent_kernel): C0 GEMM, then teams of H/32
    // workgroups hand 16-row activation tiles to each other through L2.
    void emit_euler_persistent(Program& pr) {
        const Net& n = nets[NET_BC];
        const int od = cfg.obs_dim, ad = cfg.act_dim, ap = pad16(ad), inp_b = n.in_p();
        const Layer& l0 = n.layers[0];
        {
            Op op{};
            op.type = OP_GEMM;
            GemmTask& t = op.gemm;
            t.A = X_e0; t.lda = l0.in_p; t.B = P + l0.w; t.ldb = l0.out_p; t.bias = P + l0.b; t.C = C0; t.ldc = l0.out_p;
            t.M = B; t.N = l0.out_p; t.K = l0.in_p; t.flags = GF_BIAS;
            op.reads = {X_e0, t.B};
            op.writes = {C0};
            push(pr, op);
        }
        Op op{};
        op.type = OP_PEC;
        PecArgs& a = op.pec;
        a.C0 = C0; a.a0 = X_eu + od; a.lda0 = inp_b;
        a.W0act = P + l0.w + (size_t)od * l0.out_p;
        for (int l = 0; l < 3; ++l) { a.W[l] = P + n.layers[l + 1].w; a.b[l] = P + n.layers[l + 1].b; }
        a.W4 = P + n.layers[4].w; a.b4 = P + n.layers[4].b;
        a.G[0] = pec_g[0]; a.G[1] = pec_g[1]; a.Vg = pec_vg; a.tgt = tgt;
        a.epoch = pec_epoch; a.err = pec_epoch + pec_teams;
        a.M = B; a.ad = ad; a.ap = ap; a.flow_steps = cfg.flow_steps;
        a.nteams = pec_teams; a.ntile = B / 16 / pec_teams;
        op.reads = {C0, X_eu};
        op.writes = {tgt, pec_g[0], pec_g[1], pec_vg};
        push(pr, op);
    }

    // Euler chain with 3 launches per step instead of 5 (agents/fql.py:155-171): [fold head partials -> a_s; layer 0 as a
    // rank-(act+1) update of the loop-invariant C0 = obs W0 + b0; layer 1] , [middle layers] , [last hidden layer +
    // head partials].  The head's reduction over hidden columns crosses workgroups, so each column tile publishes its
    // 16 x ap partial and the consumer folds them in fixed order (deterministic).
    void emit_euler_fused(Program& pr) {
        const Net& n = nets[NET_BC];
        const int nh = n.nl() - 1, fs = cfg.flow_steps;
        const int od = cfg.obs_dim, ad = cfg.act_dim, ap = pad16(ad), inp_b = n.in_p();
        const Layer& l0 = n.layers[0];
        {
            Op op{};
            op.type = OP_GEMM;
            GemmTask& t = op.gemm;
            t.A = X_e0; t.lda = l0.in_p; t.B = P + l0.w; t.ldb = l0.out_p; t.bias = P + l0.b; t.C = C0; t.ldc = l0.out_p;
            t.M = B; t.N = l0.out_p; t.K = l0.in_p; t.flags = GF_BIAS;
            op.reads = {X_e0, t.B};
            op.writes = {C0};
            push(pr, op);
        }
        for (int s = 0; s < fs; ++s) {
            for (int l = 1; l < nh; ++l) {
                const Layer& ly = n.layers[l];
                Op op{};
                op.type = OP_GEMM;
                GemmTask& t = op.gemm;
                t.B = P + ly.w; t.ldb = ly.out_p; t.bias = P + ly.b;
                t.M = B; t.N = ly.out_p; t.K = ly.in_p; t.ldc = ly.out_p;
                t.flags = GF_BIAS | GF_GELU;
                t.i1 = ap; t.i2 = ad; t.f0 = 1.0f / (float)fs; t.f1 = (float)s / (float)fs;
                if (l == 1) {
                    t.flags |= GF_A_EULER0;
                    t.A = C0; t.lda = l0.out_p;
                    t.ew = P + l0.w + (size_t)od * l0.out_p;
                    if (s <= 1) { t.ea_in = X_eu + od; t.i0 = inp_b; } else { t.ea_in = Abuf[(s - 1) & 1]; t.i0 = ap; }
                    op.reads = {C0, t.B, (s <= 1) ? (const void*)X_eu : (const void*)Abuf[(s - 1) & 1]};
                    if (s >= 1) {
                        t.evp = Vpart; t.eb = P + n.layers[nh].b; t.e_ntp = vp_tiles; t.ea_out = Abuf[s & 1];
                        op.reads.push_back(Vpart);
                        op.reads.push_back(P + n.layers[nh].w);
                        op.writes.push_back(Abuf[s & 1]);
                    }
                } else {
                    t.A = p_eu.g[l - 1]; t.lda = ly.in_p;
                    op.reads = {t.A, t.B};
                }
                if (l == nh - 1) {  // nh >= 3: never the same task as part 1
                    t.flags |= GF_HEAD_PART;
                    t.ew4 = P + n.layers[nh].w;
                    t.evp = Vpart;
                    t.C = nullptr;
                    op.writes.push_back(Vpart);
                } else {
                    t.C = p_eu.g[l];
                    op.writes.push_back(t.C);
                }
                push(pr, op);
            }
        }
        Op op{};
        op.type = OP_EULER_FIN;
        const bool from_x = fs == 1;
        op.ef = EulerFinishArgs{from_x ? X_eu + od : Abuf[(fs - 1) & 1], Vpart, P + n.layers[nh].b, tgt, B, ad, ap, vp_tiles,
                                from_x ? inp_b : ap, 1.0f / (float)fs};
        op.reads = {from_x ? (const void*)X_eu : (const void*)Abuf[(fs - 1) & 1], Vpart, P + n.layers[nh].w};   // (head bias: WAR against Adam)
        op.writes = {tgt};
        push(pr, op);
    }

    // The same 3-launches-per-step chain on fql_chain_kernel (512-thread workgroups, fragment-major weights, kernarg tasks)
    bool euler_finish_fused = false;   // set per program build: the head dgrad's prologue finishes the target (GF_A_EULFIN), no OP_EULER_FIN launch
    void emit_euler_chain(Program& pr) {
        const Net& n = nets[NET_BC];
        const int nh = n.nl() - 1, fs = cfg.flow_steps;
        const int od = cfg.obs_dim, ad = cfg.act_dim, ap = pad16(ad), inp_b = n.in_p();
        const Layer& l0 = n.layers[0];
        {
            Op op{};
            op.type = OP_GEMM;
            GemmTask& t = op.gemm;
            t.A = X_e0; t.lda = l0.in_p; t.B = P + l0.w; t.ldb = l0.out_p; t.bias = P + l0.b; t.C = C0; t.ldc = l0.out_p;
            t.M = B; t.N = l0.out_p; t.K = l0.in_p; t.flags = GF_BIAS | (cfg.precision == 2 ? GF_C_FRAGT : GF_C_FRAG);   // the split chain runs layer 0 transposed: C0 in the transposed-tile layout
            op.reads = {X_e0, t.B};
            op.writes = {C0};
            push(pr, op);
        }
        for (int s = 0; s < fs; ++s) {
            for (int l = 1; l < nh; ++l) {
                const Layer& ly = n.layers[l];
                Op op{};
                op.type = OP_CHAIN;
                ChainArgs& a = op.chain;
                a.Wf = wf_bc[l - 1]; a.bias = P + ly.b;
                a.M = B; a.ad = ad; a.ap = ap;
                a.inv_steps = 1.0f / (float)fs; a.t_s = (float)s / (float)fs;
                a.variant = 1;
                constexpr int chain_prio = 0;
                a.prio = chain_prio;
                op.reads = {a.Wf, P + ly.w};
                if (l == 1) {
                    a.variant = 0;
                    a.A = C0; a.W0f = wf_w0;
                    if (s <= 1) { a.ea_in = X_eu + od; a.ea_ld = inp_b; } else { a.ea_in = Abuf[(s - 1) & 1]; a.ea_ld = ap; }
                    op.reads.push_back(C0);
                    op.reads.push_back(wf_w0);
                    op.reads.push_back((s <= 1) ? (const void*)X_eu : (const void*)Abuf[(s - 1) & 1]);
                    if (s >= 1) {
                        a.evp_in = Vpart; a.eb = P + n.layers[nh].b; a.ea_out = Abuf[s & 1];
                        op.reads.push_back(Vpart);
                        op.reads.push_back(P + n.layers[nh].w);
                        op.writes.push_back(Abuf[s & 1]);
                    }
                } else {
                    a.A = p_eu.g[l - 1];
                    op.reads.push_back(a.A);
                }
                if (l == nh - 1) {   // nh >= 3: never the same launch as variant A
                    a.variant = 2;
                    a.W4f = wf_w4; a.evp_out = Vpart;
                    op.reads.push_back(wf_w4);
                    op.writes.push_back(Vpart);
                } else {
                    a.C = p_eu.g[l];
                    op.writes.push_back(a.C);
                }
                push(pr, op);
            }
        }
        if (euler_finish_fused) return;
        Op op{};
        op.type = OP_EULER_FIN;
        const bool from_x = fs == 1;
        op.ef = EulerFinishArgs{from_x ? X_eu + od : Abuf[(fs - 1) & 1], Vpart, P + n.layers[nh].b, tgt, B, ad, ap, vp_tiles,
                                from_x ? inp_b : ap, 1.0f / (float)fs};
        op.reads = {from_x ? (const void*)X_eu : (const void*)Abuf[(fs - 1) & 1], Vpart, P + n.layers[nh].w};   // (head bias: WAR against Adam)
        op.writes = {tgt};
        push(pr, op);
    }
    // The whole chain as ONE persistent XCD-resident launch (fql_xchain.h): 8 row blocks x 32 column slices, activations exchanged through
    // each XCD's own L2, hidden kernels resident in LDS.
    bool xchain_eligible() const {
        // opt-in (FQL_XCHAIN=1).  Measured at B = 256, H = 512 (profiles/r03_xcd_resident.txt): the update takes 352-359 us with this ONE launch in place of
        // the chain's 30 against 344-345 us with the launches (each member re-reads its XCD's whole activation panel and its slice of the layer's kernel
        // every phase: 3 MB through an XCD's L2 per phase).
        if (!getenv("FQL_XCHAIN") || atoi(getenv("FQL_XCHAIN")) == 0) return false;
        if (visual || cfg.precision != 0 || cfg.actor_layer_norm || cfg.act_dim > 15) return false;
        if (num_cus != XCH_NGRP * XCH_NMEM || B % 128 != 0 || B / 128 > XCH_MAXRT) return false;
        const Net& n = nets[NET_BC];
        const int nh = n.nl() - 1;
        if (nh < 2 || nh - 1 > 7) return false;
        const int H = n.layers[0].out;
        if (H % 16 || H > 512 || n.layers[0].in_p > 128 || n.layers[nh].out_p != 16) return false;
        for (int l = 0; l < nh; ++l) if (n.layers[l].out != H || n.layers[l].out_p != H) return false;
        return use_chain;   // (the kernel reads the fragment-major copies the chain launches read)
    }
    void emit_euler_xcd(Program& pr) {
        const Net& n = nets[NET_BC];
        const int nh = n.nl() - 1;
        Op op{};
        op.type = OP_XCHAIN;
        XChainArgs& a = op.xchain;
        a.B = B; a.R = B / 8; a.RT = B / 128;
        a.H = n.layers[0].out_p; a.nl = nh - 1;
        a.od = cfg.obs_dim; a.ad = cfg.act_dim; a.ap = pad16(cfg.act_dim); a.in_p = n.in_p(); a.fs = cfg.flow_steps;
        a.x_e0 = X_eu; a.x_eu = X_eu;   // (C0 masks the action / t rows of the first kernel: the Euler input serves as its observation-only input too)
        a.w0 = P + n.layers[0].w; a.b0 = P + n.layers[0].b;
        for (int l = 1; l < nh; ++l) { a.b[l - 1] = P + n.layers[l].b; a.wf[l - 1] = wf_bc[l - 1]; }
        a.w4 = P + n.layers[nh].w; a.b4 = P + n.layers[nh].b;
        a.hc[0] = p_eu.g[0]; a.hc[1] = p_eu.g[1];
        a.vp = xvp; a.tgt = tgt; a.sync = xsync;
        a.stamps = x_stamps;
        op.reads = {X_eu};
        for (const Layer& L : n.layers) op.reads.push_back(P + L.w);
        for (float* w : wf_bc) op.reads.push_back(w);
        op.writes = {tgt, xvp, p_eu.g[0], p_eu.g[1]};
        push(pr, op);
    }
    // refresh of the fragment-major copies inside a program, behind the Adam launch that rewrites the BC flow's kernels
    void emit_wfrag(Program& pr) {
        if (!use_chain) return;
        const Net& n = nets[NET_BC];
        Op op{};
        op.type = OP_WFRAG;
        for (const Layer& L : n.layers) op.reads.push_back(P + L.w);
        for (float* w : wf_bc) op.writes.push_back(w);
        op.writes.push_back(wf_w0);
        op.writes.push_back(wf_w4);
        push(pr, op);
    }

    // backward of one pass: dz[L] must already hold dLoss/dOut.  rows: view [row_off, row_off+M) of a
    // taller forward pass (one-step actor: only the (obs, z) block is differentiated).
    void emit_backward(Program& pr, const PassBuf& p, int row_off, int M, bool param_grads, bool input_grad,
                       const void* align_with = nullptr) {
        size_t first_op = pr.ops.size();
        const Net& n = nets[p.net];
        const int L = n.nl() - 1;
        auto rows = [&](float* base, int ld) { return base + (size_t)row_off * ld; };
        for (int l = L; l >= 0; --l) {
            const Layer& ly = n.layers[l];
            float* dz = p.dz[l];
            const float* xin = (l == 0) ? rows(p.x0, ly.in_p) : rows(p.xn[l - 1], ly.in_p);
            const void* xin_id = (l == 0) ? (const void*)p.x0 : (const void*)p.xn[l - 1];
            if (param_grads) {
                Op op{};
                op.type = OP_WGRAD;
                WgradTask& w = op.wgrad;
                w.X = xin; w.ldx = ly.in_p;
                w.dZ = dz; w.ldz = ly.out_p;
                w.dW = G + ly.w; w.ldw = ly.out_p;
                w.db = G + ly.b;
                w.M = M; w.Kin = ly.in_p; w.N = ly.out_p;
                op.reads = {xin_id, dz};
                op.writes = {w.dW, w.db};
                // weight gradients feed nothing but the optimizer: background lane
                // weight gradients feed nothing but the optimizer: those of a chain that runs on the critical lane are
                // issued on the side lane so they never sit between two links of that chain
                const int keep = emit_lane;
                constexpr bool wlane = false;  // (a separate wgrad lane is worse than sharing launches)
                if (wlane) emit_lane = 2;
                else if (emit_lane == 0 && !split_build) emit_lane = 1;
                if (defer_wgrads) { op.lane = emit_lane; defer_wgrads->push_back(op); }
                else push(pr, op);
                emit_lane = keep;
            }
            if (l == 0 && !input_grad) break;
            if (l == L && l > 0 && ly.out == 1 && n.layers[l - 1].ln) {
                // scalar head behind a LayerNorm (critic Q): dY = dq (x) w is rank 1, synthesised inside the LN-backward
                const Layer& prev = n.layers[l - 1];
                Op lo{};
                lo.type = OP_LNBWD;
                LnBwdTask& q = lo.ln;
                q.dY = nullptr;
                q.dq = dz; q.ldq = ly.out_p; q.wq = P + ly.w; q.ldw = ly.out_p;
                q.Z = rows(p.z[l - 1], prev.out_p);
                q.Gv = rows(p.g[l - 1], prev.out_p);
                q.stats = p.stats[l - 1] + (size_t)row_off * 2;
                q.gamma = P + prev.g;
                q.dZ = p.dz[l - 1];
                q.dgamma = param_grads ? G + prev.g : nullptr;
                q.dbeta = param_grads ? G + prev.be : nullptr;
                q.M = M; q.H = prev.out_p; q.ld = prev.out_p; q.width = prev.out;
                lo.reads = {dz, p.z[l - 1], p.g[l - 1], p.stats[l - 1], q.gamma};
                lo.writes = {q.dZ};
                if (param_grads) { lo.writes.push_back(q.dgamma); lo.writes.push_back(q.dbeta); }
                push(pr, lo);
                continue;
            }
            // dgrad: dX = dZ W^T
            Op op{};
            op.type = OP_GEMM;
            GemmTask& t = op.gemm;
            t.A = dz; t.lda = ly.out_p;
            t.B = P + ly.w; t.ldb = ly.out_p;
            t.M = M; t.N = ly.in_p; t.K = ly.out_p;
            t.ldc = ly.in_p;
            t.flags = GF_TRANS_B;
            op.reads = {dz, t.B};
            if (l == 0) {
                t.C = p.dx0;
                op.writes = {t.C};
                // state agents read nothing of dX0 but the action block (dQ/da; no encoder behind the observations): the lean 16-column launch
                if (!visual && cfg.precision != 1 && t.K % 512 == 0 && t.M % 16 == 0 && cfg.obs_dim + 16 <= t.N && cfg.act_dim <= 16) {
                    op.type = OP_DGRAD0;
                    t.i1 = cfg.obs_dim;
                }
                push(pr, op);
                break;
            }
            const Layer& prev = n.layers[l - 1];
            if (prev.ln) {
                t.C = p.dy[l - 1];
                op.writes = {t.C};
                if (want64(t.M, t.N, t.K, t.flags)) op.type = OP_GEMM64;
                push(pr, op);
                Op lo{};
                lo.type = OP_LNBWD;
                LnBwdTask& q = lo.ln;
                q.dY = p.dy[l - 1];
                q.Z = rows(p.z[l - 1], prev.out_p);
                q.Gv = rows(p.g[l - 1], prev.out_p);
                q.stats = p.stats[l - 1] + (size_t)row_off * 2;
                q.gamma = P + prev.g;
                q.dZ = p.dz[l - 1];
                q.dgamma = param_grads ? G + prev.g : nullptr;
                q.dbeta = param_grads ? G + prev.be : nullptr;
                q.M = M; q.H = prev.out_p; q.ld = prev.out_p; q.width = prev.out;
                lo.reads = {q.dY, p.z[l - 1], p.g[l - 1], p.stats[l - 1], q.gamma};
                lo.writes = {q.dZ};
                if (param_grads) { lo.writes.push_back(q.dgamma); lo.writes.push_back(q.dbeta); }
                push(pr, lo);
            } else {
                t.C = p.dz[l - 1];
                t.flags |= GF_GELUGRAD;
                t.Zprev = rows(p.z[l - 1], prev.out_p);
                op.reads.push_back(p.z[l - 1]);
                op.writes = {t.C};
                if (want64(t.M, t.N, t.K, t.flags)) op.type = OP_GEMM64;
                // a square hidden layer on the latency lane (the one-step actor's backward tail behind the Euler chain): the chain
                // kernel's variant D does it in about half the time of the generic 16-row kernel
                constexpr bool chain_dgrad = true;
                if (chain_dgrad && op.type == OP_GEMM && use_chain && t.N == t.K && t.N == cfg.actor_hidden[0] && t.M % 16 == 0 &&
                    t.lda == t.K && t.ldb == t.K && t.ldc == t.N && prev.out_p == t.N) {
                    Op co{};
                    co.type = OP_CHAIN;
                    ChainArgs& a = co.chain;
                    a.A = t.A; a.Wf = t.B; a.C = t.C; a.Zprev = t.Zprev; a.bias = nullptr;
                    a.M = t.M; a.ad = cfg.act_dim; a.ap = pad16(cfg.act_dim);
                    a.variant = 3;
                    co.reads = op.reads; co.writes = op.writes;
                    push(pr, co);
                } else
                push(pr, op);
            }
        }
        // level alignment: start this chain together with a sibling chain so their per-level tasks share launches
        if (align_with)
            for (size_t i = first_op; i < pr.ops.size(); ++i)
                if (pr.ops[i].type != OP_WGRAD) { pr.ops[i].reads.push_back(align_with); break; }
    }

    // ---------------------------------------------------------------------------------------
    // scheduling + launch tables + graph capture
    // ---------------------------------------------------------------------------------------
    static bool is_table(OpType t) { return t == OP_GEMM || t == OP_GEMM64 || t == OP_WGRAD || t == OP_LNBWD; }

    void schedule(Program& pr, std::vector<void*>& owner) {
        // list scheduling: level = 1 + max level of every op this one conflicts with (RAW, WAW, WAR)
        std::map<const void*, int> last_writer;               // buffer -> op index of last writer
        std::map<const void*, std::vector<int>> readers;      // buffer -> ops reading it since that write
        for (int oi = 0; oi < (int)pr.ops.size(); ++oi) {
            Op& op = pr.ops[oi];
            op.deps.clear();
            for (const void* r : op.reads) {
                auto it = last_writer.find(r);
                if (it != last_writer.end()) op.deps.push_back(it->second);
            }
            for (const void* w : op.writes) {
                auto it = last_writer.find(w);
                if (it != last_writer.end()) op.deps.push_back(it->second);
                auto ir = readers.find(w);
                if (ir != readers.end()) for (int r : ir->second) if (r != oi) op.deps.push_back(r);
            }
            int lv = 0;
            for (int d : op.deps) lv = std::max(lv, pr.ops[d].level + 1);
            op.level = lv;
            for (const void* r : op.reads) readers[r].push_back(oi);
            for (const void* w : op.writes) {
                last_writer[w] = oi;
                readers.erase(w);
            }
        }
        {
            // The folds of the encoders' weight-gradient partials (27 tiny tasks, one launch per convolution level = 9 launches of ~16 us in the
            // middle of the backward pass) are needed by nothing but their module's Adam launch: as late as possible they share one launch per module.
            std::vector<std::vector<int>> cons(pr.ops.size());
            for (int oi = 0; oi < (int)pr.ops.size(); ++oi)
                for (int d : pr.ops[oi].deps) cons[d].push_back(oi);
            for (int oi = (int)pr.ops.size() - 1; oi >= 0; --oi) {
                Op& op = pr.ops[oi];
                if (op.type != OP_CONV_WRED) continue;
                int lv = 1 << 30;
                for (int c : cons[oi]) lv = std::min(lv, pr.ops[c].level - 1);
                if (cons[oi].empty()) {   // (the optimizer is another program: the end of this one)
                    lv = 0;
                    for (const Op& o2 : pr.ops) lv = std::max(lv, o2.level);
                }
                if (lv > op.level) op.level = lv;
            }
        }
        int maxlv = 0;
        for (const Op& op : pr.ops) maxlv = std::max(maxlv, op.level);
        pr.launches.clear();
        pr.rec_n = 0;
        pr.two_lanes = false;
        for (bool& b : pr.lane_used) b = false;
        std::vector<int> launch_of(pr.ops.size(), -1);
        for (int lv = 0; lv <= maxlv; ++lv) {
          for (int lane = 0; lane < FQL_LANES; ++lane) {
            for (int ty = 0; ty <= OP_FINALIZE; ++ty) {
                std::vector<const Op*> sel;
                for (int oi = 0; oi < (int)pr.ops.size(); ++oi) {
                    const Op& op = pr.ops[oi];
                    if (op.level == lv && op.type == ty && op.lane == lane) sel.push_back(&op);
                }
                constexpr bool merge_side = true;
                if (merge_side && (ty == OP_WGRAD || ty == OP_LNBWD || ty == OP_POSTOS || ty == OP_LOSS_CRITIC || ty == OP_LOSS_Q ||
                                   ty == OP_LOSS_BC)) continue;  // folded into the OP_GEMM64 iteration
                if (merge_side && ty == OP_GEMM64) {
                    std::vector<const Op*> selw, sell, selm;
                    for (const Op& op : pr.ops) {
                        if (op.level != lv || op.lane != lane) continue;
                        if (op.type == OP_WGRAD) selw.push_back(&op);
                        if (op.type == OP_LNBWD) sell.push_back(&op);
                        if (op.type == OP_POSTOS || op.type == OP_LOSS_CRITIC || op.type == OP_LOSS_Q || op.type == OP_LOSS_BC) selm.push_back(&op);
                    }
                    if (sel.empty() && selw.empty() && sell.empty() && selm.empty()) continue;
                    if (lane >= 1) pr.two_lanes = true;
                    pr.lane_used[lane] = true;
                    Launch L;
                    L.type = OP_GEMM64;
                    L.side = true;
                    L.lane = lane;
                    const int li = (int)pr.launches.size();
                    for (const Op* o : sel) launch_of[o - pr.ops.data()] = li;
                    for (const Op* o : selw) launch_of[o - pr.ops.data()] = li;
                    for (const Op* o : sell) launch_of[o - pr.ops.data()] = li;
                    for (const Op* o : selm) launch_of[o - pr.ops.data()] = li;
                    // 32 x 64 tiles while a level is a latency chain (B = 256: 1.5 workgroups per CU), 64 x 64 once a task alone
                    // brings >= 128 of them (M >= 1024: +2 % at B = 1024)
                    constexpr int ri_env = 0;
                    constexpr bool xcd_order = true;
                    int ri = 1;
                    int tile = 0;
                    std::vector<GemmTask> tg;
                    // tile shape of the 32-row tasks of this launch: 32 x 64 unless 32 x 32 tiles deal out more evenly over the CUs
                    // (e.g. 384 tiles of 32 x 64 = two tile times on half the CUs; 768 of 32 x 32 = three half-size tiles everywhere)
                    constexpr int nj_env = 0;
                    int nj = 2;
                    {
                        int t64 = 0;
                        double other = 0.0;   // in units of a 32 x 32 x 512 tile
                        for (const Op* o : sel) {
                            const GemmTask& t = o->gemm;
                            const bool big = cfg.precision == 2 ? false : ri_env ? ri_env == 2 : (t.M >= 1024 && t.M % 64 == 0);
                            if (big) other += 4.0 * (t.M / 64) * (t.N / 64) * t.K / 512.0;
                            else t64 += (t.M / 32) * (t.N / 64);
                        }
                        for (const Op* o : selw) other += 0.5 * (o->wgrad.Kin / 16) * ((o->wgrad.N + 63) / 64) * o->wgrad.M / 256.0;
                        for (const Op* o : sell) other += 0.05 * o->ln.M;
                        const int ncu = std::max(1, num_cus);
                        const double m64 = 2.0 * ((t64 + ncu - 1) / ncu) + other / ncu, m32 = 1.0 * ((2 * t64 + ncu - 1) / ncu) + other / ncu;
                        if (t64 > 0 && m32 < m64 && !wide_tiles) nj = 1;
                        if (nj_env == 1 || nj_env == 2) nj = nj_env;
                    }
                    for (const Op* o : sel) {
                        GemmTask t = o->gemm;
                        const int ri_t = cfg.precision == 2 ? 1 : ri_env ? ri_env : (t.M >= 1024 && t.M % 64 == 0 ? 2 : 1);   // split bodies are 32-row tiles
                        ri = std::max(ri, ri_t);
                        if (ri_t == 2) L.tmt2 = true;
                        t.tmt = ri_t;
                        t.wk = ri_t == 2 ? 2 : nj;          // MFMA column tiles per wave: tile = 32 x (32 wk)
                        t.ntn = t.N / (32 * t.wk); t.tile0 = tile;
                        t.xg = 0;
                        if (ri_t == 1 && xcd_order && tile % 8 == 0) {   // XCD-aware tile order: gm x gn = 8 blocks of the tile grid, fewest panel fetches
                            const int ntm = t.M / 32;
                            // gm row groups x gn column groups: an A row panel is fetched by gn XCDs, a B column panel by gm -> fewest panel bytes
                            // (FQL_XCD_GM forces gm; measured on the whole update: the rule's choice 3075, forced 4: 3001-3061, 8: 2864-3026, 1: 2967-3040, off: 3025)
                            constexpr int gm_env = 0;
                            double best = 1e30;
                            for (int gm : {1, 2, 4, 8}) {
                                const int gn = 8 / gm;
                                if (ntm % gm || t.ntn % gn || (gm_env && gm != gm_env)) continue;
                                const double cost = (double)gn * t.M + (double)gm * t.N;
                                if (cost < best) { best = cost; t.xg = gm; }
                            }
                        }
                        tile += (t.M / (32 * ri_t)) * t.ntn;
                        tg.push_back(t);
                    }
                    L.tile_w = tile;
                    std::vector<WgradTask> tw;
                    const int wkt = cfg.precision == 2 ? FQL_WGRAD_KT_SPLIT : FQL_WGRAD_KT;   // weight-gradient tile height of the body this launch runs
                    int tilew = 0;
                    for (const Op* o : selw) {
                        WgradTask t = o->wgrad;
                        t.ntn = (t.N + 63) / 64; t.tile0 = tilew;
                        tilew += ((t.Kin + wkt - 1) / wkt) * t.ntn;
                        tw.push_back(t);
                    }
                    L.tile_l = L.tile_w + tilew;
                    std::vector<LnBwdTask> tl;
                    int tilel = 0;
                    for (const Op* o : sell) {
                        LnBwdTask t = o->ln;
                        t.ntiles_rows = (t.M + 3) / 4; t.tile0 = tilel;
                        tilel += t.ntiles_rows + (t.dgamma ? t.H / 16 : 0);
                        tl.push_back(t);
                    }
                    L.tile_m = L.tile_l + tilel;
                    std::vector<MiscTask> tmisc;
                    for (const Op* o : selm) {
                        MiscTask t{};
                        if (o->type == OP_POSTOS) { t.kind = MISC_POSTOS; t.po = o->postos; }
                        else if (o->type == OP_LOSS_CRITIC) { t.kind = MISC_LOSS_CRITIC; t.lc = o->lc; }
                        else if (o->type == OP_LOSS_Q) { t.kind = MISC_LOSS_Q; t.lq = o->lq; }
                        else { t.kind = MISC_LOSS_BC; t.lb = o->lb; }
                        tmisc.push_back(t);
                    }
                    L.grid = L.tile_m + (int)tmisc.size();
                    L.ntasks = (int)tg.size(); L.n_w = (int)tw.size(); L.n_l = (int)tl.size();
                    const size_t wlds = cfg.precision == 2 ? (size_t)(4 * 4 * 64 * 4 + 4 * 64) : (size_t)FQL_WGRAD_LDS_FLOATS;   // 16- / 32-input tiles
                    L.lds = sizeof(float) * (tg.empty() ? wlds
                                             : ri == 2 ? (size_t)(2 * (32 * ri + 64) * 68 + 256)
                                             : cfg.precision == 2 ? (size_t)FQL_TILE_SPLIT_LDS_FLOATS(nj)
                                             : nj == 1 ? (size_t)(2 * 32 * 68 + 2 * 64 * 36 + 128) : (size_t)(2 * 32 * 68 + 2 * 64 * 68 + 128));
                    L.lds = std::max(L.lds, sizeof(float) * (selw.empty() ? (size_t)0 : wlds));
                    auto up = [&](const void* src, size_t bytes) -> void* {
                        void* d = dalloc(owner, bytes / sizeof(float) + 4);
                        if (bytes) HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
                        return d;
                    };
                    L.table = up(tg.data(), tg.size() * sizeof(GemmTask));
                    L.table_w = up(tw.data(), tw.size() * sizeof(WgradTask));
                    L.table_l = up(tl.data(), tl.size() * sizeof(LnBwdTask));
                    L.table_m = up(tmisc.data(), tmisc.size() * sizeof(MiscTask));
                    pr.launches.push_back(L);
                    continue;
                }
                if (sel.empty()) continue;
                if (lane >= 1) pr.two_lanes = true;
                pr.lane_used[lane] = true;
                if (ty == OP_CONV || ty == OP_CONV_U8 || ty == OP_CONV_WGRAD || ty == OP_CONV_WRED) {
                    // every convolution-family op of this level and lane shares ONE launch (e.g. the same layer of the four
                    // encoder passes): arg structs in an HBM table, workgroup ranges by tile0
                    Launch L;
                    L.type = (OpType)ty;
                    L.lane = lane;
                    for (const Op* o : sel) launch_of[o - pr.ops.data()] = (int)pr.launches.size();
                    L.ntasks = (int)sel.size();
                    int tile = 0;
                    auto up = [&](const void* src, size_t bytes) {
                        void* d = dalloc(owner, bytes / sizeof(float) + 4);
                        HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
                        return d;
                    };
                    if (ty == OP_CONV || ty == OP_CONV_U8) {
                        std::vector<ConvArgs> tb;
                        long long nb_all = 0;
                        for (const Op* o : sel) nb_all += (long long)o->conv.N * (o->conv.H / o->conv.R);
                        // persistent uint8 kernel: its 162 registers (launch bound 3 waves per SIMD) allow three workgroups per CU; exactly that many are launched (a fourth
                        // round of workgroups would run alone) and dealt to the tasks in proportion to their row blocks (the one-step pass
                        // holds [obs ; next_obs], twice the images of the others), rounded DOWN so the total never exceeds the resident set
                        constexpr int u8_per_cu = 3;
                        for (const Op* o : sel) {
                            ConvArgs a = o->conv;
                            const int nb = a.N * (a.H / a.R);
                            a.tile0 = tile;
                            // float layers: a workgroup per row block by default; FQL_CONV_WGS_PER_CU=k makes them persistent too (measured: +-1 %)
                            constexpr int conv_per_cu = 0;
                            const int per_cu = ty == OP_CONV_U8 ? u8_per_cu : conv_per_cu;
                            a.nwg = per_cu > 0 ? std::min(nb, std::max(1, (int)(((long long)per_cu * num_cus * nb) / nb_all))) : nb;
                            if (a.parg) {   // convolution + max-pool in one kernel: a workgroup per (image, pooled row pair)
                                a.nwg = a.N * ((a.H + 3) / 4);
                                L.pool_fused = true;
                                L.lds = std::max(L.lds, (size_t)FQL_CONV_POOL_LDS_FLOATS(a.W, a.Ci, a.Co) * sizeof(float));
                            } else if (L.pool_fused) invalid("internal: a launch mixes fused and plain convolutions");
                            tile += a.nwg;
                            L.lds = std::max(L.lds, ((size_t)(a.R + 2) * (a.W + 2) * (a.Ci + 4) + (size_t)a.Co * (9 * a.Ci + 4)) * sizeof(float));
                            if (ty == OP_CONV && cfg.precision == 2) L.lds = std::max(L.lds, (size_t)FQL_CONV_SPLIT_LDS_WORDS(a.R, a.W, a.Ci, a.Co) * sizeof(float));
                            tb.push_back(a);
                        }
                        L.table = up(tb.data(), tb.size() * sizeof(ConvArgs));
                    } else if (ty == OP_CONV_WGRAD) {
                        std::vector<ConvWgradArgs> tb;
                        for (const Op* o : sel) {
                            ConvWgradArgs a = o->cw;
                            a.tile0 = tile; a.nwg = o->cw_grid;
                            tile += a.nwg;
                            L.lds = std::max(L.lds, ((size_t)(a.R + 2) * (a.W + 2) * (a.Ci + 4) + (size_t)a.R * a.W * (a.Co + 4)) * sizeof(float));
                            tb.push_back(a);
                        }
                        L.table = up(tb.data(), tb.size() * sizeof(ConvWgradArgs));
                    } else {
                        std::vector<ConvWredArgs> tb;
                        for (const Op* o : sel) {
                            ConvWredArgs a = o->cwr;
                            a.tile0 = tile;
                            tile += ((9 * a.Ci + 1) * a.Co + 63) / 64;
                            tb.push_back(a);
                        }
                        L.table = up(tb.data(), tb.size() * sizeof(ConvWredArgs));
                    }
                    L.grid = tile;
                    pr.launches.push_back(L);
                    continue;
                }
                if (ty == OP_DGRAD0 && !sel.empty()) {   // the members' tasks of a level in launches of up to four (the kernel takes them as its argument)
                    for (size_t i0 = 0; i0 < sel.size(); i0 += 4) {
                        Launch L;
                        L.type = OP_DGRAD0;
                        L.op = *sel[i0];
                        L.lane = lane;
                        for (size_t i = i0; i < std::min(sel.size(), i0 + 4); ++i) {
                            if (sel[i]->gemm.M != sel[i0]->gemm.M || sel[i]->gemm.i1 != sel[i0]->gemm.i1) invalid("internal: dQ/da tasks of one level differ in shape");
                            L.dg_tasks.push_back(sel[i]->gemm);
                            launch_of[sel[i] - pr.ops.data()] = (int)pr.launches.size();
                        }
                        pr.launches.push_back(L);
                    }
                    continue;
                }
                if (ty == OP_CONV_WPREP && sel.size() > 1) {   // the encoder passes of one level refresh their LDS-layout weight copies in ONE launch
                    int tot = 0;
                    for (const Op* o : sel) tot += o->wprep_n;
                    ConvWprepTask* all = (ConvWprepTask*)dalloc(owner, (size_t)tot * sizeof(ConvWprepTask) / sizeof(float) + 4);
                    int at = 0;
                    for (const Op* o : sel) {
                        HIP_CHECK(hipMemcpy(all + at, o->wprep_tasks, (size_t)o->wprep_n * sizeof(ConvWprepTask), hipMemcpyDeviceToDevice));
                        at += o->wprep_n;
                        launch_of[o - pr.ops.data()] = (int)pr.launches.size();
                    }
                    Launch L;
                    L.type = OP_CONV_WPREP;
                    L.op = *sel[0];
                    L.op.wprep_tasks = all;
                    L.op.wprep_n = tot;
                    L.lane = lane;
                    pr.launches.push_back(L);
                    continue;
                }
                if (!is_table((OpType)ty)) {
                    for (const Op* o : sel) {
                        Launch L;
                        L.type = (OpType)ty;
                        L.op = *o;
                        L.lane = lane;
                        launch_of[o - pr.ops.data()] = (int)pr.launches.size();
                        pr.launches.push_back(L);
                    }
                    continue;
                }
                Launch L;
                L.type = (OpType)ty;
                L.lane = lane;
                for (const Op* o : sel) launch_of[o - pr.ops.data()] = (int)pr.launches.size();
                L.ntasks = (int)sel.size();
                int tile = 0;
                if (ty == OP_GEMM) {
                    std::vector<GemmTask> tb;
                    for (const Op* o : sel) {
                        GemmTask t = o->gemm;
                        constexpr int tmt_side = 1;
                        t.wk = (t.N <= 16) ? 4 : 2;
                        // throughput lane: two 16-row tiles per workgroup share each B fragment
                        t.tmt = (lane == 1 && tmt_side == 2 && t.M % 32 == 0 && t.M >= 256 && t.N >= 32) ? 2 : 1;
                        t.ntn = (t.N / 16 + (4 / t.wk) - 1) / (4 / t.wk);
                        t.tile0 = tile;
                        tile += (t.M / (16 * t.tmt)) * t.ntn;
                        if (t.tmt == 2) L.tmt2 = true;
                        if (t.K > 512) L.kbig = true;
                        if (t.flags & (GF_A_EULER0 | GF_HEAD_PART | GF_A_LOSSACT)) L.euler = true;
                        L.lds = std::max(L.lds, ((size_t)16 * t.tmt * (t.K + 4) + 1024 * t.tmt + 1280) * sizeof(float));
                        tb.push_back(t);
                    }
                    L.table = dalloc(owner, tb.size() * sizeof(GemmTask) / sizeof(float) + 4);
                    HIP_CHECK(hipMemcpy(L.table, tb.data(), tb.size() * sizeof(GemmTask), hipMemcpyHostToDevice));
                } else if (ty == OP_GEMM64) {
                    std::vector<GemmTask> tb;
                    for (const Op* o : sel) {
                        GemmTask t = o->gemm;
                        constexpr int ri_env = 0;
                        const int ri = ri_env ? ri_env : (t.M >= 1024 && t.M % 64 == 0 ? 2 : 1);
                        t.wk = 2; t.tmt = ri;  // row tiles per wave: workgroup tile (32 ri) x 64
                        t.ntn = t.N / 64;
                        t.tile0 = tile;
                        tile += (t.M / (32 * ri)) * t.ntn;
                        tb.push_back(t);
                    }
                    L.lds = (size_t)(4 * 64 * 68 + 256) * sizeof(float);
                    L.table = dalloc(owner, tb.size() * sizeof(GemmTask) / sizeof(float) + 4);
                    HIP_CHECK(hipMemcpy(L.table, tb.data(), tb.size() * sizeof(GemmTask), hipMemcpyHostToDevice));
                } else if (ty == OP_WGRAD) {
                    std::vector<WgradTask> tb;
                    for (const Op* o : sel) {
                        WgradTask t = o->wgrad;
                        t.ntn = (t.N + 63) / 64;
                        t.tile0 = tile;
                        tile += ((t.Kin + FQL_WGRAD_KT - 1) / FQL_WGRAD_KT) * t.ntn;
                        tb.push_back(t);
                    }
                    L.table = dalloc(owner, tb.size() * sizeof(WgradTask) / sizeof(float) + 4);
                    HIP_CHECK(hipMemcpy(L.table, tb.data(), tb.size() * sizeof(WgradTask), hipMemcpyHostToDevice));
                } else {
                    std::vector<LnBwdTask> tb;
                    for (const Op* o : sel) {
                        LnBwdTask t = o->ln;
                        t.ntiles_rows = (t.M + 3) / 4;
                        t.tile0 = tile;
                        tile += t.ntiles_rows + (t.dgamma ? t.H / 16 : 0);
                        tb.push_back(t);
                    }
                    L.table = dalloc(owner, tb.size() * sizeof(LnBwdTask) / sizeof(float) + 4);
                    HIP_CHECK(hipMemcpy(L.table, tb.data(), tb.size() * sizeof(LnBwdTask), hipMemcpyHostToDevice));
                }
                L.grid = tile;
                pr.launches.push_back(L);
            }
          }
        }
        // cross-lane edges (computed below) are printed by the dump too: move the dump after them
        auto dump_program = [&]() {
            int cnt[FQL_LANES] = {};
            for (const Launch& L : pr.launches) {
                cnt[L.lane]++;
                int lv = -1;
                for (int oi = 0; oi < (int)pr.ops.size(); ++oi) if (launch_of[oi] == (int)(&L - pr.launches.data())) lv = pr.ops[oi].level;
                fprintf(stderr, "[fql] #%2d level %3d lane %d type %2d ntasks %2d grid %5d waits[", (int)(&L - pr.launches.data()), lv, L.lane, (int)L.type, L.ntasks, L.grid);
                for (int w : L.waits) fprintf(stderr, "%d ", w);
                fprintf(stderr, "] :");
                for (int oi = 0; oi < (int)pr.ops.size(); ++oi) {
                    if (launch_of[oi] != (int)(&L - pr.launches.data())) continue;
                    const Op& o = pr.ops[oi];
                    if (o.type == OP_GEMM || o.type == OP_GEMM64) fprintf(stderr, " g%s(%dx%dx%d%s)", o.type == OP_GEMM64 ? "64" : "16", o.gemm.M, o.gemm.N, o.gemm.K, (o.gemm.flags & GF_TRANS_B) ? "T" : "");
                    else if (o.type == OP_WGRAD) fprintf(stderr, " w(%dx%dx%d)", o.wgrad.M, o.wgrad.Kin, o.wgrad.N);
                    else if (o.type == OP_LNBWD) fprintf(stderr, " ln(%d)", o.ln.M);
                    else fprintf(stderr, " t%d", (int)o.type);
                }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[fql] launches per lane:");
            for (int l = 0; l < FQL_LANES; ++l) fprintf(stderr, " %d", cnt[l]);
            fprintf(stderr, "\n");
        };
        for (int oi = 0; oi < (int)pr.ops.size(); ++oi) {   // algorithmic MACs per launch (real, unpadded layer widths are within 1 % of these)
            const Op& o = pr.ops[oi];
            const int li = launch_of[oi];
            if (li < 0) continue;
            double m = 0.0;
            if (o.type == OP_GEMM || o.type == OP_GEMM64 || o.type == OP_HEAD_DGRAD) m = (double)o.gemm.M * o.gemm.N * o.gemm.K;
            else if (o.type == OP_DGRAD0) m = (double)o.gemm.M * 16.0 * o.gemm.K;
            else if (o.type == OP_WGRAD) m = (double)o.wgrad.M * o.wgrad.Kin * o.wgrad.N;
            else if (o.type == OP_CHAIN) {
                const double H = cfg.actor_hidden[0];
                m = (double)o.chain.M * H * H + (o.chain.variant == 0 ? (double)o.chain.M * 16.0 * H : 0.0) + (o.chain.variant == 2 ? (double)o.chain.M * H * o.chain.ap : 0.0);
            }
            else if (o.type == OP_XCHAIN) {
                const Net& nb = nets[NET_BC];
                for (const Layer& L : nb.layers) m += (double)B * L.in * L.out;
                m = m * cfg.flow_steps - (double)B * cfg.obs_dim * nb.layers[0].out * (cfg.flow_steps - 1);   // (the obs part of layer 0 is computed once)
            }
            else if (o.type == OP_CONV || o.type == OP_CONV_U8) m = (double)o.conv.N * o.conv.H * o.conv.W * 9.0 * (o.conv.transposed ? o.conv.Ci : o.conv.Ci_real) * o.conv.Co;
            else if (o.type == OP_CONV_WGRAD) m = (double)o.cw.N * o.cw.H * o.cw.W * 9.0 * o.cw.Ci_real * o.cw.Co;
            pr.launches[li].macs += m;
        }
        // cross-lane edges: a launch waits for the latest launch of the other lane it depends on (lane streams
        // are in-order, so that covers the earlier ones); skip waits already implied by an earlier wait.
        int waited_upto[FQL_LANES][FQL_LANES];  // [waiting lane][other lane]: highest launch index already waited for
        for (auto& r : waited_upto) for (int& v : r) v = -1;
        for (int li = 0; li < (int)pr.launches.size(); ++li) {
            Launch& L = pr.launches[li];
            int need[FQL_LANES];
            for (int& v : need) v = -1;
            for (int oi = 0; oi < (int)pr.ops.size(); ++oi) {
                if (launch_of[oi] != li) continue;
                for (int d : pr.ops[oi].deps) {
                    const int dl = launch_of[d];
                    const int ol = pr.launches[dl].lane;
                    if (ol != L.lane) need[ol] = std::max(need[ol], dl);
                }
            }
            for (int ol = 0; ol < FQL_LANES; ++ol)
                if (need[ol] > waited_upto[L.lane][ol]) {
                    L.waits.push_back(need[ol]);
                    pr.launches[need[ol]].record_after = true;
                    waited_upto[L.lane][ol] = need[ol];
                }
        }
        if (getenv("FQL_DUMP")) dump_program();
        // Two FORKED lanes (>= 1; lane 0 is the capture's origin stream) must not wait on each other in both directions: the HIP
        // runtime bundled with torch 2.10 (ROCm 7.0) walks the resulting cycle of parallel capture streams forever in
        // hip::Stream::EndCapture (a segmentation fault by stack exhaustion; diagnosed with rocgdb, profiles/r02_capture_crash.txt).
        // The default programs keep such edges one-directional by construction; placement overrides that do not are refused.
        bool edge[FQL_LANES][FQL_LANES] = {};
        for (const Launch& L : pr.launches)
            for (int w : L.waits) edge[pr.launches[w].lane][L.lane] = true;
        for (int a = 1; a < FQL_LANES; ++a)
            for (int b = a + 1; b < FQL_LANES; ++b)
                if (edge[a][b] && edge[b][a])
                    invalid("lane placement makes lanes %d and %d wait on each other in both directions: not capturable on this HIP runtime "
                            "(hipStreamEndCapture recursion); move one of the passes", a, b);
    }

    // one launch of a program on stream s (tl: timeline id of the diagnostics build, -1 = none); reads engine state only
    // Per-launch profiling (fql_profile_update): when set, the next launch carries these two events ON ITS DISPATCH
    // (hipExtLaunchKernelGGL): their elapsed time is the dispatch's own begin-to-end time, the quantity a rocprofv3 kernel trace reports.
    hipEvent_t prof_a = nullptr, prof_b = nullptr;
    template <typename... Pm, typename... A>
    void aql_record(void (*k)(Pm...), dim3 g, dim3 b, size_t l, A&&... a) {
        AqlDispatch d;
        d.fn = (const void*)k;
        d.grid[0] = g.x; d.grid[1] = g.y; d.grid[2] = g.z;
        d.block[0] = b.x; d.block[1] = b.y; d.block[2] = b.z;
        d.lds = (uint32_t)l;
        size_t off = 0;
        auto put = [&](auto v) {   // kernarg layout: every argument at its natural alignment
            using T = decltype(v);
            off = (off + alignof(T) - 1) & ~(alignof(T) - 1);
            d.args.resize(off + sizeof(T));
            std::memcpy(d.args.data() + off, &v, sizeof(T));
            off += sizeof(T);
        };
        (put(static_cast<Pm>(a)), ...);
        aql_rec->push_back(std::move(d));
    }
#define FQL_LAUNCH(k, g, b, l, st, ...)                                                          \
    do {                                                                                          \
        if (aql_rec) aql_record(k, g, b, (size_t)(l), __VA_ARGS__);                               \
        else if (prof_a) hipExtLaunchKernelGGL(k, g, b, (std::uint32_t)(l), st, prof_a, prof_b, 0u, __VA_ARGS__); \
        else hipLaunchKernelGGL(k, g, b, l, st, __VA_ARGS__);                                     \
    } while (0)
    void issue(const Launch& L, hipStream_t s, int tl) {
        constexpr bool u8_split = true;
        constexpr int side_prio = 0;
        switch (L.type) {
            case OP_GEMM:
                if (L.euler && L.kbig) FQL_LAUNCH((fql_gemm16_euler_kernel<true>), dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks, tl);
                else if (L.euler) FQL_LAUNCH((fql_gemm16_euler_kernel<false>), dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks, tl);
                else if (L.tmt2 && L.kbig) FQL_LAUNCH((fql_gemm16_kernel<true, true>), dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks, tl);
                else if (L.tmt2) FQL_LAUNCH((fql_gemm16_kernel<true, false>), dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks, tl);
                else if (L.kbig) FQL_LAUNCH((fql_gemm16_kernel<false, true>), dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks, tl);
                else FQL_LAUNCH((fql_gemm16_kernel<false, false>), dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks, tl);
                break;
            case OP_GEMM64:
                if (L.side && L.tmt2)
                    FQL_LAUNCH(fql_side_big_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks,
                                       (const WgradTask*)L.table_w, L.n_w, (const LnBwdTask*)L.table_l, L.n_l, L.tile_w, L.tile_l,
                                       (const MiscTask*)L.table_m, L.tile_m, 0, tl);
                else if (L.side && cfg.precision == 2)
                    FQL_LAUNCH(fql_side_split_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks,
                                       (const WgradTask*)L.table_w, L.n_w, (const LnBwdTask*)L.table_l, L.n_l, L.tile_w, L.tile_l,
                                       (const MiscTask*)L.table_m, L.tile_m, side_prio, tl);
                else if (L.side)
                    FQL_LAUNCH(fql_side_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks,
                                       (const WgradTask*)L.table_w, L.n_w, (const LnBwdTask*)L.table_l, L.n_l, L.tile_w, L.tile_l,
                                       (const MiscTask*)L.table_m, L.tile_m, side_prio, tl);
                else
                    FQL_LAUNCH(fql_gemm64_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const GemmTask*)L.table, L.ntasks);
                break;
            case OP_WGRAD:
                FQL_LAUNCH(fql_wgrad_kernel, dim3(L.grid), dim3(FQL_THREADS), 0, s, (const WgradTask*)L.table, L.ntasks);
                break;
            case OP_LNBWD:
                FQL_LAUNCH(fql_lnbwd_kernel, dim3(L.grid), dim3(FQL_THREADS), 0, s, (const LnBwdTask*)L.table, L.ntasks);
                break;
            case OP_PREP:
                { PrepArgs pa = L.op.prep; pa.tl = tl; FQL_LAUNCH(fql_prep_kernel, dim3((pa.B + 3) / 4), dim3(FQL_THREADS), 0, s, pa); }
                break;
            case OP_POSTOS:
                FQL_LAUNCH(fql_post_onestep_kernel, dim3(1), dim3(FQL_THREADS), 0, s, L.op.postos);
                break;
            case OP_PEC: {
                const PecArgs& a = L.op.pec;
                const int T = cfg.actor_hidden[0] / 32;
                const size_t lds = ((size_t)16 * (cfg.actor_hidden[0] + 4) + 1024 + 576 + 512 * (size_t)a.ntile) * sizeof(float);
                if (cfg.actor_hidden[0] == 512) FQL_LAUNCH((fql_euler_persistent_kernel<512>), dim3(a.nteams * T), dim3(FQL_THREADS), lds, s, a);
                else FQL_LAUNCH((fql_euler_persistent_kernel<256>), dim3(a.nteams * T), dim3(FQL_THREADS), lds, s, a);
                break;
            }
            case OP_EULER_FIN:
                FQL_LAUNCH(fql_euler_finish_kernel, dim3((L.op.ef.M * L.op.ef.ad + FQL_THREADS - 1) / FQL_THREADS), dim3(FQL_THREADS), 0, s, L.op.ef);
                break;
            case OP_LOSS_CRITIC:
                FQL_LAUNCH(fql_loss_critic_kernel, dim3(1), dim3(FQL_THREADS), 0, s, L.op.lc);
                break;
            case OP_LOSS_Q:
                FQL_LAUNCH(fql_loss_q_kernel, dim3(1), dim3(FQL_THREADS), 0, s, L.op.lq);
                break;
            case OP_LOSS_BC:
                FQL_LAUNCH(fql_loss_bc_kernel, dim3(1), dim3(FQL_THREADS), 0, s, L.op.lb);
                break;
            case OP_LOSS_ACTOR:
                FQL_LAUNCH(fql_loss_actor_kernel, dim3(1), dim3(FQL_THREADS), 0, s, L.op.la);
                break;
            case OP_CONV_WPREP:
                FQL_LAUNCH(fql_conv_wprep_kernel, dim3(4, L.op.wprep_n), dim3(FQL_THREADS), 0, s, L.op.wprep_tasks);
                break;
            case OP_CONV:
                if (L.pool_fused && cfg.precision == 2) FQL_LAUNCH(fql_conv3x3_pool_split_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                else if (L.pool_fused) FQL_LAUNCH(fql_conv3x3_pool_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                else if (cfg.precision == 2) FQL_LAUNCH(fql_conv3x3_split_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                else
                FQL_LAUNCH(fql_conv3x3_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                break;
            case OP_CONV_U8:
                if (L.pool_fused && cfg.precision == 2) FQL_LAUNCH(fql_conv3x3_pool_split_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                else if (L.pool_fused) FQL_LAUNCH(fql_conv3x3_pool_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                else if (cfg.precision == 2 && u8_split) FQL_LAUNCH(fql_conv3x3_u8_split_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                else
                FQL_LAUNCH(fql_conv3x3_u8_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvArgs*)L.table, L.ntasks);
                break;
            case OP_POOL: {
                const PoolArgs& a = L.op.pool;
                const size_t tot = (size_t)a.N * (a.H / 2) * (a.W / 2) * (a.C / 4);
                FQL_LAUNCH(fql_maxpool_kernel, dim3((unsigned)((tot + FQL_THREADS - 1) / FQL_THREADS)), dim3(FQL_THREADS), 0, s, a);
                break;
            }
            case OP_POOL_BWD: {
                const PoolBwdArgs& a = L.op.poolb;
                const size_t tot = (size_t)a.N * a.H * a.W * (a.C / 4);
                FQL_LAUNCH(fql_maxpool_bwd_kernel, dim3((unsigned)((tot + FQL_THREADS - 1) / FQL_THREADS)), dim3(FQL_THREADS), 0, s, a);
                break;
            }
            case OP_CONV_WGRAD:
                if (cfg.precision == 2) FQL_LAUNCH(fql_conv_wgrad_split_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvWgradArgs*)L.table, L.ntasks);
                else
                FQL_LAUNCH(fql_conv_wgrad_kernel, dim3(L.grid), dim3(FQL_THREADS), L.lds, s, (const ConvWgradArgs*)L.table, L.ntasks);
                break;
            case OP_CONV_WRED:
                FQL_LAUNCH(fql_conv_wgrad_reduce_kernel, dim3(L.grid), dim3(FQL_THREADS), 0, s, (const ConvWredArgs*)L.table, L.ntasks);
                break;
            case OP_ENC_DZ: {
                const EncDzArgs& a = L.op.edz;
                FQL_LAUNCH(fql_enc_dz_kernel, dim3((a.M * a.n + FQL_THREADS - 1) / FQL_THREADS), dim3(FQL_THREADS), 0, s, a);
                break;
            }
            case OP_CHAIN: {
                ChainArgs ca = L.op.chain; ca.tl = tl;
                {   // XCD-aware tile order of the chain launches (16-row tiles x 32-column tiles)
                    constexpr bool xcd_order = true;
                    const int ntm = ca.M / 16, ntn = cfg.actor_hidden[0] / 32;
                    ca.xg = 0;
                    if (xcd_order) {
                        constexpr int gm_env = 0;
                        double best = 1e30;
                        for (int gm : {1, 2, 4, 8}) {
                            const int gn = 8 / gm;
                            if (ntm % gm || ntn % gn || (gm_env && gm != gm_env)) continue;
                            const double cost = (double)gn * ca.M + (double)gm * cfg.actor_hidden[0];
                            if (cost < best) { best = cost; ca.xg = gm; }
                        }
                    }
                }
                if (cfg.precision == 2 && ca.variant != 3) {   // (variant D keeps fp32 operands: its dZ / W arrive as fp32 and splitting both in the kernel costs what the MFMAs save)
#define FQL_CHAIN_SPLIT(HH, VV) FQL_LAUNCH((fql_chain_split_kernel<HH, VV>), dim3((L.op.chain.M / 16) * (HH / 32)), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(HH), s, ca)
                    if (cfg.actor_hidden[0] == 512) { if (ca.variant == 0) FQL_CHAIN_SPLIT(512, 0); else if (ca.variant == 1) FQL_CHAIN_SPLIT(512, 1); else FQL_CHAIN_SPLIT(512, 2); }
                    else { if (ca.variant == 0) FQL_CHAIN_SPLIT(256, 0); else if (ca.variant == 1) FQL_CHAIN_SPLIT(256, 1); else FQL_CHAIN_SPLIT(256, 2); }
#undef FQL_CHAIN_SPLIT
                }
                else {
#define FQL_CHAIN_F32(HH, VV) FQL_LAUNCH((fql_chain_kernel<HH, VV>), dim3((L.op.chain.M / 16) * (HH / 32)), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(HH), s, ca)
                    if (cfg.actor_hidden[0] == 512) { if (ca.variant == 0) FQL_CHAIN_F32(512, 0); else if (ca.variant == 1) FQL_CHAIN_F32(512, 1); else if (ca.variant == 2) FQL_CHAIN_F32(512, 2); else FQL_CHAIN_F32(512, 3); }
                    else { if (ca.variant == 0) FQL_CHAIN_F32(256, 0); else if (ca.variant == 1) FQL_CHAIN_F32(256, 1); else if (ca.variant == 2) FQL_CHAIN_F32(256, 2); else FQL_CHAIN_F32(256, 3); }
#undef FQL_CHAIN_F32
                }
                break;
            }
            case OP_WFRAG:
                FQL_LAUNCH(fql_wfrag_kernel, dim3(wfrag_grid), dim3(FQL_THREADS), 0, s, (const WfragTask*)d_wfrag, wfrag_n, tl);
                break;
            case OP_DGRAD0: {
                Dgrad0Args a{};
                a.ntasks = (int)L.dg_tasks.size(); a.col0 = L.op.gemm.i1;
                for (int i = 0; i < a.ntasks; ++i) a.t[i] = L.dg_tasks[i];
                FQL_LAUNCH(fql_dgrad0_kernel, dim3(a.ntasks * (L.op.gemm.M / 16)), dim3(FQL_THREADS), 0, s, a);
                break;
            }
            case OP_HEAD_DGRAD:
                FQL_LAUNCH(fql_head_dgrad_kernel, dim3((L.op.gemm.M / 16) * (L.op.gemm.N / 64)), dim3(FQL_THREADS), 0, s, L.op.gemm);
                break;
            case OP_XCHAIN:   // one workgroup per CU: 8 XCDs x 32 members
                FQL_LAUNCH(fql_xchain_kernel, dim3(XCH_NGRP * XCH_NMEM), dim3(256), x_lds, s, L.op.xchain);
                break;
            case OP_ADAM: {
                AdamArgs a{P, G, Mu, Nu, P + n_train, d_chunks, d_state, d_partials, L.op.adam_c0, (int)critic_size, cfg.lr, cfg.tau, tl,
                           use_pec ? (const unsigned*)(pec_epoch + pec_teams) : (xsync ? (const unsigned*)(xsync + 16 * 32) : nullptr),
                           L.op.adam_n1 > 0 ? L.op.adam_n : -1, L.op.adam_c1};
                FQL_LAUNCH(fql_adam_kernel, dim3(L.op.adam_n < 0 ? n_chunks : L.op.adam_n + L.op.adam_n1), dim3(FQL_THREADS), 0, s, a);
                break;
            }
            case OP_FINALIZE:
                FQL_LAUNCH(fql_finalize_kernel, dim3(1), dim3(FQL_THREADS), 0, s,
                                   FinalizeArgs{d_state, d_chunks, d_partials, d_leaf_range, n_chunks, n_train_leaves, L.op.fin_mode, tl});
                break;
        }
    }

#undef FQL_LAUNCH
    // ---- threaded eager executor ---------------------------------------------------------------------------------------------
    // The lanes of a program issued as plain launches, each lane's stream fed by a host thread of its own (lane 0 by the caller).
    // Cross-lane dependencies: the producer records its event and then publishes the run's sequence number for that launch; the
    // consumer's thread waits for the number (so the event it makes its stream wait on is THIS run's record), then
    // hipStreamWaitEvent.  Nothing is captured, so the lane graph may hold edges in both directions.
    std::unique_ptr<LaneWorker> workers[FQL_LANES];
    std::atomic<bool> lanes_abort{false};

    void run_lane(Program& pr, int lane, hipStream_t s, uint64_t seq) {
        const bool diag = (&pr == &prog_full);
        if (lane != 0) HIP_CHECK(hipStreamWaitEvent(s, pr.ev_fork, 0));
        for (size_t i = 0; i < pr.launches.size(); ++i) {
            const Launch& L = pr.launches[i];
            if (L.lane != lane) continue;
            for (int w : L.waits) {
                if (!pr.launches[w].ev) continue;
                int spins = 0;
                while (pr.rec[w].load(std::memory_order_acquire) < seq) {
                    if (lanes_abort.load(std::memory_order_relaxed)) throw HipError{"threaded lanes: aborted after an error on another lane"};
                    if (++spins > 64) { __builtin_ia32_pause(); }
                }
                HIP_CHECK(hipStreamWaitEvent(s, pr.launches[w].ev, 0));
            }
            issue(L, s, diag ? (int)i : -1);
            if (L.record_after) {
                HIP_CHECK(hipEventRecord(L.ev, s));
                pr.rec[i].store(seq, std::memory_order_release);
            }
        }
        if (lane != 0) HIP_CHECK(hipEventRecord(pr.ev_join[lane], s));
        HIP_CHECK(hipGetLastError());
    }

    void worker_main(LaneWorker* w) {
        hipSetDevice(device);
        uint64_t last = 0;
        for (;;) {
            uint64_t j = last;
            for (int spin = 0; spin < 40000 && (j = w->job.load(std::memory_order_acquire)) == last && !w->stop.load(std::memory_order_relaxed); ++spin)
                __builtin_ia32_pause();
            if (j == last && !w->stop.load()) {
                std::unique_lock<std::mutex> lk(w->m);
                w->cv.wait(lk, [&] { return w->job.load(std::memory_order_acquire) != last || w->stop.load(); });
                j = w->job.load(std::memory_order_acquire);
            }
            if (w->stop.load()) return;
            try {
                run_lane(*w->pr, w->lane, w->s, w->pr->run_seq);
            } catch (const HipError& e) {
                w->err = e.msg;
                lanes_abort.store(true);
            }
            last = j;
            w->done.store(j, std::memory_order_release);
        }
    }

    void stop_workers() {
        for (auto& w : workers) {
            if (!w) continue;
            w->stop.store(true);
            { std::lock_guard<std::mutex> lk(w->m); }
            w->cv.notify_one();
            if (w->th.joinable()) w->th.join();
            w.reset();
        }
    }

    void run_threaded(Program& pr, hipStream_t s0) {
        if (!pr.two_lanes) { run_launches(pr, s0); return; }
        hipStream_t ls[FQL_LANES] = {s0, stream2, stream3, stream4};
        if (pr.rec_n != pr.launches.size()) {      // first run of this program: events and sequence slots
            pr.rec.reset(new std::atomic<uint64_t>[pr.launches.size()]);
            for (size_t i = 0; i < pr.launches.size(); ++i) pr.rec[i].store(0);
            pr.rec_n = pr.launches.size();
            pr.run_seq = 0;
            for (Launch& L : pr.launches)
                if (L.record_after && !L.ev) HIP_CHECK(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
            if (!pr.ev_fork) HIP_CHECK(hipEventCreateWithFlags(&pr.ev_fork, hipEventDisableTiming));
            for (int l = 1; l < FQL_LANES; ++l)
                if (pr.lane_used[l] && !pr.ev_join[l]) HIP_CHECK(hipEventCreateWithFlags(&pr.ev_join[l], hipEventDisableTiming));
        }
        const uint64_t seq = ++pr.run_seq;
        lanes_abort.store(false);
        HIP_CHECK(hipEventRecord(pr.ev_fork, s0));
        for (int l = 1; l < FQL_LANES; ++l) {
            if (!pr.lane_used[l]) continue;
            if (!workers[l]) {
                workers[l].reset(new LaneWorker);
                workers[l]->lane = l;
                workers[l]->th = std::thread([this, w = workers[l].get()] { worker_main(w); });
            }
            LaneWorker* w = workers[l].get();
            w->pr = &pr; w->s = ls[l]; w->err.clear();
            // job numbers are per worker (programs alternate), the run's sequence number travels in the program
            w->job.store(w->job.load(std::memory_order_relaxed) + 1, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(w->m); }
            w->cv.notify_one();
        }
        std::string err0;
        try { run_lane(pr, 0, s0, seq); } catch (const HipError& e) { err0 = e.msg; lanes_abort.store(true); }
        for (int l = 1; l < FQL_LANES; ++l) {
            if (!pr.lane_used[l]) continue;
            LaneWorker* w = workers[l].get();
            const uint64_t want = w->job.load(std::memory_order_relaxed);
            while (w->done.load(std::memory_order_acquire) != want) __builtin_ia32_pause();
            if (err0.empty() && !w->err.empty()) err0 = w->err;
            if (err0.empty()) HIP_CHECK(hipStreamWaitEvent(s0, pr.ev_join[l], 0));
        }
        if (!err0.empty()) throw HipError{err0};
    }

    // fork = true: the lanes of a multi-lane program on the lane streams (graph capture, or eager issue with FQL_NO_GRAPH=2); otherwise
    // everything goes to `s0` in emission order, which is a topological order of the program.
    void run_launches(Program& pr, hipStream_t s0, bool fork = false) {
        const bool par = fork && pr.two_lanes;
        hipStream_t ls[FQL_LANES] = {s0, stream2, stream3, stream4};
        if (par) {
            if (!pr.ev_fork) HIP_CHECK(hipEventCreateWithFlags(&pr.ev_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(pr.ev_fork, s0));
            for (int l = 1; l < FQL_LANES; ++l) if (pr.lane_used[l]) HIP_CHECK(hipStreamWaitEvent(ls[l], pr.ev_fork, 0));
        }
        static const bool trace_l = getenv("FQL_TRACE") != nullptr;
        for (Launch& L : pr.launches) {
            hipStream_t s = par ? ls[L.lane] : s0;
            if (trace_l) {
                fprintf(stderr, "[fql] launch %d type %d lane %d grid %d waits %zu:", (int)(&L - pr.launches.data()), (int)L.type, L.lane, L.grid, L.waits.size());
                for (int w : L.waits) fprintf(stderr, " %d(lane %d)", w, pr.launches[w].lane);
                fprintf(stderr, "\n");
            }
            if (par)
                for (int w : L.waits) if (pr.launches[w].ev) HIP_CHECK(hipStreamWaitEvent(s, pr.launches[w].ev, 0));
            issue(L, s, (&pr == &prog_full) ? (int)(&L - pr.launches.data()) : -1);
            if (par && L.record_after) {
                if (!L.ev) HIP_CHECK(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(L.ev, s));
            }
        }
        if (par)
            for (int l = 1; l < FQL_LANES; ++l) {
                if (!pr.lane_used[l]) continue;
                if (!pr.ev_join[l]) HIP_CHECK(hipEventCreateWithFlags(&pr.ev_join[l], hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(pr.ev_join[l], ls[l]));
                HIP_CHECK(hipStreamWaitEvent(s0, pr.ev_join[l], 0));
            }
        HIP_CHECK(hipGetLastError());
    }

    // The program as AQL packet templates for the engine's own HSA queues (fql_aql.h): one queue per lane, the barrier bit inside a lane,
    // completion signal -> barrier-AND packet across lanes.  Left off (ap.ok = false, the captured graph runs) when the HSA runtime does not
    // come up or FQL_AQL=0.
    void build_aql(Program& pr, AqlProgram& ap) {
        static const bool off = getenv("FQL_AQL") && atoi(getenv("FQL_AQL")) == 0;
        static const bool trace_a = getenv("FQL_TRACE") != nullptr;
        ap = AqlProgram{};
        if (off || visual) return;   // (visual updates gather their frames with HIP launches in front of the program and are device-bound anyway)
        void* kargs = nullptr;
        try {
            if (!aql.up && aql.why.empty()) aql.init(device, (const void*)&fql_create);
            if (!aql.up) throw AqlError{aql.why};
            std::vector<AqlDispatch> rec;
            aql_rec = &rec;
            try { for (const Launch& L : pr.launches) issue(L, nullptr, (&pr == &prog_full) ? (int)(&L - pr.launches.data()) : -1); } catch (...) { aql_rec = nullptr; throw; }   // (timeline ids: diagnostics build only)
            aql_rec = nullptr;
            const size_t n = pr.launches.size();
            if (rec.size() != n) throw AqlError{"a launch did not record exactly one dispatch"};

            std::vector<const AqlKernelInfo*> ki(n);
            std::vector<size_t> koff(n);
            size_t tot = 0;
            for (size_t i = 0; i < n; ++i) {
                ki[i] = &aql.kernel(rec[i].fn);
                if (ki[i]->kernarg_size < rec[i].args.size()) throw AqlError{"recorded arguments exceed the kernel's kernarg segment"};
                koff[i] = tot;
                tot += ((size_t)ki[i]->kernarg_size + 63) & ~(size_t)63;
            }
            std::vector<uint8_t> host(tot);
            for (size_t i = 0; i < n; ++i) AqlRuntime::fill_kernarg(host.data() + koff[i], rec[i], *ki[i]);
            HIP_CHECK(hipMalloc(&kargs, tot));   // device memory: with the blocks in host memory every launch starts 0.7 us later (417 us per update against 355)
            HIP_CHECK(hipMemcpy(kargs, host.data(), tot, hipMemcpyHostToDevice));
            std::vector<int> sig(n, AQL_SIG_NONE);
            int nsig = 0;
            bool used[FQL_AQL_LANES] = {};
            for (const Launch& L : pr.launches) {
                if (L.lane < 0 || L.lane >= FQL_AQL_LANES) throw AqlError{"lane out of range"};
                used[L.lane] = true;
                for (int w : L.waits) if (sig[w] == AQL_SIG_NONE) sig[w] = nsig++;
            }
            if (!used[0]) throw AqlError{"program without a lane 0"};
            int endsig[FQL_AQL_LANES];
            for (int l = 1; l < FQL_AQL_LANES; ++l) endsig[l] = used[l] ? nsig++ : AQL_SIG_NONE;
            // fences between packets: agent scope both ways, as the HIP runtime sets them for kernels of one stream.  Measured (profiles/r03_aql.txt): system scope
            // 416 us per update, agent 355, acquire dropped 336 with the same results over 1300 updates (the invalidate is then left to eviction: not a guarantee,
            // not shipped), release dropped 332 and wrong.
            constexpr int AG = HSA_FENCE_SCOPE_AGENT, SY = HSA_FENCE_SCOPE_SYSTEM, NO = HSA_FENCE_SCOPE_NONE;
            for (int l = 1; l < FQL_AQL_LANES; ++l)
                if (used[l]) {   // the previous update has finished on every lane
                    AqlPacket b = AqlRuntime::barrier_packet(false, NO, NO);
                    b.deps[0] = AQL_SIG_PREV_DONE;
                    ap.lane[l].push_back(b);
                }
            bool first[FQL_AQL_LANES] = {true, true, true, true};
            for (size_t i = 0; i < n; ++i) {
                const Launch& L = pr.launches[i];
                for (size_t w0 = 0; w0 < L.waits.size(); w0 += 5) {
                    AqlPacket b = AqlRuntime::barrier_packet(false, NO, NO);
                    for (size_t j = 0; j < 5 && w0 + j < L.waits.size(); ++j) b.deps[j] = sig[L.waits[w0 + j]];
                    ap.lane[L.lane].push_back(b);
                }
                AqlPacket k = AqlRuntime::kernel_packet(rec[i], *ki[i], (uint64_t)(uintptr_t)kargs + koff[i], true, first[L.lane] ? SY : AG, AG);
                first[L.lane] = false;
                k.complete = sig[i];
                ap.lane[L.lane].push_back(k);
            }
            for (int l = 1; l < FQL_AQL_LANES; ++l)
                if (used[l]) {
                    AqlPacket b = AqlRuntime::barrier_packet(true, NO, AG);
                    b.complete = endsig[l];
                    ap.lane[l].push_back(b);
                }
            {
                AqlPacket b = AqlRuntime::barrier_packet(true, NO, SY);
                int j = 0;
                for (int l = 1; l < FQL_AQL_LANES; ++l) if (used[l]) b.deps[j++] = endsig[l];
                b.complete = AQL_SIG_DONE;
                ap.lane[0].push_back(b);
            }
            aql.ensure_signals(nsig);
            for (int l = 0; l < FQL_AQL_LANES; ++l) if (used[l]) aql.ensure_queue(l);
            ap.nsig = nsig;
            ap.kernargs = kargs;
            ap.ok = true;
            if (trace_a) fprintf(stderr, "[fql] AQL program: %zu launches, packets per lane %zu %zu %zu %zu, %d signals, %zu bytes of kernargs\n", n,
                                 ap.lane[0].size(), ap.lane[1].size(), ap.lane[2].size(), ap.lane[3].size(), nsig, tot);
        } catch (const AqlError& e) {
            aql_rec = nullptr;
            if (kargs) hipFree(kargs);
            ap = AqlProgram{};
            if (aql.why.empty()) aql.why = e.msg;
            if (trace_a || getenv("FQL_AQL")) fprintf(stderr, "[fql] AQL path off: %s\n", e.msg.c_str());
        }
    }

    void capture(Program& pr) {
        if (pr.exec) { hipGraphExecDestroy(pr.exec); pr.exec = nullptr; }
        if (pr.graph) { hipGraphDestroy(pr.graph); pr.graph = nullptr; }
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        try {
            run_launches(pr, stream, true);
        } catch (...) {
            hipGraph_t g = nullptr;
            hipStreamEndCapture(stream, &g);
            if (g) hipGraphDestroy(g);
            throw;
        }
        static const bool trace_c = getenv("FQL_TRACE") != nullptr;
        if (trace_c) fprintf(stderr, "[fql] end capture\n");
        HIP_CHECK(hipStreamEndCapture(stream, &pr.graph));
        if (trace_c) {
            size_t nn = 0;
            hipGraphGetNodes(pr.graph, nullptr, &nn);
            fprintf(stderr, "[fql] instantiate (%zu nodes)\n", nn);
        }
        HIP_CHECK(hipGraphInstantiate(&pr.exec, pr.graph, nullptr, nullptr, 0));
        if (trace_c) fprintf(stderr, "[fql] instantiated\n");
    }

    // one graph per segment: seg 0 = lane-0 launches up to the last one lane 1 waits on, seg 1 = lane 1, seg 2 = lane 0 up to
    // its first wait on lane 1, seg 3 = the rest of lane 0.  Returns false if the program does not have that shape.
    bool capture_split(Program& pr) {
        int last_needed_by_lane1 = -1, first_lane0_wait = -1;
        for (int li = 0; li < (int)pr.launches.size(); ++li) {
            const Launch& L = pr.launches[li];
            if (L.lane > 1) return false;
            for (int w : L.waits) {
                if (L.lane == 1) last_needed_by_lane1 = std::max(last_needed_by_lane1, w);
                if (L.lane == 0 && first_lane0_wait < 0) first_lane0_wait = li;
            }
        }
        if (last_needed_by_lane1 < 0 || first_lane0_wait < 0 || first_lane0_wait <= last_needed_by_lane1) return false;
        auto seg_of = [&](int li) {
            const Launch& L = pr.launches[li];
            if (L.lane == 1) return 1;
            if (li <= last_needed_by_lane1) return 0;
            return li < first_lane0_wait ? 2 : 3;
        };
        for (int seg = 0; seg < 4; ++seg) {
            Program sub;
            for (int li = 0; li < (int)pr.launches.size(); ++li)
                if (seg_of(li) == seg) { Launch L = pr.launches[li]; L.waits.clear(); L.record_after = false; L.ev = nullptr; L.lane = 0; sub.launches.push_back(L); }
            if (sub.launches.empty()) return false;
            HIP_CHECK(hipStreamSynchronize(stream));
            HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            try {
                run_launches(sub, stream, false);
            } catch (...) {
                hipGraph_t g = nullptr;
                hipStreamEndCapture(stream, &g);
                if (g) hipGraphDestroy(g);
                throw;
            }
            HIP_CHECK(hipStreamEndCapture(stream, &split_graph[seg]));
            HIP_CHECK(hipGraphInstantiate(&split_exec[seg], split_graph[seg], nullptr, nullptr, 0));
        }
        if (!ev_a) HIP_CHECK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
        if (!ev_b) HIP_CHECK(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
        return true;
    }
    void free_split() {
        for (int i = 0; i < 4; ++i) {
            if (split_exec[i]) { hipGraphExecDestroy(split_exec[i]); split_exec[i] = nullptr; }
            if (split_graph[i]) { hipGraphDestroy(split_graph[i]); split_graph[i] = nullptr; }
        }
        split_ok = false;
    }
    // enqueue the split update: s0 carries lane 0, s1 lane 1; on return s1 holds everything bucket 0 of the gradient
    // buffer depends on, s0 everything bucket 1 depends on
    void launch_split(hipStream_t s0, hipStream_t s1) {
        HIP_CHECK(hipGraphLaunch(split_exec[0], s0));
        HIP_CHECK(hipEventRecord(ev_a, s0));
        HIP_CHECK(hipStreamWaitEvent(s1, ev_a, 0));
        HIP_CHECK(hipGraphLaunch(split_exec[1], s1));
        HIP_CHECK(hipEventRecord(ev_b, s1));
        HIP_CHECK(hipGraphLaunch(split_exec[2], s0));
        if (!ev_c) HIP_CHECK(hipEventCreateWithFlags(&ev_c, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev_c, s0));          // the Euler chain (the last reader of the BC flow's weights on lane 0) is behind this
        HIP_CHECK(hipStreamWaitEvent(s0, ev_b, 0));
        HIP_CHECK(hipGraphLaunch(split_exec[3], s0));
        split_begun = true;
    }

    void free_program(Program& pr) {
        if (pr.exec) hipGraphExecDestroy(pr.exec);
        if (pr.graph) hipGraphDestroy(pr.graph);
        for (Launch& L : pr.launches) if (L.ev) hipEventDestroy(L.ev);
        if (pr.ev_fork) hipEventDestroy(pr.ev_fork);
        for (hipEvent_t e : pr.ev_join) if (e) hipEventDestroy(e);
        pr = Program{};
    }

    // ---------------------------------------------------------------------------------------
    // the training step as a program (agents/fql.py:94-133)
    // ---------------------------------------------------------------------------------------
    void build_step_program(Program& pr, bool with_grads) {
        const int od = cfg.obs_dim, ad = cfg.act_dim;
        const int inp_c = nets[NET_OS].in_p(), inp_b = nets[NET_BC].in_p();
        const int ap = pad16(ad);
        DevState* st = d_state;
        const void* INFO = &st->info[0];
        const void *I_CR = &st->info[0], *I_BC = &st->info[5], *I_Q = &st->info[7], *I_MSE = &st->info[9], *I_ACT = &st->info[4];
        // Lane 0 carries the critical path (prep -> Euler chain -> actor loss -> one-step backward); the rest of
        // the step runs beside it on lane 1 (a second graph branch) and only meets it at the actor loss.
        if (visual) {   // every module encodes the batch images first (agents/fql.py:196-202); the encodings feed prep
            // (data-parallel split program: on lane 0, so that lane 1 depends on lane 0 only through the prep kernel)
            place("enc", split_build ? 0 : 1, true);
            emit_encoder_forward(pr, eb_os);   // [obs ; next_obs]: sample_actions(next_obs) fql.py:25, onestep(obs) fql.py:65,82
            emit_encoder_forward(pr, eb_c);    // critic(obs, .) fql.py:36,70 (same stored encoder and images: one pass)
            emit_encoder_forward(pr, eb_t);    // target_critic(next_obs, .) fql.py:28
            emit_encoder_forward(pr, eb_bc);   // actor_bc_flow(obs, .) fql.py:58 and actor_bc_flow_encoder(obs) fql.py:163 (shared)
        }
        emit_lane = 0;
        {   // batch gather + noise + every network input
            Op op{};
            op.type = OP_PREP;
            op.prep = PrepArgs{d_src, st, seed, B, od, ad, inp_c, inp_b, ap, X_os, X_bc, X_eu, X_c1, X_c2, X_ct, vel, w_rew, w_mask, w_act,
                               fused_euler ? X_e0 : nullptr, nullptr, nullptr, nullptr, nullptr};
            op.prep.tl = -1; op.prep.part = 0; op.prep.xsync = use_xchain ? xsync : nullptr;
            // One batch-assembly launch per lane (state agents): the side lane then has no dependency on the critical lane at the
            // start of the update.  (On this runtime a graph branch whose first node waits for a node of the other branch starts
            // only ~16 launches of that branch later: profiles/r02_timeline_concurrent.txt.)
            constexpr bool split_prep = true;
            if (visual) {
                op.prep.E_c = eb_c.E; op.prep.E_t = eb_t.E; op.prep.E_bc = eb_bc.E; op.prep.E_os = eb_os.E;
                op.reads = {eb_c.E, eb_t.E, eb_bc.E, eb_os.E};
            }
            if (split_prep && !visual && !split_build) {
                Op a = op, b = op;
                a.prep.part = 1;
                a.writes = {X_eu};
                if (fused_euler) a.writes.push_back(X_e0);
                b.prep.part = 2;
                b.writes = {X_os, X_bc, X_c1, X_c2, X_ct, vel, w_rew, w_mask, w_act};
                emit_lane = 0; push(pr, a);
                emit_lane = 1; push(pr, b);
                emit_lane = 0;
            } else {
                op.writes = {X_os, X_bc, X_eu, X_c1, X_c2, X_ct, vel, w_rew, w_mask, w_act};
                if (fused_euler) op.writes.push_back(X_e0);
                push(pr, op);
            }
        }
        // Three lanes (state agents, single-GPU programs): lane 1 carries only what the one-step actor's backward waits for - the
        // Q-gradient path (one-step forward, critic(obs, actor actions) forward and input-gradient chain) - in light launches;
        // the critic-loss passes, the BC pass, every weight gradient and Adam go to lane 2.  Edges between the two forked
        // lanes run 1 -> 2 only: torch's bundled HIP runtime recurses forever in hipStreamEndCapture when two forked streams wait
        // on each other in BOTH directions (hip::Stream::EndCapture walks the cycle; the system ROCm 7.2 runtime does not).
        constexpr bool lanes3_env = true;
        const bool lanes3 = lanes3_env && !visual && !split_build;
        const int fill_lane = lanes3 ? 2 : 1;
        if (!split_build) fill_lane_full = fill_lane;
        wide_tiles = lanes3;
        place("os", 1, true);
        // one-step actor on [next_obs|eps1 ; obs|z ; obs|eps2]  (agents/fql.py:25,65,82)
        emit_forward(pr, p_os, with_grads, GF_OS_SCATTER, X_ct, X_c2);
        {   // mse metric only (agents/fql.py:82-83); the clipped actions were scattered by the head's epilogue
            Op op{};
            op.type = OP_POSTOS;
            op.postos = PostOsArgs{p_os.out, w_act, X_ct, X_c2, st, B, od, ad, inp_c, ap};
            op.reads = {p_os.out, w_act};
            op.writes = {I_MSE};
            push(pr, op);
        }
        place("c1f", fill_lane, true);   // (forward and loss / backward placed separately: FQL_LANE_c1f / FQL_LANE_c1)
        // critic(obs, actions) with grad params; target critic(next_obs, next_actions)  (fql.py:28,36)
        for (int e = 0; e < 2; ++e) emit_forward(pr, p_c1[e], with_grads);
        // three lanes: the target-critic pass rides on lane 1 (its action block comes from the one-step forward there, and lane 1 is otherwise
        // idle once the Q-gradient path has ended): 2500 -> 2580 updates/s in steady state
        constexpr int ct_lane3 = -1;   // (three-lane programs only: the others have no lane 2)
        place("ct", lanes3 ? (ct_lane3 >= 0 ? ct_lane3 : 1) : fill_lane, true);
        for (int e = 0; e < 2; ++e) emit_forward(pr, p_ct[e], false);
        place("c1", fill_lane, true);
        {
            Op op{};
            op.type = OP_LOSS_CRITIC;
            op.lc = LossCriticArgs{p_c1[0].out, p_c1[1].out, p_ct[0].out, p_ct[1].out, w_rew, w_mask,
                                   with_grads ? p_c1[0].dz.back() : nullptr, with_grads ? p_c1[1].dz.back() : nullptr,
                                   st, B, cfg.q_agg, with_grads ? 1 : 0, cfg.discount};
            op.reads = {p_c1[0].out, p_c1[1].out, p_ct[0].out, p_ct[1].out, w_rew, w_mask};
            op.writes = {I_CR};
            if (with_grads) { op.writes.push_back(p_c1[0].dz.back()); op.writes.push_back(p_c1[1].dz.back()); }
            push(pr, op);
        }
        // The critic's weight gradients gate nothing but Adam: they are emitted after the Q-gradient chain, whose levels on lane 1
        // then carry fewer tiles and finish earlier - and with them the one-step actor's backward tail (2127 -> 2170 updates/s;
        // deferring the BC flow's too: 2161, the critic's whole backward chain: 2050).  FQL_LATE_WGRAD: 0 off, 1 both, 2 critic, 3 bc.
        constexpr int late_wgrad_env = -1;
        const int late_wgrad = late_wgrad_env >= 0 ? late_wgrad_env : (lanes3 ? 0 : 2);   // (three lanes: they are on lane 2, out of the Q-gradient path's launches anyway)
        std::vector<Op> late_ops;
        if ((late_wgrad == 1 || late_wgrad == 2) && with_grads) defer_wgrads = &late_ops;
        if (with_grads)
            for (int e = 0; e < 2; ++e) emit_backward(pr, p_c1[e], 0, B, true, visual);
        // (not in the data-parallel split program: there lane 1 must finish bucket 0 without waiting for lane 0's tail)
        const bool enc_align = !split_build;
        if (with_grads && visual && !enc_align) {   // the critic's encoder sees the critic loss only (the actor loss uses stored params)
            place("enc", 1, true);
            emit_encoder_backward(pr, eb_c, 0, B, p_c1[0].dx0, p_c1[1].dx0, nets[NET_C0].in_p());
        }
        // BC flow-matching pass (fql.py:52-59).  FQL_BC_LATE=<lane> (three-lane programs): emitted BEHIND the Q-gradient path on that lane instead of
        // on lane 2 in front of it - lane 1 idles from the end of the Q-gradient path to the end of the update while lane 2 carries the longest tail.
        constexpr int bc_late_env = 0;
        const int bc_late = (lanes3 && with_grads) ? bc_late_env : 0;
        auto emit_bc = [&](int bc_lane) {
            // (three lanes: the BC flow's FORWARD on lane 1 in front of the one-step pass, its loss and backward stay on lane 2 - -0.5 .. -1.0 % per
            // fp32 update in three in-process A/Bs under AQL dispatch, nothing in bf16x3; experiments/lane_sweep.sh)
            place("bcf", lanes3 ? 1 : bc_lane, true);
            emit_forward(pr, p_bc, with_grads);
            place("bc", bc_lane, true);
            {
                Op op{};
                op.type = OP_LOSS_BC;
                op.lb = LossBcArgs{p_bc.out, vel, with_grads ? p_bc.dz.back() : nullptr, st, B, ad, ap, with_grads ? 1 : 0};
                op.reads = {p_bc.out, vel};
                op.writes = {I_BC};
                if (with_grads) op.writes.push_back(p_bc.dz.back());
                push(pr, op);
            }
            defer_wgrads = ((late_wgrad == 1 || late_wgrad == 3) && with_grads) ? &late_ops : nullptr;
            if (with_grads) emit_backward(pr, p_bc, 0, B, true, visual);
            if (with_grads && visual && !enc_align) {
                place("enc", 1, true);
                emit_encoder_backward(pr, eb_bc, 0, B, p_bc.dx0, nullptr, nets[NET_BC].in_p());
            }
        };
        if (!bc_late) emit_bc(fill_lane);
        // Q term: critic(obs, clip(actor_actions)) with stored params, input-differentiable (fql.py:69-76)
        place("c2", 1, true);
        for (int e = 0; e < 2; ++e) emit_forward(pr, p_c2[e], with_grads);
        {
            Op op{};
            op.type = OP_LOSS_Q;
            // without normalize_q_loss dQ = -1/(2B) is a constant (filled once at workspace build): the input-gradient
            // chain through the critic then does not wait for the Q heads or this kernel
            const bool dyn_dq = with_grads && cfg.normalize_q_loss;
            op.lq = LossQArgs{p_c2[0].out, p_c2[1].out, dyn_dq ? p_c2[0].dz.back() : nullptr,
                              dyn_dq ? p_c2[1].dz.back() : nullptr, st, B, cfg.normalize_q_loss, dyn_dq ? 1 : 0};
            op.reads = {p_c2[0].out, p_c2[1].out};
            op.writes = {I_Q};
            if (dyn_dq) { op.writes.push_back(p_c2[0].dz.back()); op.writes.push_back(p_c2[1].dz.back()); }
            push(pr, op);
        }
        defer_wgrads = nullptr;
        if (with_grads) {
            // two lanes: in phase with the c1 chain (shared launches); three lanes: on its own, as early as its inputs exist
            constexpr bool c2_align = true;
            for (int e = 0; e < 2; ++e) emit_backward(pr, p_c2[e], 0, B, false, true, (c2_align && !lanes3) ? I_CR : nullptr);
        }
        for (Op& w : late_ops) {   // the critic's and the BC flow's weight gradients only after the Q-gradient chain (they gate nothing)
            w.reads.push_back(p_c2[0].dx0); w.reads.push_back(p_c2[1].dx0);
            emit_lane = w.lane;
            push(pr, w);
        }
        if (bc_late) emit_bc(bc_late);
        // Euler chain through the BC flow (fql.py:155-171): flow_steps sequential forwards
        place("eu", 0, false);
        const int fs = cfg.flow_steps;
        // The last Euler step's target is finished inside the one-step head dgrad's prologue (GF_A_EULFIN: one launch less on the critical lane,
        // bit-identical target).  The 16 partial loads per element lengthen the prologue of all 256 workgroups by most of what the launch boundary
        // saves: -0.45 % per update in both orders of an in-process A/B (experiments/ab_inproc.py; round 2, under the graph's host cost, saw nothing).
        constexpr bool fuse_ef_env = true;
        euler_finish_fused = fuse_ef_env && with_grads && !cfg.actor_layer_norm && !use_pec && !use_xchain && fused_euler && use_chain &&
                             fs > 1 && vp_tiles <= 32;
        if (use_xchain) emit_euler_xcd(pr);
        else if (use_pec) emit_euler_persistent(pr);
        else if (fused_euler && use_chain) emit_euler_chain(pr);
        else if (fused_euler) emit_euler_fused(pr);
        else
        for (int s = 0; s < fs; ++s)
            emit_forward(pr, p_eu, false, GF_EULER | (s == fs - 1 ? GF_EULER_LAST : 0), X_eu, tgt, 1.0f / (float)fs,
                         (float)(s + 1) / (float)fs);
        // The actor-loss gradient is built inside the one-step actor's head dgrad (GF_A_LOSSACT), so the loss kernel only
        // reports scalars and leaves the critical path (it rides on lane 1): 2083 -> 2130 updates/s.
        constexpr bool fuse_la_env = true;
        const bool fuse_la = with_grads && fuse_la_env && !cfg.actor_layer_norm;
        Op la_op{};
        int la_lane = 0;
        {
            Op op{};
            op.type = OP_LOSS_ACTOR;
            const bool kgrad = with_grads && !fuse_la;
            op.la = LossActorArgs{p_os.out + (size_t)B * ap, tgt, kgrad ? p_c2[0].dx0 : nullptr,
                                  kgrad ? p_c2[1].dx0 : nullptr, kgrad ? p_os_bwd.dz.back() : nullptr, st, B, od, ad,
                                  inp_c, ap, kgrad ? 1 : 0, cfg.alpha};
            op.reads = {p_os.out, tgt, I_BC, I_Q};
            op.writes = {I_ACT};
            if (kgrad) {
                op.reads.push_back(p_c2[0].dx0); op.reads.push_back(p_c2[1].dx0);
                op.writes.push_back(p_os_bwd.dz.back());
            }
            // FQL_TAIL_MERGE (see adam_for; default on): this kernel, first of lane 2's tail, also takes the edge to the end of lane 1's Q-gradient chain that
            // the critic's Adam behind it needs (write after read of the critic's kernels) - both cross-lane waits of the tail on ONE launch
            constexpr bool tail_merge = true;   // default on
            if (tail_merge && with_grads && !kgrad) { op.reads.push_back(p_c2[0].dx0); op.reads.push_back(p_c2[1].dx0); }
            const int keep = emit_lane;
            if (fuse_la && !split_build) emit_lane = fill_lane;
            if (euler_finish_fused) { la_op = op; la_lane = emit_lane; }   // pushed behind the head dgrad, which now WRITES the target
            else push(pr, op);
            emit_lane = keep;
        }
        if (with_grads) {
            // The one-step actor's weight gradients gate the last Adam launch.  On lane 1 they would queue behind the deferred critic
            // weight gradients and the other two Adam launches; emitted here as ONE launch on lane 0 behind the last dgrad they
            // cost ~10 us of tail instead.  (FQL_OS_WGRAD_SIDE=1: the old placement.)
            constexpr bool os_w_side = false;
            std::vector<Op> os_w;
            if (!os_w_side) defer_wgrads = &os_w;
            const size_t first = pr.ops.size();
            emit_backward(pr, p_os_bwd, B, B, true, visual);
            defer_wgrads = nullptr;
            if (fuse_la) {
                // the head's wgrad (emitted first) reads dA, which the head dgrad (emitted right after it) now PRODUCES:
                // swap them so the scheduler sees the write before the read, then turn the dgrad's A operand into the builder
                float* da = p_os_bwd.dz.back();
                size_t id = first;
                while (id < pr.ops.size() && !(pr.ops[id].type == OP_GEMM && pr.ops[id].gemm.A == da)) ++id;
                if (id >= pr.ops.size() || id > first + 1) invalid("internal: unexpected one-step backward program shape");
                if (id == first + 1) {   // (when the wgrads are emitted in place; deferred ones follow the chain anyway)
                    if (pr.ops[first].type != OP_WGRAD || pr.ops[first].wgrad.dZ != da) invalid("internal: unexpected one-step backward program shape");
                    std::swap(pr.ops[first], pr.ops[first + 1]);
                }
                Op& d = pr.ops[first];
                GemmTask& t = d.gemm;
                t.flags |= GF_A_LOSSACT;
                t.ea_in = p_os.out + (size_t)B * ap; t.evp = tgt; t.i0 = ap; t.i1 = od; t.i2 = ad;
                t.ew = p_c2[0].dx0; t.ew4 = p_c2[1].dx0; t.e_ntp = inp_c;
                t.ea_out = da;
                // the dedicated launch (fql_head_dgrad_kernel) where the task has its shape: a 16-wide contraction into multiples of 64 columns
                if (t.K == 16 && t.N % 64 == 0 && t.M % 16 == 0 && (t.flags & GF_TRANS_B) && (t.flags & GF_GELUGRAD) && !(t.flags & (GF_BIAS | GF_RELUGRAD | GF_A_LN)) &&
                    cfg.precision != 1)
                    d.type = OP_HEAD_DGRAD;
                t.f0 = cfg.alpha * 2.0f / (float)(B * ad);
                d.reads.erase(std::remove(d.reads.begin(), d.reads.end(), (const void*)da), d.reads.end());
                for (const void* r : {(const void*)p_os.out, (const void*)tgt, (const void*)p_c2[0].dx0, (const void*)p_c2[1].dx0}) d.reads.push_back(r);
                d.writes.push_back(da);
                if (euler_finish_fused) {
                    const Net& nb = nets[NET_BC];
                    t.flags |= GF_A_EULFIN;
                    t.aux = Abuf[(fs - 1) & 1]; t.aux2 = Vpart; t.eb = P + nb.layers[nb.nl() - 1].b; t.f1 = 1.0f / (float)fs; t.ln_width = vp_tiles;
                    d.reads.erase(std::remove(d.reads.begin(), d.reads.end(), (const void*)tgt), d.reads.end());
                    for (const void* r : {(const void*)Abuf[(fs - 1) & 1], (const void*)Vpart, (const void*)(P + nb.layers[nb.nl() - 1].w)}) d.reads.push_back(r);   // (head bias: WAR against Adam)
                    d.writes.push_back(tgt);
                }
            }
            if (euler_finish_fused) {   // the metric kernel reads the target the head dgrad has just written
                const int keep = emit_lane;
                emit_lane = la_lane;
                push(pr, la_op);
                emit_lane = keep;
            }
            // Three lanes: lane 2 is idle by the time the tail runs, so each layer's weight gradient goes there as soon as its dz exists
            // (beside the remaining dgrads of the tail); only the first layer's small one is left behind the last dgrad.
            constexpr int os_w_l2_env = -1;
            const bool os_w_l2 = lanes3 && (os_w_l2_env >= 0 ? os_w_l2_env != 0 : false);
            for (Op& w : os_w) {
                if (!os_w_l2) w.reads.push_back(p_os_bwd.dz[0]);   // after the last dgrad: all five in one launch
                emit_lane = os_w_l2 ? 2 : 0;
                push(pr, w);
            }
            emit_lane = 0;
        }
        if (with_grads && visual) {   // the obs half of the [obs ; next_obs] pass
            place("enc", split_build ? 0 : 1, true);
            emit_encoder_backward(pr, eb_os, 0, B, p_os_bwd.dx0, nullptr, nets[NET_OS].in_p());
            if (enc_align) {   // all three encoder backward passes level-aligned with the last one: their ops share launches
                emit_encoder_backward(pr, eb_c, 0, B, p_c1[0].dx0, p_c1[1].dx0, nets[NET_C0].in_p(), p_os_bwd.dx0);
                emit_encoder_backward(pr, eb_bc, 0, B, p_bc.dx0, nullptr, nets[NET_BC].in_p(), p_os_bwd.dx0);
            }
        }
        if (!with_grads) {
            Op op{};
            op.type = OP_FINALIZE;
            op.fin_mode = 0;
            op.reads = {I_CR, I_BC, I_Q, I_MSE, I_ACT};
            op.writes = {INFO};
            push(pr, op);
        }
    }

    // Adam (+ Polyak for the critic) of one module (0 BC flow, 1 one-step actor, 2 critic) as its own launch on `lane`
    // m2 >= 0: a second module (its nets in net_ids as well) rides in the same launch
    void adam_for(Program& pr, int m, std::initializer_list<int> net_ids, int lane, int m2 = -1) {
        DevState* st = d_state;
            Op a{};
            a.type = OP_ADAM;
            a.adam_c0 = mod_chunk0[m]; a.adam_n = mod_chunkn[m];
            a.reads = {st};
            a.writes = {d_partials + mod_chunk0[m] * 4};
            if (m2 >= 0) { a.adam_c1 = mod_chunk0[m2]; a.adam_n1 = mod_chunkn[m2]; a.writes.push_back(d_partials + mod_chunk0[m2] * 4); }
            if (m2 == -2) {   // all chunks (the modules' ranges tile [0, n_chunks))
                a.adam_c0 = 0; a.adam_n = -1;
                for (int mm = 0; mm < 3; ++mm) a.writes.push_back(d_partials + mod_chunk0[mm] * 4);
            }
            // FQL_TAIL_MERGE (default on; =0 off): the critic's Adam, the first launch of lane 2's tail, also waits for the Euler target the actor-loss kernel behind it reads -
            // one launch with two cross-lane edges instead of two launches with one each (every such edge costs the waiting lane ~10 us)
            constexpr bool tail_merge = true;   // default on
            if (tail_merge && m == 2 && tgt) a.reads.push_back(tgt);
            for (int mm : {m, m2}) if (visual && mm >= 0) {
                const int m = mm;
                const int ei = m == 2 ? ENC_C : (m == 0 ? ENC_BC : ENC_OS);
                for (const EncStack& st : encs[ei].stacks)
                    for (const ConvL& c : st.conv) {
                        a.reads.push_back(G + c.w); a.reads.push_back(G + c.b);
                        a.writes.push_back(P + c.w);
                        if (m == 2) a.writes.push_back(P + n_train + c.w);
                    }
                const Layer& D = encs[ei].dense;
                a.reads.push_back(G + D.w); a.reads.push_back(G + D.b);
                a.writes.push_back(P + D.w);
                if (m == 2) a.writes.push_back(P + n_train + D.w);
            }
            for (int ni : net_ids)
                for (const Layer& L : nets[ni].layers) {
                    a.reads.push_back(G + L.w); a.reads.push_back(G + L.b);
                    a.writes.push_back(P + L.w);
                    if (L.ln) { a.reads.push_back(G + L.g); a.reads.push_back(G + L.be); a.writes.push_back(P + L.g); }
                    if (ni == NET_C0 || ni == NET_C1) a.writes.push_back(P + n_train + L.w);  // Polyak target
                }
            emit_lane = lane;
            push(pr, a);
        }

    // The optimizer half of the begin / end pair (data-parallel step: gradients are all-reduced between the halves): per-module Adam
    // as in the fused program - critic and BC flow (gradient bucket 0) + the chain's weight-copy refresh on lane 1, the one-step
    // actor (bucket 1) and the bookkeeping on lane 0.  fql_update_end captures it as one two-lane graph; fql_update_end_split issues
    // lane 1 on the stream that carried bucket 0's all-reduce, so those Adam launches overlap lane 0's tail and bucket 1's reduce.
    // fql_update_end runs it as ONE lane (a captured two-lane graph of five launches pays more for its fork and join than the two Adam launches
    // overlap: the plain data-parallel step at world size 1, collective forced, 2167 -> 2298 updates/s).
    void build_opt_program(Program& pr, bool two_lanes) {
        DevState* st = d_state;
        if (two_lanes) {
            adam_for(pr, 2, {NET_C0, NET_C1}, 1);
            adam_for(pr, 0, {NET_BC}, 1);
            if (use_chain) { emit_lane = 1; emit_wfrag(pr); }
            adam_for(pr, 1, {NET_OS}, 0);
        } else {   // every chunk of every module in ONE Adam launch, then the copies' refresh, then the bookkeeping
            adam_for(pr, 2, {NET_C0, NET_C1, NET_BC, NET_OS}, 0, -2);
            if (use_chain) { emit_lane = 0; emit_wfrag(pr); }
        }
        Op f{};
        f.type = OP_FINALIZE;
        f.fin_mode = 1;
        f.reads = {d_partials + mod_chunk0[0] * 4, d_partials + mod_chunk0[1] * 4, d_partials + mod_chunk0[2] * 4, &st->info[0],
                   &st->info[4], &st->info[5], &st->info[7], &st->info[9]};
        f.writes = {&st->info[10], st};
        emit_lane = 0;
        push(pr, f);
    }
    // the optimizer program as plain launches on two caller streams (no capture: five launches)
    void run_opt_split(hipStream_t s0, hipStream_t s1) {
        Program& pr = prog_opt_split;
        // lane 1 rewrites the critic's and the BC flow's parameters (and the chain's weight copies): it must not pass lane 0's last
        // reader of them, the Euler chain.  After a split begin that point is ev_c; otherwise everything enqueued on s0 so far.
        if (!ev_c) HIP_CHECK(hipEventCreateWithFlags(&ev_c, hipEventDisableTiming));
        if (!split_begun) HIP_CHECK(hipEventRecord(ev_c, s0));
        HIP_CHECK(hipStreamWaitEvent(s1, ev_c, 0));
        split_begun = false;
        for (Launch& L : pr.launches) {
            hipStream_t s = L.lane == 0 ? s0 : s1;
            for (int w : L.waits) if (pr.launches[w].ev) HIP_CHECK(hipStreamWaitEvent(s, pr.launches[w].ev, 0));
            issue(L, s, -1);
            if (L.record_after) {
                if (!L.ev) HIP_CHECK(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(L.ev, s));
            }
        }
        if (pr.lane_used[1]) {   // s0 ends behind everything of the update
            if (!pr.ev_join[1]) HIP_CHECK(hipEventCreateWithFlags(&pr.ev_join[1], hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(pr.ev_join[1], s1));
            HIP_CHECK(hipStreamWaitEvent(s0, pr.ev_join[1], 0));
        }
        HIP_CHECK(hipGetLastError());
    }

    // fwd + bwd + optimizer in one graph (single-GPU calls): Adam of a module is issued on the lane that produced its
    // gradients as soon as they exist, so 3/4 of the optimizer pass overlaps the tail of the critical lane
    void build_full_program(Program& pr) {
        build_step_program(pr, true);
        DevState* st = d_state;
        // the critic's and the BC flow's Adam as ONE launch (FQL_ADAM_MERGE=0: two): both sit at the end of the same lane behind the same
        // dependencies, and every graph node costs the host 3-4 us
        constexpr bool adam_merge = true;
        if (adam_merge) adam_for(pr, 2, {NET_C0, NET_C1, NET_BC}, fill_lane_full, 0);
        else {
            adam_for(pr, 2, {NET_C0, NET_C1}, fill_lane_full);
            adam_for(pr, 0, {NET_BC}, fill_lane_full);
        }
        emit_wfrag(pr);   // lane 1, behind the BC flow's Adam: the next update's chain reads the copies
        adam_for(pr, 1, {NET_OS}, 0);
        Op f{};
        f.type = OP_FINALIZE;
        f.fin_mode = 1;
        f.reads = {d_partials + mod_chunk0[0] * 4, d_partials + mod_chunk0[1] * 4, d_partials + mod_chunk0[2] * 4, &st->info[0],
                   &st->info[4], &st->info[5], &st->info[7], &st->info[9]};
        f.writes = {&st->info[10], st};
        emit_lane = 0;
        push(pr, f);
    }

    int64_t macs_per_update() const {
        auto macs = [&](const Net& n) { int64_t m = 0; for (const Layer& L : n.layers) m += (int64_t)L.in * L.out; return m; };
        auto first = [&](const Net& n) { return (int64_t)n.layers[0].in * n.layers[0].out; };
        const int64_t os = macs(nets[NET_OS]), bc = macs(nets[NET_BC]), cr = macs(nets[NET_C0]);
        const int64_t fwd = 3 * os + (cfg.flow_steps + 1) * bc + 6 * cr;
        const int64_t bwd = 2 * (2 * cr - first(nets[NET_C0])) + (2 * bc - first(nets[NET_BC])) + (2 * os - first(nets[NET_OS])) + 2 * cr;
        int64_t total = (fwd + bwd) * (int64_t)B;
        if (visual) {
            // encoder: 5 distinct forward image-batches per update (onestep on obs and next_obs, critic, target, bc_flow; the
            // reference's other three encoder calls repeat one of these on the same images and parameters) and 3 backward
            // passes = dgrad (none into the images) + wgrad; plus the MLPs' layer-0 input gradients that feed them
            const EncNet& en = encs[ENC_C];
            int64_t f = (int64_t)en.dense.in * en.dense.out, first_conv = 0;
            for (size_t s = 0; s < en.stacks.size(); ++s)
                for (size_t j = 0; j < en.stacks[s].conv.size(); ++j) {
                    const ConvL& c = en.stacks[s].conv[j];
                    const int H = j == 0 ? en.stacks[s].H : en.stacks[s].H / 2, W = j == 0 ? en.stacks[s].W : en.stacks[s].W / 2;
                    const int64_t m = (int64_t)H * W * 9 * c.cin * c.cout;
                    f += m;
                    if (s == 0 && j == 0) first_conv = m;
                }
            total += (int64_t)B * (5 * f + 3 * (2 * f - first_conv));
            total += (int64_t)B * enc_dim * (2 * (int64_t)nets[NET_C0].layers[0].out + nets[NET_BC].layers[0].out + nets[NET_OS].layers[0].out);
        }
        return total;
    }

    void free_workspace() {
        free_program(prog_fwdbwd);
        free_program(prog_opt);
        free_program(prog_opt_split);
        free_program(prog_loss);
        free_program(prog_split);
        free_program(prog_full);
        aql_drain();
        if (aql_full.kernargs) { hipFree(aql_full.kernargs); }
        aql_full = AqlProgram{};
        use_xchain = false; xsync = nullptr; xvp = nullptr; x_stamps = nullptr;
        free_split();
        for (void* p : ws_allocs) hipFree(p);
        ws_allocs.clear();
    }

    void build_workspace(int batch) {
        if (batch <= 0 || batch % 16 != 0) invalid("batch_size must be a positive multiple of 16 (got %d)", batch);
        HIP_CHECK(hipStreamSynchronize(stream));
        free_workspace();
        B = batch;
        const int od = cfg.obs_dim, ad = cfg.act_dim;
        const int inp_c = nets[NET_OS].in_p(), inp_b = nets[NET_BC].in_p();
        auto& W = ws_allocs;
        in_obs = dalloc(W, (size_t)B * od); in_nobs = dalloc(W, (size_t)B * od);
        in_act = dalloc(W, (size_t)B * ad); in_rew = dalloc(W, B); in_mask = dalloc(W, B);
        for (int i = 0; i < 5; ++i) in_noise[i] = dalloc(W, (size_t)B * ad);
        in_idx = (int64_t*)dalloc(W, (size_t)B * 4);   // [2][B]: indices in use, staging of host indices (frames path)
        X_os = dalloc(W, (size_t)3 * B * inp_c); X_bc = dalloc(W, (size_t)B * inp_b); X_eu = dalloc(W, (size_t)B * inp_b);
        X_c1 = dalloc(W, (size_t)B * inp_c); X_c2 = dalloc(W, (size_t)B * inp_c); X_ct = dalloc(W, (size_t)B * inp_c);
        const int ap = pad16(ad);
        vel = dalloc(W, (size_t)B * ap); w_act = dalloc(W, (size_t)B * ap); tgt = dalloc(W, (size_t)B * ap);
        w_rew = dalloc(W, B); w_mask = dalloc(W, B);
        {   // fused Euler chain (layers 0+1 and last-hidden+head per launch): plain actor MLPs with >= 2 hidden layers
            const Net& nb = nets[NET_BC];
            const int nh = nb.nl() - 1;
            fused_euler = !cfg.actor_layer_norm && ad <= 15 && nh >= 3;   // 16-wide rank update: act + t
            if (fused_euler) {
                X_e0 = dalloc(W, (size_t)B * inp_b);
                C0 = dalloc(W, (size_t)B * nb.layers[0].out_p);
                Abuf[0] = dalloc(W, (size_t)B * ap); Abuf[1] = dalloc(W, (size_t)B * ap);
                vp_tiles = (nb.layers[nh - 1].out_p / 16 + 1) / 2;
                if (vp_tiles > 32) fused_euler = false;
                Vpart = dalloc(W, (size_t)std::max(vp_tiles, 1) * B * ap);
                // persistent chain: one launch for all flow_steps x layers; needs every workgroup resident (1 per CU)
                const int H = nb.layers[0].out_p;
                bool same = nh == 4 && (H == 512 || H == 256);
                for (int l = 0; l < nh; ++l) same = same && nb.layers[l].out_p == H && nb.layers[l].out == H;
                // opt-in (FQL_PEC=1): alone the chain drops from 331 to 220 us, but beside the other lanes the update time is
                // unchanged on this box (the lanes barely overlap), so the default stays with plain launches
                // teams: as many as fit one workgroup per CU; each takes B / 16 / teams row tiles
                pec_teams = std::min(B / 16, num_cus / (H / 32));
                while (pec_teams > 1 && (B / 16) % pec_teams) --pec_teams;
                // default: on when every team has >= 4 row tiles to walk per phase (B >= 1024: +5 %); at B = 256 / 512 the chain
                // is shorter but the update is not.  FQL_PEC=1 / 0 forces it on / off.
                const int pec_tiles = pec_teams >= 1 ? B / 16 / pec_teams : 0;
                // opt-in only (FQL_PEC=1): since round 2 the launch-based chain on fql_chain_kernel is faster at every batch size
                // measured (B = 1024: 989 against 970 updates/s), and a persistent kernel that needs co-residency stays off by default
                const bool pec_want = getenv("FQL_PEC") ? atoi(getenv("FQL_PEC")) != 0 : false;
                use_pec = fused_euler && same && pec_want && pec_teams >= 1 && pec_tiles <= PEC_MAX_TILES && cfg.flow_steps <= 20;
                if (use_pec) {
                    // every workgroup of the persistent chain spins on its team mates: all of them must be resident at once
                    const size_t lds = ((size_t)16 * (H + 4) + 1024 + 576 + 512 * (size_t)pec_tiles) * sizeof(float);
                    int per_cu = 0;
                    const hipError_t oe = H == 512 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fql_euler_persistent_kernel<512>, FQL_THREADS, lds)
                                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fql_euler_persistent_kernel<256>, FQL_THREADS, lds);
                    if (oe != hipSuccess || per_cu < 1 || pec_teams * (H / 32) > num_cus) {   // one workgroup per CU: never rely on a second slot
                        (void)hipGetLastError();
                        use_pec = false;
                    }
                }
                if (use_pec) {
                    pec_epoch = (unsigned*)dalloc(W, (size_t)pec_teams + 1);
                    for (int i = 0; i < 2; ++i) pec_g[i] = (fql_u64*)dalloc(W, (size_t)B * H * 2);
                    pec_vg = (fql_u64*)dalloc(W, (size_t)(H / 32) * B * 16 * 2);
                }
            }
        }
        use_xchain = xchain_eligible();
        if (use_xchain) {
            const Net& nb = nets[NET_BC];
            x_lds = (size_t)FQL_XCHAIN_LDS_FLOATS * sizeof(float);
            int per_cu = 0;   // every workgroup waits for its XCD's other 31: all 256 must be resident, one per CU
            const void* kf = (const void*)fql_xchain_kernel;
            hipError_t oe = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x_lds);
            if (oe == hipSuccess) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fql_xchain_kernel, 256, x_lds);
            if (oe != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                use_xchain = false;
            } else {
                xsync = (unsigned*)dalloc(W, 17 * 32);
                xvp = dalloc(W, (size_t)XCH_NMEM * B * 16);
#ifdef FQL_XSTAMPS
                x_stamps = (unsigned long long*)dalloc(W, (size_t)256 * 64 * 2 * 2);
#endif
            }
        }
        if (visual) {
            const size_t ib = (size_t)cfg.img_h * cfg.img_w * cfg.img_c;
            img_all = (unsigned char*)dalloc(W, (2 * (size_t)B * ib + 3) / 4 + 4);
            in_init = (int64_t*)dalloc(W, (size_t)B * 2);
            in_crop = (int*)dalloc(W, (size_t)B * 2); in_crop_user = (int*)dalloc(W, (size_t)B * 2);
            eb_os = make_enc_buf(W, ENC_OS, 2 * B, img_all, true);
            eb_c = make_enc_buf(W, ENC_C, B, img_all, true);
            eb_t = make_enc_buf(W, ENC_T, B, img_all + (size_t)B * ib, false);
            eb_bc = make_enc_buf(W, ENC_BC, B, img_all, true);
        }
        p_os = make_pass(W, NET_OS, 3 * B, X_os, false, false);
        {   // backward view of the (obs, z) block: own gradient buffers, forward buffers shared with p_os
            p_os_bwd = p_os;
            p_os_bwd.M = B;
            const Net& n = nets[NET_OS];
            for (int l = 0; l < n.nl(); ++l) p_os_bwd.dz.push_back(dalloc(W, (size_t)B * n.layers[l].out_p));
            for (int l = 0; l + 1 < n.nl(); ++l) p_os_bwd.dy.push_back(n.layers[l].ln ? dalloc(W, (size_t)B * n.layers[l].out_p) : nullptr);
            if (visual) p_os_bwd.dx0 = dalloc(W, (size_t)B * n.in_p());
        }
        p_bc = make_pass(W, NET_BC, B, X_bc, true, visual);
        p_eu = make_pass(W, NET_BC, B, X_eu, false, false);
        for (int e = 0; e < 2; ++e) {
            p_c1[e] = make_pass(W, NET_C0 + e, B, X_c1, true, visual);
            p_c2[e] = make_pass(W, NET_C0 + e, B, X_c2, true, true);
            p_ct[e] = make_pass(W, NET_T0 + e, B, X_ct, false, false);
        }
        if (!cfg.normalize_q_loss) {
            std::vector<float> h((size_t)B * 16, 0.f);
            for (int b = 0; b < B; ++b) h[(size_t)b * 16] = -1.0f / (2.0f * (float)B);
            for (int e = 0; e < 2; ++e) HIP_CHECK(hipMemcpy(p_c2[e].dz.back(), h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        static const bool trace = getenv("FQL_TRACE") != nullptr;
#define FQL_TR(msg) do { if (trace) fprintf(stderr, "[fql] %s\n", msg); } while (0)
        build_step_program(prog_fwdbwd, true);
        build_opt_program(prog_opt, false);
        build_opt_program(prog_opt_split, true);
        build_step_program(prog_loss, false);
        FQL_TR("programs built");
        schedule(prog_fwdbwd, W); FQL_TR("scheduled fwdbwd"); schedule(prog_opt, W); schedule(prog_opt_split, W); schedule(prog_loss, W);
        FQL_TR("scheduled");
        capture(prog_fwdbwd); FQL_TR("captured fwdbwd"); capture(prog_opt); capture(prog_loss);
        FQL_TR("captured");
        {
            build_full_program(prog_full);
            schedule(prog_full, W);
            FQL_TR("scheduled full");
            capture(prog_full);
            FQL_TR("captured full");
            aql_tried = false;   // (the AQL form of this program is built by the first update that can use it: run_full)
        }
        {
            split_build = true;
            build_step_program(prog_split, true);
            split_build = false;
            schedule(prog_split, W);
            split_ok = capture_split(prog_split);
        }
        launches_per_update = prog_full.exec ? (int64_t)prog_full.launches.size()
                                             : (int64_t)prog_fwdbwd.launches.size() + (int64_t)prog_opt.launches.size();
        src_valid = false;
    }

    // ---------------------------------------------------------------------------------------
    // inputs
    // ---------------------------------------------------------------------------------------
    static bool is_device_ptr(const void* p) {
        hipPointerAttribute_t a;
        hipError_t e = hipPointerGetAttributes(&a, p);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeUnified;
    }
    bool staged_host = false;
    // host memory is borrowed for the call only: drain the staging copies before returning
    void drain_staging(hipStream_t s) {
        if (staged_host) { HIP_CHECK(hipStreamSynchronize(s)); staged_host = false; }
    }
    // HIP is about to write a buffer the update reads: updates still on the AQL queues must have finished, and the next AQL submit must wait for the stream
    void hip_touch() { aql_drain(); hip_dirty = true; }
    const float* stage(const float* p, float* staging, size_t n, hipStream_t s) {
        if (!p) return nullptr;
        if (is_device_ptr(p)) return p;
        hip_touch();
        HIP_CHECK(hipMemcpyAsync(staging, p, n * sizeof(float), hipMemcpyHostToDevice, s));
        staged_host = true;
        return staging;
    }
    void set_source(const SrcDesc& d, hipStream_t s) {
        if (src_valid && std::memcmp(&d, &h_src_shadow, sizeof d) == 0) return;
        SrcDesc* slot = &h_src_ring[src_ring_pos];
        src_ring_pos = (src_ring_pos + 1) % 64;
        *slot = d;
        hip_touch();
        HIP_CHECK(hipMemcpyAsync(d_src, slot, sizeof d, hipMemcpyHostToDevice, s));
        h_src_shadow = d;
        src_valid = true;
    }
    void source_from_batch(const float* obs, const float* act, const float* rew, const float* mask, const float* nobs,
                           int batch, const fql_noise* nz, int advance, hipStream_t s) {
        if (batch != B) invalid("batch_size %d does not match the engine's workspace (%d); call fql_set_batch_size", batch, B);
        if (!obs || !act || !rew || !mask || !nobs) invalid("batch pointers must not be NULL");
        const int od = cfg.obs_dim, ad = cfg.act_dim;
        SrcDesc d{};
        if (visual) {   // uint8 [B, H, W, C] batches into the fixed image buffer the encoder launches read: [obs ; next_obs]
            const size_t ib = (size_t)cfg.img_h * cfg.img_w * cfg.img_c;
            HIP_CHECK(hipMemcpyAsync(img_all, obs, (size_t)B * ib, hipMemcpyDefault, s));
            HIP_CHECK(hipMemcpyAsync(img_all + (size_t)B * ib, nobs, (size_t)B * ib, hipMemcpyDefault, s));
            if (!is_device_ptr(obs) || !is_device_ptr(nobs)) staged_host = true;
        } else {
            d.obs = stage(obs, in_obs, (size_t)B * od, s);
            d.nobs = stage(nobs, in_nobs, (size_t)B * od, s);
        }
        d.act = stage(act, in_act, (size_t)B * ad, s);
        d.rew = stage(rew, in_rew, B, s);
        d.mask = stage(mask, in_mask, B, s);
        fill_noise(d, nz, s);
        d.advance = advance;
        drain_staging(s);
        set_source(d, s);
    }
    void fill_noise(SrcDesc& d, const fql_noise* nz, hipStream_t s) {
        const int ad = cfg.act_dim;
        if (!nz) return;
        d.eps1 = stage(nz->eps1, in_noise[0], (size_t)B * ad, s);
        d.x0 = stage(nz->x0, in_noise[1], (size_t)B * ad, s);
        d.t = stage(nz->t, in_noise[2], B, s);
        d.z = stage(nz->z, in_noise[3], (size_t)B * ad, s);
        d.eps2 = stage(nz->eps2, in_noise[4], (size_t)B * ad, s);
    }
    void source_from_dataset(const int64_t* idx, int batch, int64_t lo, int64_t hi, const fql_noise* nz, hipStream_t s) {
        if (visual) { source_from_frames(idx, nullptr, batch, lo, hi, nz, s); return; }
        if (batch != B) invalid("batch_size %d does not match the engine's workspace (%d)", batch, B);
        if (!ds_obs || ds_size <= 0) throw Invalid{"no dataset uploaded (fql_dataset_upload)"};
        if (lo == 0 && hi == 0) hi = ds_size;
        if (lo < 0 || hi > ds_size || lo >= hi) invalid("bad sampling range [%lld, %lld) for dataset of %lld rows", (long long)lo, (long long)hi, (long long)ds_size);
        SrcDesc d{};
        d.obs = ds_obs; d.act = ds_act; d.rew = ds_rew; d.mask = ds_mask; d.nobs = ds_nobs;
        if (idx) {
            if (is_device_ptr(idx)) d.idx = idx;
            else {
                hip_touch();
                HIP_CHECK(hipMemcpyAsync(in_idx, idx, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, s));
                staged_host = true;
                d.idx = in_idx;
            }
        } else {
            d.use_rng_idx = 1; d.lo = lo; d.span = hi - lo;
        }
        fill_noise(d, nz, s);
        d.advance = 1;
        drain_staging(s);
        set_source(d, s);
    }
    // indices of a balanced batch into one device array: rows [0, split) index the dataset, rows [split, B) the replay ring
    const int64_t* stage_two_idx(const int64_t* a, const int64_t* b, int split, int64_t* dst, hipStream_t s) {
        if (!a && !b) return nullptr;
        if (!a || !b) throw Invalid{"balanced sampling: give both index arrays or neither"};
        hip_touch();
        HIP_CHECK(hipMemcpyAsync(dst, a, (size_t)split * sizeof(int64_t), hipMemcpyDefault, s));
        HIP_CHECK(hipMemcpyAsync(dst + split, b, (size_t)(B - split) * sizeof(int64_t), hipMemcpyDefault, s));
        if (!is_device_ptr(a) || !is_device_ptr(b)) staged_host = true;
        return dst;
    }
    // main.py:255-259: batch = concat(train_dataset.sample(B // 2), replay_buffer.sample(B // 2)), gathered on the device
    void source_balanced(const int64_t* idx_ds, const int64_t* idx_rb, const int32_t* crop, int batch, const fql_noise* nz, hipStream_t s) {
        if (batch != B) invalid("batch_size %d does not match the engine's workspace (%d)", batch, B);
        if (B % 2) invalid("balanced sampling draws batch_size // 2 rows from each source: batch_size %d must be even", B);
        if (ds_size <= 0 || !(visual ? (void*)ds_frames : (void*)ds_obs)) throw Invalid{"balanced sampling: no training dataset uploaded"};
        if (rb_cap <= 0) throw Invalid{"balanced sampling: no replay ring (fql_replay_create)"};
        if (rb_size <= 0) throw Invalid{"balanced sampling: the replay ring is empty (the reference would draw randint(0), a ValueError)"};
        const int split = B / 2;
        if (visual) { source_from_frames(idx_ds, crop, batch, 0, 0, nz, s, split, idx_rb); return; }
        SrcDesc d{};
        d.obs = ds_obs; d.act = ds_act; d.rew = ds_rew; d.mask = ds_mask; d.nobs = ds_nobs;
        d.obs2 = rb_obs; d.act2 = rb_act; d.rew2 = rb_rew; d.mask2 = rb_mask; d.nobs2 = rb_nobs;
        d.split = split;
        d.idx = stage_two_idx(idx_ds, idx_rb, split, in_idx, s);
        if (!d.idx) { d.use_rng_idx = 1; d.lo = 0; d.span = ds_size; d.lo2 = 0; d.span2 = rb_size; }
        fill_noise(d, nz, s);
        d.advance = 1;
        drain_staging(s);
        set_source(d, s);
    }
    // Dataset.sample for image datasets (utils/datasets.py:68-92): indices, frame stacking, random crop -> img_all, on the device
    void source_from_frames(const int64_t* idx, const int32_t* crop, int batch, int64_t lo, int64_t hi, const fql_noise* nz, hipStream_t s,
                            int split = 0, const int64_t* idx_rb = nullptr) {
        if (batch != B) invalid("batch_size %d does not match the engine's workspace (%d)", batch, B);
        if (!visual || !ds_frames || ds_size <= 0) throw Invalid{"no frames dataset uploaded (fql_dataset_upload_frames)"};
        if (lo == 0 && hi == 0) hi = ds_size;
        if (lo < 0 || hi > ds_size || lo >= hi) invalid("bad sampling range [%lld, %lld) for dataset of %lld rows", (long long)lo, (long long)hi, (long long)ds_size);
        const int64_t* idx_dev = nullptr;
        if (split > 0) {
            if (!rb_frames) throw Invalid{"balanced sampling: the replay ring holds no frames"};
            idx_dev = stage_two_idx(idx, idx_rb, split, in_idx + B, s);
        } else if (idx) {
            if (is_device_ptr(idx)) idx_dev = idx;
            else {
                HIP_CHECK(hipMemcpyAsync(in_idx + B, idx, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, s));   // second half of in_idx
                staged_host = true;
                idx_dev = in_idx + B;
            }
        }
        const int* crop_dev = nullptr;
        if (crop) {
            if (is_device_ptr(crop)) crop_dev = crop;
            else {
                HIP_CHECK(hipMemcpyAsync(in_crop_user, crop, (size_t)B * 2 * sizeof(int), hipMemcpyHostToDevice, s));
                staged_host = true;
                crop_dev = in_crop_user;
            }
        }
        ImgIndexArgs ia{idx_dev, crop_dev, ds_init, d_state, seed, lo, hi - lo, ds_paug, B, 3, in_idx, in_init, in_crop,
                        split, rb_init, 0, rb_size};
        hipLaunchKernelGGL(fql_img_index_kernel, dim3((B + FQL_THREADS - 1) / FQL_THREADS), dim3(FQL_THREADS), 0, s, ia);
        const size_t ib = (size_t)cfg.img_h * cfg.img_w * cfg.img_c;
        ImgGatherArgs ga{ds_frames, ds_next_frames, in_idx, in_init, in_crop, img_all, img_all + (size_t)B * ib,
                         B, cfg.img_h, cfg.img_w, cfg.img_c / ds_fs, ds_fs, 3, split, rb_frames, rb_next_frames};
        const size_t tot = ((size_t)B * ib) >> 2;   // one thread per four output bytes (img_w is a multiple of 32: rows are whole dwords)
        hipLaunchKernelGGL(fql_img_gather_kernel, dim3((unsigned)((tot + FQL_THREADS - 1) / FQL_THREADS)), dim3(FQL_THREADS), 0, s, ga);
        HIP_CHECK(hipGetLastError());
        SrcDesc d{};
        d.act = ds_act; d.rew = ds_rew; d.mask = ds_mask;
        d.idx = in_idx;
        if (split > 0) { d.split = split; d.act2 = rb_act; d.rew2 = rb_rew; d.mask2 = rb_mask; }
        fill_noise(d, nz, s);
        d.advance = 1;
        drain_staging(s);
        set_source(d, s);
    }
    void finish_info(float* info, int n, hipStream_t s) {
        if (!info) return;
        hip_touch();
        if (is_device_ptr(info)) {
            HIP_CHECK(hipMemcpyAsync(info, &d_state->info[0], n * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else {
            HIP_CHECK(hipMemcpyAsync(info, &d_state->info[0], n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
    }

    // ---------------------------------------------------------------------------------------
    // eval paths
    // ---------------------------------------------------------------------------------------
    Eval& get_eval(int n_pad) {
        auto it = evals.find(n_pad);
        if (it != evals.end()) return *it->second;
        auto ev = std::make_unique<Eval>();
        ev->n_pad = n_pad;
        auto& W = ev->allocs;
        const int od = cfg.obs_dim, ad = cfg.act_dim;
        ev->X = dalloc(W, (size_t)n_pad * nets[NET_OS].in_p());
        ev->Xf = dalloc(W, (size_t)n_pad * nets[NET_BC].in_p());
        ev->tgt = dalloc(W, (size_t)n_pad * pad16(ad));
        ev->st_obs = dalloc(W, (size_t)n_pad * od);
        ev->st_noise = dalloc(W, (size_t)n_pad * ad);
        ev->st_out = dalloc(W, (size_t)n_pad * ad);
        ev->p_os = make_pass(W, NET_OS, n_pad, ev->X, false, false);
        ev->p_eu = make_pass(W, NET_BC, n_pad, ev->Xf, false, false);
        emit_lane = 0;
        emit_forward(ev->prog_os, ev->p_os, false, GF_CLIP_OUT);
        const int fs = cfg.flow_steps;
        for (int s = 0; s < fs; ++s)
            emit_forward(ev->prog_flow, ev->p_eu, false, GF_EULER | (s == fs - 1 ? GF_EULER_LAST : 0), ev->Xf, ev->tgt,
                         1.0f / (float)fs, (float)(s + 1) / (float)fs);
        schedule(ev->prog_os, W);
        schedule(ev->prog_flow, W);
        if (visual) {
            const size_t ib = (size_t)cfg.img_h * cfg.img_w * cfg.img_c;
            ev->st_img = (unsigned char*)dalloc(W, ((size_t)n_pad * ib + 3) / 4 + 4);
            ev->eb_os = make_enc_buf(W, ENC_OS, n_pad, ev->st_img, false);
            ev->eb_bc = make_enc_buf(W, ENC_BC, n_pad, ev->st_img, false);
            emit_lane = 1;   // throughput kernels (the encoder's Dense has K = flat > 1024)
            emit_encoder_forward(ev->prog_enc_os, ev->eb_os);
            emit_encoder_forward(ev->prog_enc_bc, ev->eb_bc);
            emit_lane = 0;
            schedule(ev->prog_enc_os, W);
            schedule(ev->prog_enc_bc, W);
        }
        Eval& ref = *ev;
        evals[n_pad] = std::move(ev);
        return ref;
    }
    void eval_rows(bool flow, const float* obs, const float* noise, uint64_t sd, int n, float* out, hipStream_t s, bool own_stream = false) {
        if (n <= 0) invalid("n must be positive (got %d)", n);
        if (!obs || !out) invalid("observations/out must not be NULL");
        if (flow && !noise) invalid("noises must not be NULL");
        const int od = cfg.obs_dim, ad = cfg.act_dim;
        const int chunk_cap = visual ? 256 : 4096;
        for (int start = 0; start < n; start += chunk_cap) {
            const int m = std::min(chunk_cap, n - start);
            Eval& ev = get_eval(visual ? (m + 63) & ~63 : pad16(m));
            const float* o;
            if (visual) {   // `obs` is uint8 [n, H, W, C]: encode with the module's encoder, then proceed on the encodings
                const size_t ib = (size_t)cfg.img_h * cfg.img_w * cfg.img_c;
                const unsigned char* src = reinterpret_cast<const unsigned char*>(obs) + (size_t)start * ib;
                HIP_CHECK(hipMemcpyAsync(ev.st_img, src, (size_t)m * ib, hipMemcpyDefault, s));
                if (!is_device_ptr(src)) staged_host = true;
                run_launches(flow ? ev.prog_enc_bc : ev.prog_enc_os, s);
                o = flow ? ev.eb_bc.E : ev.eb_os.E;
            } else {
                o = stage(obs + (size_t)start * od, ev.st_obs, (size_t)m * od, s);
            }
            const float* z = noise ? stage(noise + (size_t)start * ad, ev.st_noise, (size_t)m * ad, s) : nullptr;
            drain_staging(s);
            AssembleArgs a{o, z, flow ? ev.Xf : ev.X, seed, sd + (uint64_t)start, m, ev.n_pad, od, ad,
                           flow ? nets[NET_BC].in_p() : nets[NET_OS].in_p()};
            hipLaunchKernelGGL(fql_assemble_kernel, dim3((ev.n_pad + 3) / 4), dim3(FQL_THREADS), 0, s, a);
            run_launches(flow ? ev.prog_flow : ev.prog_os, s);
            float* dst = out + (size_t)start * ad;
            const bool dev_out = is_device_ptr(dst);
            const float* src16 = flow ? ev.tgt : ev.p_os.out;
            hipLaunchKernelGGL(fql_extract_kernel, dim3((m * ad + 255) / 256), dim3(256), 0, s, src16, dev_out ? dst : ev.st_out, m, ad, pad16(ad));
            HIP_CHECK(hipGetLastError());
            if (!dev_out) {
                HIP_CHECK(hipMemcpyAsync(dst, ev.st_out, (size_t)m * ad * sizeof(float), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
            } else if (own_stream) {
                // device output on the engine's private stream: the caller has no handle to order its reads against, so finish here
                HIP_CHECK(hipStreamSynchronize(s));
            }
        }
    }
};

// ================================================================================================
// C ABI
// ================================================================================================
// FQL_TRY: every entry point that may use HIP on the engine's buffers - updates still on the engine's AQL queues finish first.
// FQL_TRY_FAST: the update entry points that can go to those queues themselves (run_full decides).
#define FQL_TRY(h, ...) FQL_TRY_(h, (h)->hip_touch(), __VA_ARGS__)
#define FQL_TRY_FAST(h, ...) FQL_TRY_(h, (void)0, __VA_ARGS__)
#define FQL_TRY_(h, pre, ...)                              \
    try {                                                  \
        pre;                                               \
        __VA_ARGS__;                                       \
        return FQL_OK;                                     \
    } catch (const Invalid& e) {                           \
        (h)->err = e.msg;                                  \
        return FQL_E_INVALID;                              \
    } catch (const HipError& e) {                          \
        (h)->err = e.msg;                                  \
        return FQL_E_HIP;                                  \
    } catch (const std::exception& e) {                    \
        (h)->err = e.what();                               \
        return FQL_E_HIP;                                  \
    }

static const char* k_info_names[FQL_NUM_INFO] = {
    "critic/critic_loss", "critic/q_mean", "critic/q_max", "critic/q_min", "actor/actor_loss",
    "actor/bc_flow_loss", "actor/distill_loss", "actor/q_loss", "actor/q", "actor/mse",
    "grad/max", "grad/min", "grad/norm"};

// ---- online fine-tuning: ring capacity, uint8 ring insert, the replay ring of balanced sampling ---------------------------------
namespace {
// copy `rows` rows of `w` elements into a fresh zero-filled allocation of `cap` rows; frees the old one
template <typename T>
void regrow(T*& p, int64_t rows, int64_t cap, size_t w) {
    T* q = nullptr;
    HIP_CHECK(hipMalloc((void**)&q, (size_t)cap * w * sizeof(T)));
    HIP_CHECK(hipMemset(q, 0, (size_t)cap * w * sizeof(T)));
    if (p && rows > 0) HIP_CHECK(hipMemcpy(q, p, (size_t)rows * w * sizeof(T), hipMemcpyDeviceToDevice));
    if (p) hipFree(p);
    p = q;
}
}  // namespace

// one transition into row `pos` of a ring: the float fields through the staging row + fql_dataset_add_kernel, frames as two byte copies
static void ring_insert(fql_handle h, float* obs, float* act, float* rew, float* mask, float* nobs, unsigned char* frames,
                        unsigned char* next_frames, int64_t pos, const float* ob, const float* ac, float reward, float m,
                        const float* nob, const uint8_t* frame, const uint8_t* next_frame) {
    const int od = frames ? 0 : h->cfg.obs_dim, ad = h->cfg.act_dim;   // frames rings keep no float observations
    if (od > 1024) invalid("obs_dim too large for the ring-insert kernel");
    if (!h->ds_row) HIP_CHECK(hipMalloc((void**)&h->ds_row, (size_t)(2 * h->cfg.obs_dim + ad + 2) * sizeof(float)));
    std::vector<float> row(2 * od + ad + 2);
    if (od) std::memcpy(row.data(), ob, od * sizeof(float));
    std::memcpy(row.data() + od, ac, ad * sizeof(float));
    row[od + ad] = reward; row[od + ad + 1] = m;
    if (od) std::memcpy(row.data() + od + ad + 2, nob, od * sizeof(float));
    HIP_CHECK(hipMemcpyAsync(h->ds_row, row.data(), row.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(fql_dataset_add_kernel, dim3(1), dim3(std::max(64, pad16(std::max(od, ad)))), 0, h->stream, obs, act, rew, mask, nobs,
                       h->ds_row, pos, od, ad);
    HIP_CHECK(hipGetLastError());
    if (frames) {
        const size_t fb = (size_t)h->cfg.img_h * h->cfg.img_w * (h->cfg.img_c / h->ds_fs);
        HIP_CHECK(hipMemcpyAsync(frames + (size_t)pos * fb, frame, fb, hipMemcpyDefault, h->stream));
        HIP_CHECK(hipMemcpyAsync(next_frames + (size_t)pos * fb, next_frame, fb, hipMemcpyDefault, h->stream));
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));   // the caller's buffers are borrowed for the call only
}

extern "C" {

const char* fql_info_name(int i) { return (i >= 0 && i < FQL_NUM_INFO) ? k_info_names[i] : ""; }
int fql_abi_version(void) { return FQL_ABI_VERSION; }

void fql_default_config(fql_config* c) {
    std::memset(c, 0, sizeof *c);
    c->num_actor_hidden = 4; c->num_value_hidden = 4;
    for (int i = 0; i < 4; ++i) c->actor_hidden[i] = c->value_hidden[i] = 512;
    c->layer_norm = 1; c->actor_layer_norm = 0;
    c->lr = 3e-4f; c->discount = 0.99f; c->tau = 0.005f; c->alpha = 300.0f;
    c->q_agg = 0; c->flow_steps = 10; c->normalize_q_loss = 0; c->batch_size = 256; c->precision = 0;
}

const char* fql_last_error(fql_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fql_create(const fql_config* cfg, uint64_t seed, fql_handle* out) {
    if (!cfg || !out) { g_create_error = "cfg/out must not be NULL"; return FQL_E_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        g_create_error = "no HIP device visible: the FQL engine has no CPU fallback";
        return FQL_E_NODEVICE;
    }
    auto* h = new fql_engine();
    try {
        h->cfg = *cfg;
        h->seed = seed;
        if (cfg->encoder != 0) {   // visual agent (agents/fql.py:196-202): the MLPs see the 512-wide encoding instead of obs
            if (cfg->encoder != 1 && cfg->encoder != 2) invalid("encoder %d not available (0 = none, 1 = impala_small, 2 = impala)", cfg->encoder);
            if (cfg->img_h <= 0 || cfg->img_w <= 0 || cfg->img_h % 32 || cfg->img_w % 32 || cfg->img_w > 128 || cfg->img_c <= 0 || cfg->img_c > 16)
                invalid("image shape must be [H, W, C] with H, W multiples of 32 (W <= 128) and C <= 16 (got %d, %d, %d)", cfg->img_h,
                        cfg->img_w, cfg->img_c);
            {   // the uint8 first layer stages (R + 2) image rows as <= 4 dwords per thread (ConvTile::NU8)
                const int R0 = std::min(cfg->img_h, std::max(1, 128 / cfg->img_w));
                if ((R0 + 2) * cfg->img_w * cfg->img_c > 4096 || (cfg->img_w * cfg->img_c) % 4)
                    invalid("image rows of %d x %d channels are too wide for the first-layer kernel", cfg->img_w, cfg->img_c);
            }
            h->visual = true;
            h->enc_dim = 512;          // mlp_hidden_dims = (512,) (utils/encoders.py:69)
            h->cfg.obs_dim = h->enc_dim;
            cfg = &h->cfg;
        }
        if (cfg->obs_dim <= 0 || cfg->act_dim <= 0) invalid("obs_dim and act_dim must be positive (got %d, %d)", cfg->obs_dim, cfg->act_dim);
        if (cfg->num_actor_hidden < 1 || cfg->num_actor_hidden > FQL_MAX_HIDDEN || cfg->num_value_hidden < 1 || cfg->num_value_hidden > FQL_MAX_HIDDEN)
            invalid("hidden layer counts must be in [1, %d]", FQL_MAX_HIDDEN);
        if (cfg->flow_steps < 1) invalid("flow_steps must be >= 1");
        if (cfg->q_agg != 0 && cfg->q_agg != 1) invalid("q_agg must be 0 (mean) or 1 (min)");
        if (cfg->precision != 0 && cfg->precision != 2) invalid("precision %d not available (0 = fp32 MFMA, 2 = bf16 x 3 split MFMA)", cfg->precision);
        HIP_CHECK(hipGetDevice(&h->device));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, h->device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            g_create_error = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 (MI355X) only";
            delete h;
            return FQL_E_NODEVICE;
        }
        // (stream priorities were tried for the graph lanes in round 1: no gain with two streams, a 4x slowdown with three)
        h->num_cus = prop.multiProcessorCount;
        // (stream priorities and CU masks were tried for the graph lanes in rounds 1-2: no gain with two streams, a 4x slowdown with three)
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream4, hipStreamNonBlocking));
        h->build_nets();
        HIP_CHECK(hipMalloc((void**)&h->P, h->n_total * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&h->G, h->n_train * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&h->Mu, h->n_train * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&h->Nu, h->n_train * sizeof(float)));
        HIP_CHECK(hipMemset(h->G, 0, h->n_train * sizeof(float)));
        HIP_CHECK(hipMemset(h->Mu, 0, h->n_train * sizeof(float)));
        HIP_CHECK(hipMemset(h->Nu, 0, h->n_train * sizeof(float)));
        h->build_enc_weights();
        h->build_leaves();
        h->init_params();
        h->build_chain_weights();
        h->refresh_chain_weights(h->stream);
        HIP_CHECK(hipMalloc((void**)&h->d_state, sizeof(DevState)));
        HIP_CHECK(hipMalloc((void**)&h->d_src, sizeof(SrcDesc)));
        HIP_CHECK(hipHostMalloc((void**)&h->h_src_ring, 64 * sizeof(SrcDesc), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&h->h_info_ring, fql_engine::kInfoRing * 16 * sizeof(float), hipHostMallocDefault));
        for (auto& e : h->info_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DevState st{};
        st.rng_step = 0; st.adam_count = 0; st.train_step = 1; st.b1pow = 1.0; st.b2pow = 1.0; st.grad_scale = 1.0f; st.lam = 1.0f;
        const float ninf = -INFINITY, pinf = INFINITY;
        int bi; std::memcpy(&bi, &ninf, 4); st.gmax = bi >= 0 ? bi : bi ^ 0x7FFFFFFF;
        std::memcpy(&bi, &pinf, 4); st.gmin = bi >= 0 ? bi : bi ^ 0x7FFFFFFF;
        HIP_CHECK(hipMemcpy(h->d_state, &st, sizeof st, hipMemcpyHostToDevice));
        h->build_workspace(cfg->batch_size);
    } catch (const Invalid& e) {
        g_create_error = e.msg; fql_destroy(h); return FQL_E_INVALID;
    } catch (const HipError& e) {
        g_create_error = e.msg; fql_destroy(h); return FQL_E_HIP;
    }
    *out = h;
    return FQL_OK;
}

int fql_destroy(fql_handle h) {
    if (!h) return FQL_OK;
    h->stop_workers();
    try { h->aql_drain(); } catch (...) {}
    if (h->stream) hipStreamSynchronize(h->stream);
    hipDeviceSynchronize();
    h->free_workspace();
    h->aql.shutdown();
    for (auto& kv : h->evals) {
        for (void* p : kv.second->allocs) hipFree(p);
    }
    hipFree(h->P); hipFree(h->G); hipFree(h->Mu); hipFree(h->Nu); hipFree(h->d_chunks); hipFree(h->d_partials); hipFree(h->d_leaf_range); hipFree(h->d_state); hipFree(h->d_src);
    if (h->h_src_ring) hipHostFree(h->h_src_ring);
    if (h->h_info_ring) hipHostFree(h->h_info_ring);
    for (auto& e : h->info_ev) if (e) hipEventDestroy(e);
    hipFree(h->ds_obs); hipFree(h->ds_act); hipFree(h->ds_rew); hipFree(h->ds_mask); hipFree(h->ds_nobs); hipFree(h->ds_row);
    hipFree(h->ds_frames); hipFree(h->ds_next_frames); hipFree(h->ds_init);
    hipFree(h->rb_obs); hipFree(h->rb_act); hipFree(h->rb_rew); hipFree(h->rb_mask); hipFree(h->rb_nobs);
    hipFree(h->rb_frames); hipFree(h->rb_next_frames); hipFree(h->rb_init);
    for (void* q : h->enc_allocs) hipFree(q);
    for (void* q : h->chain_allocs) hipFree(q);
    if (h->stream) hipStreamDestroy(h->stream);
    if (h->stream2) hipStreamDestroy(h->stream2);
    if (h->stream3) hipStreamDestroy(h->stream3);
    if (h->stream4) hipStreamDestroy(h->stream4);
    delete h;
    return FQL_OK;
}

int fql_set_batch_size(fql_handle h, int batch_size) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, { if (batch_size != h->B) h->build_workspace(batch_size); });
}

int fql_num_leaves(fql_handle h) { return h ? (int)h->leaves.size() : FQL_E_INVALID; }

int fql_leaf_info(fql_handle h, int index, char* name, int name_cap, int* ndim, int64_t shape[4]) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (index < 0 || index >= (int)h->leaves.size()) invalid("leaf index %d out of range", index);
        const Leaf& l = h->leaves[index];
        if (name && name_cap > 0) { std::strncpy(name, l.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
        if (ndim) *ndim = l.ndim;
        if (shape) for (int i = 0; i < 4; ++i) shape[i] = l.shape[i];
    });
}

static int leaf_rw(fql_handle h, const char* leaf, float* host, size_t n, int which, bool to_device) {
    if (!h) return FQL_E_INVALID;
    try {
        if (!leaf || !host) invalid("leaf/host pointer must not be NULL");
        const Leaf* l = h->find_leaf(leaf);
        if (!l) { h->err = std::string("unknown leaf: ") + leaf; return FQL_E_NOTFOUND; }
        if (n != fql_engine::leaf_count(*l)) invalid("leaf %s has %zu elements, got %zu", leaf, fql_engine::leaf_count(*l), n);
        float* arena = h->P;
        if (which >= 0) {
            if (!l->trainable && which >= 0) {
                // optax keeps (zero-gradient) moments for the target leaves too; they stay exactly 0
                if (to_device) return FQL_OK;
                std::memset(host, 0, n * sizeof(float));
                return FQL_OK;
            }
            arena = which == 0 ? h->Mu : h->Nu;
        }
        h->leaf_io(*l, arena, host, to_device);
        if (to_device && which < 0 && l->name.rfind("modules_actor_bc_flow/", 0) == 0) h->refresh_chain_weights(h->stream);
        return FQL_OK;
    } catch (const Invalid& e) { h->err = e.msg; return FQL_E_INVALID; }
    catch (const HipError& e) { h->err = e.msg; return FQL_E_HIP; }
}
int fql_get_param(fql_handle h, const char* leaf, float* out, size_t n) { return leaf_rw(h, leaf, out, n, -1, false); }
int fql_set_param(fql_handle h, const char* leaf, const float* in, size_t n) { return leaf_rw(h, leaf, const_cast<float*>(in), n, -1, true); }
int fql_get_opt_state(fql_handle h, int which, const char* leaf, float* out, size_t n) {
    if (which != 0 && which != 1) return FQL_E_INVALID;
    return leaf_rw(h, leaf, out, n, which, false);
}
int fql_set_opt_state(fql_handle h, int which, const char* leaf, const float* in, size_t n) {
    if (which != 0 && which != 1) return FQL_E_INVALID;
    return leaf_rw(h, leaf, const_cast<float*>(in), n, which, true);
}

int fql_get_step(fql_handle h, int64_t* adam_count, int64_t* train_step) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        HIP_CHECK(hipStreamSynchronize(h->stream));
        HIP_CHECK(hipDeviceSynchronize());
        DevState st;
        HIP_CHECK(hipMemcpy(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost));
        if (adam_count) *adam_count = st.adam_count;
        if (train_step) *train_step = st.train_step;
    });
}
int fql_set_step(fql_handle h, int64_t adam_count, int64_t train_step) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (adam_count < 0) invalid("adam_count must be >= 0");
        HIP_CHECK(hipDeviceSynchronize());
        DevState st;
        HIP_CHECK(hipMemcpy(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost));
        st.adam_count = adam_count; st.train_step = train_step;
        st.b1pow = std::pow(0.9, (double)adam_count); st.b2pow = std::pow(0.999, (double)adam_count);
        HIP_CHECK(hipMemcpy(h->d_state, &st, sizeof st, hipMemcpyHostToDevice));
    });
}

// NULL = the engine's own stream; FQL_STREAM_LEGACY ((void*)1 == hipStreamLegacy) = the legacy default stream, i.e. the stream
// torch's *default* stream maps to (its handle reads 0, which cannot be told from "no stream" here); anything else is a hipStream_t.
static hipStream_t pick(fql_handle h, void* s) { return s ? (hipStream_t)s : h->stream; }

static void run_program(fql_handle h, Program& pr, hipStream_t s) {
    // FQL_NO_GRAPH (diagnostics; the captured graph is the product path): 1 = eager launches on one stream, 2 = eager launches on the lane
    // streams, 3 = one host thread per lane (run_threaded; tests/test_gpu_streams.py runs the parity suite under it)
    static const int no_graph = getenv("FQL_NO_GRAPH") ? atoi(getenv("FQL_NO_GRAPH")) : 0;
    if (no_graph == 3) h->run_threaded(pr, s);
    else if (no_graph == 2) h->run_launches(pr, s, true);
    else if (no_graph) h->run_launches(pr, s);
    else HIP_CHECK(hipGraphLaunch(pr.exec, s));
}

// The whole update (prog_full).  The caller left the stream to the engine: pre-built AQL packets on the engine's own queues (fql_aql.h; the
// stream is drained first if HIP work may be pending on it).  Otherwise, or when that path is off: the captured graph on `s`.
static void run_full(fql_handle h, hipStream_t s, bool own_stream) {
    static const int no_graph = getenv("FQL_NO_GRAPH") ? atoi(getenv("FQL_NO_GRAPH")) : 0;
    if (own_stream && !no_graph && h->prog_full.exec && !h->aql_tried) {
        h->aql_tried = true;
        h->hip_touch();
        h->build_aql(h->prog_full, h->aql_full);
    }
    if (own_stream && h->aql_full.ok && !no_graph && h->prog_full.exec) {
        if (h->hip_dirty) { HIP_CHECK(hipStreamSynchronize(s)); h->hip_dirty = false; }
        try { h->aql.submit(h->aql_full); } catch (const AqlError& e) { throw HipError{e.msg}; }
        h->last_update_aql = true;
        return;
    }
    h->last_update_aql = false;
    h->hip_touch();
    if (h->prog_full.exec) run_program(h, h->prog_full, s);
    else { run_program(h, h->prog_fwdbwd, s); run_program(h, h->prog_opt, s); }
}

int fql_update_begin(fql_handle h, const float* obs, const float* act, const float* rew, const float* mask,
                     const float* nobs, int batch_size, const fql_noise* noise, void* stream) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        hipStream_t s = pick(h, stream);
        h->source_from_batch(obs, act, rew, mask, nobs, batch_size, noise, 1, s);
        run_program(h, h->prog_fwdbwd, s);
        h->began = true;
        h->split_begun = false;
    });
}
int fql_update_end(fql_handle h, float* info13, void* stream) {
    if (!h) return FQL_E_INVALID;
    if (!h->began) { h->err = "fql_update_end without fql_update_begin"; return FQL_E_STATE; }
    FQL_TRY(h, {
        hipStream_t s = pick(h, stream);
        run_program(h, h->prog_opt, s);
        h->began = false;
        h->finish_info(info13, FQL_NUM_INFO, s);
    });
}
int fql_update(fql_handle h, const float* obs, const float* act, const float* rew, const float* mask, const float* nobs,
               int batch_size, const fql_noise* noise, float* info13, void* stream) {
    if (h && h->prog_full.exec && !h->began) {
        FQL_TRY_FAST(h, {
            hipStream_t s = pick(h, stream);
            h->source_from_batch(obs, act, rew, mask, nobs, batch_size, noise, 1, s);
            run_full(h, s, stream == nullptr);
            h->finish_info(info13, FQL_NUM_INFO, s);
        });
    }
    int rc = fql_update_begin(h, obs, act, rew, mask, nobs, batch_size, noise, stream);
    if (rc != FQL_OK) return rc;
    return fql_update_end(h, info13, stream);
}
int fql_grad_buffer(fql_handle h, void** device_ptr, size_t* num_floats) {
    if (!h || !device_ptr || !num_floats) return FQL_E_INVALID;
    *device_ptr = h->G;
    *num_floats = h->n_train;
    return FQL_OK;
}
int fql_set_grad_scale(fql_handle h, float scale) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(&h->d_state->grad_scale, &scale, sizeof scale, hipMemcpyHostToDevice));
    });
}

int fql_set_rng_stream(fql_handle h, uint64_t stream_id) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(&h->d_state->rng_stream, &stream_id, sizeof stream_id, hipMemcpyHostToDevice));
    });
}

int fql_total_loss(fql_handle h, const float* obs, const float* act, const float* rew, const float* mask, const float* nobs,
                   int batch_size, const fql_noise* noise, float* loss, float* info10, void* stream) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        hipStream_t s = pick(h, stream);
        h->source_from_batch(obs, act, rew, mask, nobs, batch_size, noise, 0, s);
        run_program(h, h->prog_loss, s);
        float tmp[10];
        HIP_CHECK(hipMemcpyAsync(tmp, &h->d_state->info[0], sizeof tmp, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (loss) *loss = tmp[0] + tmp[4];
        if (info10) std::memcpy(info10, tmp, sizeof tmp);
    });
}

int fql_sample_actions(fql_handle h, const float* obs, int n, const float* noise, uint64_t seed, float* out, void* stream) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, h->eval_rows(false, obs, noise, seed, n, out, pick(h, stream), stream == nullptr));
}
int fql_flow_actions(fql_handle h, const float* obs, const float* noises, int n, float* out, void* stream) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, h->eval_rows(true, obs, noises, 0, n, out, pick(h, stream), stream == nullptr));
}

int fql_dataset_upload(fql_handle h, int64_t n, int64_t capacity, const float* obs, const float* act, const float* rew,
                       const float* mask, const float* nobs) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (n < 0 || capacity < std::max<int64_t>(n, 1)) invalid("need capacity >= max(n, 1) (n=%lld capacity=%lld)", (long long)n, (long long)capacity);
        if (n > 0 && (!obs || !act || !rew || !mask || !nobs)) invalid("dataset pointers must not be NULL");
        HIP_CHECK(hipDeviceSynchronize());
        hipFree(h->ds_obs); hipFree(h->ds_act); hipFree(h->ds_rew); hipFree(h->ds_mask); hipFree(h->ds_nobs);
        h->ds_obs = h->ds_act = h->ds_rew = h->ds_mask = h->ds_nobs = nullptr;
        const int od = h->cfg.obs_dim, ad = h->cfg.act_dim;
        auto up = [&](float*& dst, const float* src, size_t w) {
            HIP_CHECK(hipMalloc((void**)&dst, (size_t)capacity * w * sizeof(float)));
            HIP_CHECK(hipMemset(dst, 0, (size_t)capacity * w * sizeof(float)));
            if (n > 0) HIP_CHECK(hipMemcpy(dst, src, (size_t)n * w * sizeof(float), hipMemcpyDefault));
        };
        up(h->ds_obs, obs, od); up(h->ds_act, act, ad); up(h->ds_rew, rew, 1); up(h->ds_mask, mask, 1); up(h->ds_nobs, nobs, od);
        if (!h->ds_row) HIP_CHECK(hipMalloc((void**)&h->ds_row, (size_t)(2 * od + ad + 2) * sizeof(float)));
        h->ds_size = n; h->ds_cap = capacity; h->ds_ptr = n % capacity;
        h->src_valid = false;
    });
}
int fql_dataset_add(fql_handle h, const float* ob, const float* ac, float reward, float mask, const float* nob) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (!h->ds_obs) throw Invalid{"no dataset allocated (fql_dataset_upload)"};
        if (!ob || !ac || !nob) invalid("transition pointers must not be NULL");
        const int od = h->cfg.obs_dim, ad = h->cfg.act_dim;
        if (od > 1024) invalid("obs_dim too large for the ring-insert kernel");
        std::vector<float> row(2 * od + ad + 2);
        std::memcpy(row.data(), ob, od * sizeof(float));
        std::memcpy(row.data() + od, ac, ad * sizeof(float));
        row[od + ad] = reward; row[od + ad + 1] = mask;
        std::memcpy(row.data() + od + ad + 2, nob, od * sizeof(float));
        HIP_CHECK(hipMemcpyAsync(h->ds_row, row.data(), row.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(fql_dataset_add_kernel, dim3(1), dim3(std::max(64, pad16(std::max(od, ad)))), 0, h->stream, h->ds_obs,
                           h->ds_act, h->ds_rew, h->ds_mask, h->ds_nobs, h->ds_row, h->ds_ptr, od, ad);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(h->stream));
        // utils/datasets.py:489-491
        h->ds_ptr = (h->ds_ptr + 1) % h->ds_cap;
        h->ds_size = std::max(h->ds_ptr, h->ds_size);
    });
}
int fql_dataset_size(fql_handle h, int64_t* size, int64_t* pointer) {
    if (!h) return FQL_E_INVALID;
    if (size) *size = h->ds_size;
    if (pointer) *pointer = h->ds_ptr;
    return FQL_OK;
}
// ---- online fine-tuning: ring capacity, uint8 ring insert, the replay ring of balanced sampling (helpers above extern "C")
int fql_noise_from_jax_keys(fql_handle h, const uint32_t* keys10, int partitionable, int batch_size, fql_noise* out, void* stream) {
    if (!h || !keys10 || !out) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (batch_size != h->B) invalid("batch_size %d does not match the engine's workspace (%d); call fql_set_batch_size", batch_size, h->B);
        hipStream_t s = pick(h, stream);
        const int B = h->B, ad = h->cfg.act_dim;
        JaxNoiseArgs a{};
        for (int w = 0; w < 5; ++w) {
            a.key[w][0] = keys10[2 * w]; a.key[w][1] = keys10[2 * w + 1];
            a.out[w] = h->in_noise[w];
            a.n[w] = w == 2 ? B : B * ad;
        }
        a.partitionable = partitionable ? 1 : 0;
        hipLaunchKernelGGL(fql_jax_noise_kernel, dim3((B * ad + FQL_THREADS - 1) / FQL_THREADS, 5), dim3(FQL_THREADS), 0, s, a);
        HIP_CHECK(hipGetLastError());
        out->eps1 = h->in_noise[0]; out->x0 = h->in_noise[1]; out->t = h->in_noise[2]; out->z = h->in_noise[3]; out->eps2 = h->in_noise[4];
    });
}
int fql_dataset_reserve(fql_handle h, int64_t capacity) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        const bool fr = h->visual;
        if (!(fr ? (void*)h->ds_frames : (void*)h->ds_obs)) throw Invalid{"fql_dataset_reserve: no dataset uploaded"};
        if (capacity < h->ds_cap) invalid("fql_dataset_reserve: capacity %lld below the current %lld rows", (long long)capacity, (long long)h->ds_cap);
        if (capacity == h->ds_cap) return FQL_OK;
        HIP_CHECK(hipDeviceSynchronize());
        const int od = h->cfg.obs_dim, ad = h->cfg.act_dim;
        const int64_t rows = h->ds_cap;
        regrow(h->ds_act, rows, capacity, ad); regrow(h->ds_rew, rows, capacity, 1); regrow(h->ds_mask, rows, capacity, 1);
        if (fr) {
            const size_t fb = (size_t)h->cfg.img_h * h->cfg.img_w * (h->cfg.img_c / h->ds_fs);
            regrow(h->ds_frames, rows, capacity, fb); regrow(h->ds_next_frames, rows, capacity, fb);
            // Dataset.__init__ computes initial_locs once (utils/datasets.py:58-62); rows inserted later fall behind the LAST initial
            // location of the initial data (searchsorted(..., side='right') - 1, utils/datasets.py:75) - kept as the reference has it
            std::vector<int64_t> init((size_t)capacity);
            HIP_CHECK(hipMemcpy(init.data(), h->ds_init, (size_t)rows * sizeof(int64_t), hipMemcpyDeviceToHost));
            int64_t last = 0;
            for (int64_t i = 0; i < rows; ++i) last = std::max(last, init[i]);
            for (int64_t i = rows; i < capacity; ++i) init[i] = last;
            hipFree(h->ds_init); h->ds_init = nullptr;
            HIP_CHECK(hipMalloc((void**)&h->ds_init, (size_t)capacity * sizeof(int64_t)));
            HIP_CHECK(hipMemcpy(h->ds_init, init.data(), (size_t)capacity * sizeof(int64_t), hipMemcpyHostToDevice));
        } else {
            regrow(h->ds_obs, rows, capacity, od); regrow(h->ds_nobs, rows, capacity, od);
        }
        h->ds_cap = capacity;
        h->ds_ptr = h->ds_size % capacity;
        h->src_valid = false;
    });
}
int fql_dataset_add_frames(fql_handle h, const uint8_t* frame, const uint8_t* next_frame, const float* ac, float reward, float mask) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (!h->visual || !h->ds_frames) throw Invalid{"no frames dataset allocated (fql_dataset_upload_frames)"};
        if (!frame || !next_frame || !ac) invalid("transition pointers must not be NULL");
        ring_insert(h, nullptr, h->ds_act, h->ds_rew, h->ds_mask, nullptr, h->ds_frames, h->ds_next_frames, h->ds_ptr, nullptr, ac, reward, mask,
                    nullptr, frame, next_frame);
        h->ds_ptr = (h->ds_ptr + 1) % h->ds_cap;          // utils/datasets.py:489-491
        h->ds_size = std::max(h->ds_ptr, h->ds_size);
    });
}
int fql_replay_create(fql_handle h, int64_t capacity) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (capacity < 1) invalid("fql_replay_create: capacity %lld", (long long)capacity);
        if (h->visual && !h->ds_frames) throw Invalid{"fql_replay_create: upload the frames dataset first (frame_stack and p_aug are shared, main.py:117-120)"};
        HIP_CHECK(hipDeviceSynchronize());
        const int od = h->cfg.obs_dim, ad = h->cfg.act_dim;
        regrow(h->rb_act, 0, capacity, ad); regrow(h->rb_rew, 0, capacity, 1); regrow(h->rb_mask, 0, capacity, 1);
        if (h->visual) {
            const size_t fb = (size_t)h->cfg.img_h * h->cfg.img_w * (h->cfg.img_c / h->ds_fs);
            regrow(h->rb_frames, 0, capacity, fb); regrow(h->rb_next_frames, 0, capacity, fb);
            regrow(h->rb_init, 0, capacity, 1);   // ReplayBuffer.create: terminals all zero => initial_locs = [0] (utils/datasets.py:58-62)
        } else {
            regrow(h->rb_obs, 0, capacity, od); regrow(h->rb_nobs, 0, capacity, od);
        }
        h->rb_cap = capacity; h->rb_size = 0; h->rb_ptr = 0;
        h->src_valid = false;
    });
}
int fql_replay_add(fql_handle h, const float* ob, const float* ac, float reward, float mask, const float* nob) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (h->rb_cap <= 0 || h->visual) throw Invalid{"fql_replay_add: no state replay ring (fql_replay_create; frames agents use fql_replay_add_frames)"};
        if (!ob || !ac || !nob) invalid("transition pointers must not be NULL");
        ring_insert(h, h->rb_obs, h->rb_act, h->rb_rew, h->rb_mask, h->rb_nobs, nullptr, nullptr, h->rb_ptr, ob, ac, reward, mask, nob, nullptr, nullptr);
        h->rb_ptr = (h->rb_ptr + 1) % h->rb_cap;
        h->rb_size = std::max(h->rb_ptr, h->rb_size);
    });
}
int fql_replay_add_frames(fql_handle h, const uint8_t* frame, const uint8_t* next_frame, const float* ac, float reward, float mask) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (h->rb_cap <= 0 || !h->visual) throw Invalid{"fql_replay_add_frames: no frames replay ring (fql_replay_create on a frames agent)"};
        if (!frame || !next_frame || !ac) invalid("transition pointers must not be NULL");
        ring_insert(h, nullptr, h->rb_act, h->rb_rew, h->rb_mask, nullptr, h->rb_frames, h->rb_next_frames, h->rb_ptr, nullptr, ac, reward, mask,
                    nullptr, frame, next_frame);
        h->rb_ptr = (h->rb_ptr + 1) % h->rb_cap;
        h->rb_size = std::max(h->rb_ptr, h->rb_size);
    });
}
int fql_replay_size(fql_handle h, int64_t* size, int64_t* pointer) {
    if (!h) return FQL_E_INVALID;
    if (size) *size = h->rb_size;
    if (pointer) *pointer = h->rb_ptr;
    return FQL_OK;
}
int fql_update_balanced(fql_handle h, const int64_t* idx_dataset, const int64_t* idx_replay, const int32_t* crop_froms, int batch_size,
                        const fql_noise* noise, float* info13, void* stream) {
    if (!h) return FQL_E_INVALID;
    if (h->began) { h->err = "fql_update_balanced between fql_update_begin and fql_update_end"; return FQL_E_STATE; }
    FQL_TRY_FAST(h, {
        hipStream_t s = pick(h, stream);
        if (h->visual) h->hip_touch();
        h->source_balanced(idx_dataset, idx_replay, crop_froms, batch_size, noise, s);
        run_full(h, s, stream == nullptr);
        h->finish_info(info13, FQL_NUM_INFO, s);
    });
}

int fql_update_from_dataset_begin(fql_handle h, const int64_t* idx, int batch_size, int64_t lo, int64_t hi,
                                  const fql_noise* noise, void* stream) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        hipStream_t s = pick(h, stream);
        h->source_from_dataset(idx, batch_size, lo, hi, noise, s);
        run_program(h, h->prog_fwdbwd, s);
        h->began = true;
        h->split_begun = false;
    });
}
int fql_update_from_dataset_begin_split(fql_handle h, const int64_t* idx, int batch_size, int64_t lo, int64_t hi,
                                        const fql_noise* noise, void* stream0, void* stream1) {
    if (!h) return FQL_E_INVALID;
    if (!h->split_ok || !stream0 || !stream1 || stream0 == stream1) { h->err = "split update unavailable (needs two distinct streams and a two-lane program)"; return FQL_E_STATE; }
    FQL_TRY(h, {
        h->source_from_dataset(idx, batch_size, lo, hi, noise, (hipStream_t)stream0);
        h->launch_split((hipStream_t)stream0, (hipStream_t)stream1);
        h->began = true;
    });
}
int fql_update_begin_split(fql_handle h, const float* obs, const float* act, const float* rew, const float* mask, const float* nobs,
                           int batch_size, const fql_noise* noise, void* stream0, void* stream1) {
    if (!h) return FQL_E_INVALID;
    if (!h->split_ok || !stream0 || !stream1 || stream0 == stream1) { h->err = "split update unavailable (needs two distinct streams and a two-lane program)"; return FQL_E_STATE; }
    FQL_TRY(h, {
        h->source_from_batch(obs, act, rew, mask, nobs, batch_size, noise, 1, (hipStream_t)stream0);
        h->launch_split((hipStream_t)stream0, (hipStream_t)stream1);
        h->began = true;
    });
}
int fql_update_end_split(fql_handle h, float* info13, void* stream0, void* stream1) {
    if (!h) return FQL_E_INVALID;
    if (!h->began) { h->err = "fql_update_end_split without fql_update_begin"; return FQL_E_STATE; }
    if (!stream0 || !stream1 || stream0 == stream1) { h->err = "fql_update_end_split needs two distinct streams"; return FQL_E_STATE; }
    FQL_TRY(h, {
        h->run_opt_split((hipStream_t)stream0, (hipStream_t)stream1);
        h->began = false;
        h->finish_info(info13, FQL_NUM_INFO, (hipStream_t)stream0);
    });
}
int fql_grad_buckets(fql_handle h, size_t offsets[2], size_t lengths[2]) {
    if (!h || !offsets || !lengths) return FQL_E_INVALID;
    // bucket 0: both critic members + the BC flow actor (all produced on lane 1); bucket 1: the one-step actor (lane 0's tail)
    offsets[0] = 0; lengths[0] = h->nets[NET_OS].off;
    offsets[1] = h->nets[NET_OS].off; lengths[1] = h->n_train - h->nets[NET_OS].off;
    return h->split_ok ? FQL_OK : FQL_E_STATE;
}
int fql_update_from_dataset(fql_handle h, const int64_t* idx, int batch_size, int64_t lo, int64_t hi, const fql_noise* noise,
                            float* info13, void* stream) {
    if (h && h->visual) return fql_update_from_frames(h, idx, nullptr, batch_size, lo, hi, noise, info13, stream);
    if (h && h->prog_full.exec && !h->began) {
        FQL_TRY_FAST(h, {
            hipStream_t s = pick(h, stream);
            h->source_from_dataset(idx, batch_size, lo, hi, noise, s);
            run_full(h, s, stream == nullptr);
            h->finish_info(info13, FQL_NUM_INFO, s);
        });
    }
    int rc = fql_update_from_dataset_begin(h, idx, batch_size, lo, hi, noise, stream);
    if (rc != FQL_OK) return rc;
    return fql_update_end(h, info13, stream);
}

int fql_info_enqueue(fql_handle h, void* stream, uint64_t* ticket) {
    if (!h || !ticket) return FQL_E_INVALID;
    FQL_TRY(h, {
        hipStream_t s = pick(h, stream);
        const uint64_t k = h->info_seq++;
        const int slot = (int)(k % fql_engine::kInfoRing);
        HIP_CHECK(hipMemcpyAsync(h->h_info_ring[slot], &h->d_state->info[0], FQL_NUM_INFO * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipEventRecord(h->info_ev[slot], s));
        *ticket = k;
    });
}
int fql_info_wait(fql_handle h, uint64_t ticket, float* info13_host) {
    if (!h || !info13_host) return FQL_E_INVALID;
    if (ticket >= h->info_seq) { h->err = "unknown info ticket"; return FQL_E_INVALID; }
    if (h->info_seq - ticket > (uint64_t)fql_engine::kInfoRing) { h->err = "info ticket expired (more than 64 later tickets were taken before it was read)"; return FQL_E_STATE; }
    FQL_TRY(h, {
        const int slot = (int)(ticket % fql_engine::kInfoRing);
        HIP_CHECK(hipEventSynchronize(h->info_ev[slot]));
        std::memcpy(info13_host, h->h_info_ring[slot], FQL_NUM_INFO * sizeof(float));
        if (h->use_pec) {   // (opt-in persistent chain) a hand-off time-out makes the optimizer skip the update: say so at the first read
            unsigned e = 0;
            HIP_CHECK(hipMemcpy(&e, h->pec_epoch + h->pec_teams, sizeof e, hipMemcpyDeviceToHost));
            if (e) throw HipError{"persistent Euler chain: a team hand-off timed out (that update and every later one were not applied)"};
        }
    });
}

int fql_synchronize(fql_handle h, int* mode) {
    if (!h) return FQL_E_INVALID;
    if (mode) *mode = h->last_update_aql ? 1 : 0;
    FQL_TRY(h, {
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->hip_dirty = false;
    });
}

int fql_read_info(fql_handle h, float* info13_host) {
    if (!h || !info13_host) return FQL_E_INVALID;
    FQL_TRY(h, {
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(info13_host, &h->d_state->info[0], FQL_NUM_INFO * sizeof(float), hipMemcpyDeviceToHost));
        if (h->use_pec) {
            unsigned e = 0;
            HIP_CHECK(hipMemcpy(&e, h->pec_epoch + h->pec_teams, sizeof e, hipMemcpyDeviceToHost));
            if (e) throw HipError{"persistent Euler chain: a team hand-off timed out (results of this update are invalid)"};
        }
    });
}

int fql_stats(fql_handle h, int64_t* launches_per_update, int64_t* macs_per_update, int64_t* param_count) {
    if (!h) return FQL_E_INVALID;
    if (launches_per_update) *launches_per_update = h->launches_per_update;
    if (macs_per_update) *macs_per_update = h->macs_per_update();
    if (param_count) {
        int64_t n = 0;
        for (const Leaf& l : h->leaves) n += (int64_t)fql_engine::leaf_count(l);
        *param_count = n;
    }
    return FQL_OK;
}
void* fql_stream(fql_handle h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"

// Per-launch device times of ONE update, measured with HIP events on the engine's stream (used by bench.py for the per-kernel
// roofline: rocprofv3 serialises the graph's lanes, and so does this pass - the launches run eagerly, in program order, on one
// stream, an event before and after each).  It IS a real update (state advances).  Needs a device-resident dataset.
// Out, per launch (up to cap): op type (OpType), lane, workgroups, microseconds, algorithmic MACs of its GEMM-shaped tasks.
// An event-to-event interval holds the launch gap in front of the kernel besides the kernel: *null_us receives the same interval for a
// one-workgroup kernel that returns at once (median of 32), which the caller subtracts (minus the ~1.4 us such a kernel itself shows
// in a rocprofv3 trace) to compare with rocprofv3's per-kernel durations.
extern "C" int fql_profile_update(fql_handle h, int batch_size, int cap, int* type, int* lane, int* grid, float* us, double* macs, float* null_us) {
    if (!h || !type || !us) return FQL_E_INVALID;
    if (!h->prog_full.exec) { h->err = "no single-graph update program"; return FQL_E_STATE; }
    try {
        hipStream_t s = h->stream;
        if (null_us) *null_us = 0.f;   // (no calibration needed: the events ride on the dispatches)
        h->source_from_dataset(nullptr, batch_size, 0, 0, nullptr, s);
        Program& pr = h->prog_full;
        const int n = (int)pr.launches.size();
        std::vector<hipEvent_t> ev(2 * n);
        for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
        for (int i = 0; i < n; ++i) {   // the update's launches in program order on one stream, each dispatch carrying its own event pair
            h->prof_a = ev[2 * i]; h->prof_b = ev[2 * i + 1];
            h->issue(pr.launches[i], s, -1);
        }
        h->prof_a = h->prof_b = nullptr;
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s));
        for (int i = 0; i < n && i < cap; ++i) {
            const Launch& L = pr.launches[i];
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            type[i] = (int)L.type; if (lane) lane[i] = L.lane; if (grid) grid[i] = L.grid; us[i] = ms * 1e3f;
            if (macs) macs[i] = L.macs;
        }
        for (auto& e : ev) hipEventDestroy(e);
        return std::min(n, cap);
    } catch (const Invalid& e) { h->prof_a = h->prof_b = nullptr; h->err = e.msg; return FQL_E_INVALID; }
    catch (const HipError& e) { h->prof_a = h->prof_b = nullptr; h->err = e.msg; return FQL_E_HIP; }
}

// Diagnostic only (not in include/fql_amd.h): copy a workspace buffer of the last update to the host, so tests can look at what the
// production RNG / gather path actually produced.  which: 0 X_os [3B, inp_c], 1 X_bc [B, inp_b], 2 vel [B, ap], 3 w_act [B, ap],
// 4 X_c1 [B, inp_c], 5 w_rew [B], 6 w_mask [B], 7 in_idx (int64 [B], frames path), 8 in_crop (int32 [B][2], frames path),
// 9 X_eu [B, inp_b].  *dims receives {rows, row stride in elements}.
extern "C" int fql_debug_workspace(fql_handle h, int which, void* out, size_t bytes, int* dims) {
    if (!h || !out) return FQL_E_INVALID;
    const int B = h->B, ic = h->nets[NET_OS].in_p(), ib = h->nets[NET_BC].in_p(), ap = pad16(h->cfg.act_dim);
    const void* src = nullptr;
    size_t n = 0, esz = sizeof(float);
    int rows = B, ld = 1;
    switch (which) {
        case 0: src = h->X_os; rows = 3 * B; ld = ic; break;
        case 1: src = h->X_bc; ld = ib; break;
        case 2: src = h->vel; ld = ap; break;
        case 3: src = h->w_act; ld = ap; break;
        case 4: src = h->X_c1; ld = ic; break;
        case 5: src = h->w_rew; break;
        case 6: src = h->w_mask; break;
        case 7: src = h->in_idx; esz = sizeof(int64_t); break;
        case 8: src = h->in_crop; esz = sizeof(int); ld = 2; break;
        case 9: src = h->X_eu; ld = ib; break;
        default: return FQL_E_INVALID;
    }
    if (!src) return FQL_E_NOTFOUND;
    n = (size_t)rows * ld * esz;
    if (dims) { dims[0] = rows; dims[1] = ld; }
    if (bytes < n) return FQL_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return FQL_E_HIP;
    return hipMemcpy(out, src, n, hipMemcpyDeviceToHost) == hipSuccess ? FQL_OK : FQL_E_HIP;
}

// Diagnostic only (not in include/fql_amd.h), -DFQL_TIMELINE builds: per launch of the single-graph update (prog_full) lane, op type,
// grid and the device wall-clock times (us, relative to the first entry) of its first / last workgroup entry and last exit.
// reset = 1 clears the table (call before the update to be examined).  Returns the number of launches, or < 0.
extern "C" int fql_debug_timeline(fql_handle h, int reset, int cap, int* lane, int* type, int* grid, double* t_first, double* t_last, double* t_exit) {
#ifdef FQL_TIMELINE
    if (!h) return FQL_E_INVALID;
    try { h->aql_drain(); } catch (...) { return FQL_E_HIP; }
    hipDeviceSynchronize();
    static std::vector<unsigned long long> host((size_t)FQL_TL_MAX * FQL_TL_WGS * 2);
    // updates dispatched as AQL packets run the engine's OWN copy of the code object: its instance of the table, not the one HIP registered
    void* aql_tab = nullptr;
    if (h->aql.up && h->last_update_aql) {
        hsa_executable_symbol_t sym;
        if (hsa_executable_get_symbol_by_name(h->aql.exe, "g_fql_tl", &h->aql.agent, &sym) == HSA_STATUS_SUCCESS) {
            uint64_t addr = 0;
            if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr) == HSA_STATUS_SUCCESS) aql_tab = (void*)(uintptr_t)addr;
        }
    }
    if (reset) {
        std::fill(host.begin(), host.end(), 0ull);
        if (h->aql.up) {   // both instances
            hsa_executable_symbol_t sym;
            uint64_t addr = 0;
            if (hsa_executable_get_symbol_by_name(h->aql.exe, "g_fql_tl", &h->aql.agent, &sym) == HSA_STATUS_SUCCESS &&
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr) == HSA_STATUS_SUCCESS && addr)
                if (hipMemcpy((void*)(uintptr_t)addr, host.data(), host.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return FQL_E_HIP;
        }
        return hipMemcpyToSymbol(HIP_SYMBOL(g_fql_tl), host.data(), host.size() * 8) == hipSuccess ? 0 : FQL_E_HIP;
    }
    if (aql_tab) { if (hipMemcpy(host.data(), aql_tab, host.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return FQL_E_HIP; }
    else if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(g_fql_tl), host.size() * 8) != hipSuccess) return FQL_E_HIP;
    const int n = std::min<int>((int)h->prog_full.launches.size(), std::min(cap, FQL_TL_MAX));
    unsigned long long t0 = ~0ull;
    std::vector<unsigned long long> a(n, ~0ull), b(n, 0), c(n, 0);
    for (int i = 0; i < n; ++i) {
        grid[i] = 0;
        for (int w = 0; w < FQL_TL_WGS; ++w) {
            const unsigned long long e = host[((size_t)i * FQL_TL_WGS + w) * 2], x = host[((size_t)i * FQL_TL_WGS + w) * 2 + 1];
            if (!e) continue;
            grid[i]++;
            a[i] = std::min(a[i], e); b[i] = std::max(b[i], e); c[i] = std::max(c[i], x);
        }
        if (grid[i]) t0 = std::min(t0, a[i]);
    }
    for (int i = 0; i < n; ++i) {
        const Launch& L = h->prog_full.launches[i];
        lane[i] = L.lane; type[i] = (int)L.type;
        t_first[i] = grid[i] ? (double)(a[i] - t0) / 100.0 : -1.0;
        t_last[i] = grid[i] ? (double)(b[i] - t0) / 100.0 : -1.0;
        t_exit[i] = c[i] ? (double)(c[i] - t0) / 100.0 : -1.0;
    }
    return n;
#else
    (void)h; (void)reset; (void)cap; (void)lane; (void)type; (void)grid; (void)t_first; (void)t_last; (void)t_exit;
    return FQL_E_STATE;
#endif
}

// Diagnostic only (not in include/fql_amd.h): copy an encoder-pass buffer to the host.  enc: 0 critic, 1 bc_flow, 2 onestep, 3 target;
// code: 100 s + {0: c0, 1: pool, 2: c1[0], 3: y[0], 4: arg (bytes)}, 900: frelu, 901: z, 902: E, 903: dz, 904: dA, 905: dB, 906: dC
extern "C" int fql_debug_enc(fql_handle h, int enc, int code, void* out, size_t bytes) {
    if (!h || !h->visual) return FQL_E_INVALID;
    EncBuf* b = enc == 0 ? &h->eb_c : enc == 1 ? &h->eb_bc : enc == 2 ? &h->eb_os : &h->eb_t;
    const void* src = nullptr;
    if (code >= 900) {
        const void* tab[] = {b->frelu, b->z, b->E, b->dz, b->dA, b->dB, b->dC};
        src = tab[code - 900];
    } else {
        EncBuf::St& s = b->st[code / 100];
        switch (code % 100) {
            case 0: src = s.c0; break;
            case 1: src = s.pool; break;
            case 2: src = s.c1[0]; break;
            case 3: src = s.y[0]; break;
            default: src = s.arg; break;
        }
    }
    if (!src) return FQL_E_NOTFOUND;
    hipDeviceSynchronize();
    return hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? FQL_OK : FQL_E_HIP;
}

int fql_dataset_upload_frames(fql_handle h, int64_t n, const uint8_t* frames, const uint8_t* next_frames, const float* act,
                              const float* rew, const float* mask, const float* terminals, int frame_stack, float p_aug) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        if (!h->visual) throw Invalid{"fql_dataset_upload_frames needs an agent created with an encoder"};
        if (n <= 0 || !frames || !next_frames || !act || !rew || !mask || !terminals) invalid("frames dataset pointers must not be NULL (n=%lld)", (long long)n);
        if (frame_stack < 1 || h->cfg.img_c % frame_stack) invalid("frame_stack %d must divide the %d image channels", frame_stack, h->cfg.img_c);
        if (p_aug < 0.f || p_aug > 1.f) invalid("p_aug must be in [0, 1]");
        HIP_CHECK(hipDeviceSynchronize());
        hipFree(h->ds_frames); hipFree(h->ds_next_frames); hipFree(h->ds_init);
        hipFree(h->ds_act); hipFree(h->ds_rew); hipFree(h->ds_mask);
        h->ds_frames = h->ds_next_frames = nullptr; h->ds_init = nullptr; h->ds_act = h->ds_rew = h->ds_mask = nullptr;
        const size_t fb = (size_t)h->cfg.img_h * h->cfg.img_w * (h->cfg.img_c / frame_stack);
        HIP_CHECK(hipMalloc((void**)&h->ds_frames, (size_t)n * fb));
        HIP_CHECK(hipMalloc((void**)&h->ds_next_frames, (size_t)n * fb));
        HIP_CHECK(hipMemcpy(h->ds_frames, frames, (size_t)n * fb, hipMemcpyDefault));
        HIP_CHECK(hipMemcpy(h->ds_next_frames, next_frames, (size_t)n * fb, hipMemcpyDefault));
        auto up = [&](float*& dst, const float* src, size_t w) {
            HIP_CHECK(hipMalloc((void**)&dst, (size_t)n * w * sizeof(float)));
            HIP_CHECK(hipMemcpy(dst, src, (size_t)n * w * sizeof(float), hipMemcpyDefault));
        };
        up(h->ds_act, act, h->cfg.act_dim); up(h->ds_rew, rew, 1); up(h->ds_mask, mask, 1);
        // utils/datasets.py:58-62: terminal_locs = nonzero(terminals > 0); initial_locs = [0, terminal_locs[:-1] + 1];
        // a row's episode start = the last initial_loc <= row (searchsorted(..., side='right') - 1, utils/datasets.py:75)
        std::vector<float> term((size_t)n);
        HIP_CHECK(hipMemcpy(term.data(), terminals, (size_t)n * sizeof(float), hipMemcpyDefault));
        std::vector<int64_t> tl;
        for (int64_t i = 0; i < n; ++i) if (term[i] > 0.f) tl.push_back(i);
        std::vector<int64_t> init((size_t)n);
        int64_t cur = 0;
        size_t k = 0;   // next terminal whose successor may start an episode; the LAST terminal never starts one (terminal_locs[:-1])
        for (int64_t i = 0; i < n; ++i) {
            while (k + 1 < tl.size() && tl[k] + 1 <= i) { cur = tl[k] + 1; ++k; }
            init[i] = cur;
        }
        HIP_CHECK(hipMalloc((void**)&h->ds_init, (size_t)n * sizeof(int64_t)));
        HIP_CHECK(hipMemcpy(h->ds_init, init.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
        h->ds_fs = frame_stack; h->ds_paug = p_aug;
        h->ds_size = n; h->ds_cap = n; h->ds_ptr = 0;
        h->src_valid = false;
    });
}
int fql_update_from_frames(fql_handle h, const int64_t* idx, const int32_t* crop_froms, int batch_size, int64_t lo, int64_t hi,
                           const fql_noise* noise, float* info13, void* stream) {
    if (!h) return FQL_E_INVALID;
    FQL_TRY(h, {
        hipStream_t s = pick(h, stream);
        h->source_from_frames(idx, crop_froms, batch_size, lo, hi, noise, s);
        if (h->prog_full.exec) run_program(h, h->prog_full, s);
        else { run_program(h, h->prog_fwdbwd, s); run_program(h, h->prog_opt, s); }
        h->finish_info(info13, FQL_NUM_INFO, s);
    });
}
