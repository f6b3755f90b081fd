// nn_host.cpp -- ResNet weights (blob layout of include/diee.h, BatchNorm folding, MFMA fragment
// packing) and the forward pass driver.  Reference: src/alphazero/nnet.rs:57-133.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "bg_device.h"
#include "engine.h"
#include "launch.h"
#include "nn_host.h"

namespace diee {

namespace {

constexpr int F = 256, BLOCKS = 19, A = 1352, CIN = 6, PH = 32, VH = 3;
constexpr int kFullRestBoards = 928;      // above it a remainder rides with the 4-board full-pass launch (kFullRest in nn_kernels.hip)

struct ConvOff { size_t w, b; int cout, cin; };
struct BnOff { size_t g, b, m, v; int c; };
struct BlobLayout {
    ConvOff init_conv; BnOff init_bn;
    ConvOff c1[BLOCKS], c2[BLOCKS]; BnOff b1[BLOCKS], b2[BLOCKS];
    ConvOff p_conv; BnOff p_bn; size_t p_fcw, p_fcb;
    ConvOff v_conv; BnOff v_bn; size_t v_fcw, v_fcb;
    size_t total;
};

BlobLayout make_layout() {
    BlobLayout L;
    size_t off = 0;
    auto conv = [&](int cout, int cin) { ConvOff c{off, 0, cout, cin}; off += (size_t)cout * cin * 9; c.b = off; off += cout; return c; };
    auto bn = [&](int c) { BnOff b{off, off + c, off + 2 * (size_t)c, off + 3 * (size_t)c, c}; off += 4 * (size_t)c; return b; };
    L.init_conv = conv(F, CIN); L.init_bn = bn(F);
    for (int i = 0; i < BLOCKS; ++i) {            // ResBlock::new creation order, nnet.rs:38-45
        L.c1[i] = conv(F, F); L.c2[i] = conv(F, F); L.b1[i] = bn(F); L.b2[i] = bn(F);
    }
    L.p_conv = conv(PH, F); L.p_bn = bn(PH); L.p_fcw = off; off += (size_t)A * PH * 24; L.p_fcb = off; off += A;
    L.v_conv = conv(VH, F); L.v_bn = bn(VH); L.v_fcw = off; off += (size_t)VH * 24; L.v_fcb = off; off += 1;
    L.total = off;
    return L;
}
const BlobLayout& layout() { static const BlobLayout L = make_layout(); return L; }

uint16_t f2bf_host(float x) {                       // round-to-nearest-even
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// fold eval-mode BatchNorm (eps 1e-5) into conv weight/bias: w' = w*s, b' = (b-mean)*s + beta
void fold(const float* blob, const ConvOff& c, const BnOff& bn, std::vector<float>& w, std::vector<float>& b) {
    w.assign((size_t)c.cout * c.cin * 9, 0.f); b.assign(c.cout, 0.f);
    for (int n = 0; n < c.cout; ++n) {
        const float s = blob[bn.g + n] / std::sqrt(blob[bn.v + n] + 1e-5f);
        for (int k = 0; k < c.cin * 9; ++k) w[(size_t)n * c.cin * 9 + k] = blob[c.w + (size_t)n * c.cin * 9 + k] * s;
        b[n] = (blob[c.b + n] - blob[bn.m + n]) * s + blob[bn.b + n];
    }
}

// pack folded conv weights w[cout][cin][3][3] into MFMA B fragments:
//   [nslice][kstep = cs*9 + tap][lane][j]:  n = nslice*32 + (lane&31),  c = cs*16 + 8*(lane>>5) + j
void pack_conv(const std::vector<float>& w, int cout, int cin, int n_pad, int cin_pad, std::vector<uint16_t>& out) {
    const int csteps = cin_pad / 16, ksteps = csteps * 9, nsl = n_pad / 32;
    out.assign((size_t)nsl * ksteps * 64 * 8, 0);
    for (int s = 0; s < nsl; ++s)
        for (int cs = 0; cs < csteps; ++cs)
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = s * 32 + (lane & 31), c = cs * 16 + 8 * (lane >> 5) + j;
                        if (n >= cout || c >= cin) continue;
                        out[(((size_t)s * ksteps + cs * 9 + t) * 64 + lane) * 8 + j] =
                            f2bf_host(w[((size_t)n * cin + c) * 9 + t]);
                    }
}

// 16-column B fragments for v_mfma_f32_16x16x32_bf16:
//   [n/16][kstep = cs32*9 + tap][lane][j]:  n = nf*16 + (lane&15),  c = cs32*32 + 8*(lane>>4) + j
void pack_conv16(const std::vector<float>& w, int cout, int cin, std::vector<uint16_t>& out) {
    const int csteps = cin / 32, ksteps = csteps * 9, nfr = cout / 16;
    out.assign((size_t)nfr * ksteps * 64 * 8, 0);
    for (int s = 0; s < nfr; ++s)
        for (int cs = 0; cs < csteps; ++cs)
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = s * 16 + (lane & 15), c = cs * 32 + 8 * (lane >> 4) + j;
                        out[(((size_t)s * ksteps + cs * 9 + t) * 64 + lane) * 8 + j] = f2bf_host(w[((size_t)n * cin + c) * 9 + t]);
                    }
}

// init block for the 16x16x32 path: one 32-channel k-step per tap, channels >= cin are zero
void pack_init16(const std::vector<float>& w, int cout, int cin, std::vector<uint16_t>& out) {
    const int nfr = cout / 16;
    out.assign((size_t)nfr * 9 * 64 * 8, 0);
    for (int s = 0; s < nfr; ++s)
        for (int t = 0; t < 9; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = s * 16 + (lane & 15), c = 8 * (lane >> 4) + j;
                    if (c < cin) out[(((size_t)s * 9 + t) * 64 + lane) * 8 + j] = f2bf_host(w[((size_t)n * cin + c) * 9 + t]);
                }
}

}  // namespace

size_t weights_count_bg() { return layout().total; }

// tch-default initialisation ([unvendored, from memory]): conv/linear weights
// U(+-1/sqrt(fan_in)), conv bias 0, linear bias U(+-1/sqrt(fan_in)), BN gamma U(0,1), beta 0,
// running mean 0, running var 1.  Philox keyed by (seed, tensor id): reproducible anywhere.
void random_weights_bg(uint64_t seed, float* blob) {
    const BlobLayout& L = layout();
    uint32_t tid = 0;
    auto uni = [&](size_t off, size_t n, float lo, float hi) {
        for (size_t i = 0; i < n; i += 4) {
            uint32_t o[4];
            philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(i / 4), (uint32_t)((i / 4) >> 32), tid, 0x57E16u, o);
            for (size_t k = 0; k < 4 && i + k < n; ++k)
                blob[off + i + k] = lo + (hi - lo) * ((float)(o[k] >> 8) * (1.0f / 16777216.0f));
        }
        ++tid;
    };
    auto cst = [&](size_t off, size_t n, float v) { for (size_t i = 0; i < n; ++i) blob[off + i] = v; };
    auto conv = [&](const ConvOff& c) {
        const float bd = 1.0f / std::sqrt((float)(c.cin * 9));
        uni(c.w, (size_t)c.cout * c.cin * 9, -bd, bd); cst(c.b, c.cout, 0.f);
    };
    auto bn = [&](const BnOff& b) { uni(b.g, b.c, 0.f, 1.f); cst(b.b, b.c, 0.f); cst(b.m, b.c, 0.f); cst(b.v, b.c, 1.f); };
    conv(L.init_conv); bn(L.init_bn);
    for (int i = 0; i < BLOCKS; ++i) { conv(L.c1[i]); conv(L.c2[i]); bn(L.b1[i]); bn(L.b2[i]); }
    conv(L.p_conv); bn(L.p_bn);
    { const float bd = 1.0f / std::sqrt((float)(PH * 24)); uni(L.p_fcw, (size_t)A * PH * 24, -bd, bd); uni(L.p_fcb, A, -bd, bd); }
    conv(L.v_conv); bn(L.v_bn);
    { const float bd = 1.0f / std::sqrt((float)(VH * 24)); uni(L.v_fcw, VH * 24, -bd, bd); uni(L.v_fcb, 1, -bd, bd); }
}

void free_net(NetWeights* n) { delete n; }

// "a:b,c:d" -> pairs (validated by Engine::set_option)
template <class Row>
static void parse_table(const std::string& t, std::vector<Row>& out) {
    out.clear();
    size_t pos = 0;
    while (pos < t.size() && t != "none") {
        const size_t c = t.find(':', pos), e2 = t.find(',', pos);
        if (c == std::string::npos) break;
        out.push_back({atoi(t.substr(pos, c - pos).c_str()), atoi(t.substr(c + 1, (e2 == std::string::npos ? t.size() : e2) - c - 1).c_str())});
        if (e2 == std::string::npos) break;
        pos = e2 + 1;
    }
}

// Options -> the network's dispatch state.  Called after load_weights, after every diee_set_option and when an in-launch hand-over
// starved (NetWeights::starved): the one place that decides which tower kernels this ctx may launch.
void Engine::apply_options() {
    if (!net) return;
    NetWeights& W = *net;
    W.fused_heads = opt.fused_heads != 0; W.cluster_heads = opt.cluster_heads != 0; W.cluster_init = opt.cluster_init != 0;
    W.compact = opt.compact != 0; W.cl_pack = opt.cl_pack != 0; W.trace_dispatch = opt.trace_dispatch != 0;
    // kernels whose workgroups wait for each other INSIDE a launch (cluster tower, pair tower) need the GPU to themselves
    const bool handoffs = !opt.shared_gpu && !W.starved;
    bool pair = opt.tower_pair != 0 && handoffs;
    if (pair && !tower_pair_device_ok(device)) {
        // both members of a pair must be resident together and -- their hand-off stores are plain: they stay in the XCD's L2 -- on
        // ONE XCD, which the kernel gets from blockIdx & 7 under round-robin dispatch to 8 XCDs; a device or partition that
        // is not 8 XCDs x >= 32 CUs would spin every pair launch to its timeout before the (loud) fallback
        pair = false;
        if (!W.told_no_pair) fprintf(stderr, "[diee] pair tower: device %d is not 8 XCDs x 32 CUs (a partition?); using the single-workgroup geometries\n", device);
        W.told_no_pair = true;
    }
    W.pair_tower = pair;
    if (opt.tower_table == "default") W.tower_table = pair ? NetWeights::default_tower_table() : NetWeights::default_tower_table_no_pair();
    else {
        std::vector<NetWeights::TowerRule> t;
        parse_table(opt.tower_table, t);
        for (const auto& r : t)
            if (r.geometry != 10 && r.geometry != 11 && !tower_geometry_supported(r.geometry)) {
                opt.tower_table = "default";
                throw EngineError(DIEE_ERR_ARG, "option tower_table: geometry " + std::to_string(r.geometry) + " is not in this build (product: 3, 5, 6, 14 and "
                                  "10 / 11 = the pair tower; python die-e_amd/build.py --dev has the others)");
            }
        W.tower_table = t;
    }
    if (!pair) for (auto& r : W.tower_table) if (r.geometry == 10 || r.geometry == 11) r.geometry = 3;      // (a table that names the pair tower)
    if (!handoffs) W.cluster_table.clear();
    else if (opt.tower_cl == "default") W.cluster_table = NetWeights::default_cluster_table();
    else parse_table(opt.tower_cl, W.cluster_table);
}

void Engine::load_weights(const float* blob, size_t n) {
    HIPCHK(hipSetDevice(device));
    const BlobLayout& L = layout();
    if (n != L.total) throw EngineError(DIEE_ERR_ARG, "weight blob has " + std::to_string(n) + " floats, expected " + std::to_string(L.total));
    if (!net) { net = new NetWeights(); nn_setup_kernels(); }
    apply_options();
    NetWeights& W = *net;
    std::vector<float> w, b;
    std::vector<uint16_t> pk;
    W.wtower.ensure((size_t)38 * 8 * 144 * 64 * 8);
    W.btower.ensure((size_t)38 * 256);
    W.wtower16.ensure((size_t)38 * 16 * 72 * 64 * 8);
    auto up_conv = [&](int layer, const ConvOff& c, const BnOff& bn, int n_pad, int cin_pad) {
        fold(blob, c, bn, w, b);
        pack_conv(w, c.cout, c.cin, n_pad, cin_pad, pk);
        std::vector<float> bp(n_pad, 0.f);
        memcpy(bp.data(), b.data(), sizeof(float) * c.cout);
        if (layer >= 1 && layer <= 38) {
            h2d(W.wl(layer), pk.data(), pk.size());
            h2d(W.bl(layer), bp.data(), (size_t)n_pad);
            sync();
            pack_conv16(w, c.cout, c.cin, pk);
            h2d(W.wtower16.p + (size_t)(layer - 1) * 16 * 72 * 64 * 8, pk.data(), pk.size());
        } else {
            W.wconv[layer].ensure(pk.size());
            h2d(W.wconv[layer].p, pk.data(), pk.size());
            W.bconv[layer].ensure(n_pad);
            h2d(W.bconv[layer].p, bp.data(), (size_t)n_pad);
        }
        sync();     // pk / bp are reused
    };
    up_conv(0, L.init_conv, L.init_bn, 256, 16);
    pack_init16(w, 256, CIN, pk);                  // `w` still holds the folded init-block weights
    W.winit16.ensure(pk.size()); h2d(W.winit16.p, pk.data(), pk.size()); sync();
    for (int i = 0; i < BLOCKS; ++i) { up_conv(1 + 2 * i, L.c1[i], L.b1[i], 256, 256); up_conv(2 + 2 * i, L.c2[i], L.b2[i], 256, 256); }
    {   // heads share one conv launch: channels 0..31 policy (nnet.rs:76), 32..34 value (nnet.rs:88)
        std::vector<float> wp, bp, wv, bv;
        fold(blob, L.p_conv, L.p_bn, wp, bp); fold(blob, L.v_conv, L.v_bn, wv, bv);
        w.assign((size_t)35 * F * 9, 0.f); b.assign(35, 0.f);
        memcpy(w.data(), wp.data(), sizeof(float) * wp.size()); memcpy(w.data() + wp.size(), wv.data(), sizeof(float) * wv.size());
        memcpy(b.data(), bp.data(), sizeof(float) * 32); memcpy(b.data() + 32, bv.data(), sizeof(float) * 3);
        pack_conv(w, 35, F, 64, 256, pk);
        W.wconv[39].ensure(pk.size()); h2d(W.wconv[39].p, pk.data(), pk.size());
        sync();
        {   // the same head convs as 16-column fragments (4 fragments = 64 channels, 35 real)
            std::vector<float> w64((size_t)64 * F * 9, 0.f);
            memcpy(w64.data(), w.data(), sizeof(float) * w.size());
            pack_conv16(w64, 64, F, pk);
            W.whead16.ensure(pk.size()); h2d(W.whead16.p, pk.data(), pk.size());
            sync();
        }
        std::vector<float> bpad(64, 0.f); memcpy(bpad.data(), b.data(), sizeof(float) * 35);
        W.bconv[39].ensure(64); h2d(W.bconv[39].p, bpad.data(), (size_t)64);
        sync();
    }
    {   // policy FC [1352][768], reference k = c*24 + p (flatten of [32][4][6], nnet.rs:79); ours k' = p*32 + c
        const int nsl = 43, ksteps = 48;
        pk.assign((size_t)nsl * ksteps * 64 * 8, 0);
        for (int s = 0; s < nsl; ++s)
            for (int ks = 0; ks < ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int a = s * 32 + (lane & 31), kp = ks * 16 + 8 * (lane >> 5) + j;
                        if (a >= A) continue;
                        const int p = kp / 32, c = kp % 32;
                        pk[(((size_t)s * ksteps + ks) * 64 + lane) * 8 + j] = f2bf_host(blob[L.p_fcw + (size_t)a * 768 + c * 24 + p]);
                    }
        W.wfc.ensure(pk.size()); h2d(W.wfc.p, pk.data(), pk.size());
        std::vector<float> bf(1376, 0.f); memcpy(bf.data(), blob + L.p_fcb, sizeof(float) * A);
        W.bfc.ensure(1376); h2d(W.bfc.p, bf.data(), (size_t)1376);
        std::vector<float> wv(73, 0.f);             // value FC, ours k' = p*3 + c
        for (int p = 0; p < 24; ++p) for (int c = 0; c < 3; ++c) wv[p * 3 + c] = blob[L.v_fcw + c * 24 + p];
        wv[72] = blob[L.v_fcb];
        W.wv.ensure(73); h2d(W.wv.p, wv.data(), (size_t)73);
        sync();
    }
    W.loaded = true;
}

static uint16_t* pair_exchange(Engine& e) {
    NetWeights& W = *e.net;
    if (!W.pair_tower) return nullptr;
    if (!W.pair_ex.p) {
        W.pair_ex.ensure(tower_pair_exchange_bytes() / 2);
        HIPCHK(hipMemsetAsync(W.pair_ex.p, 0, tower_pair_exchange_bytes(), e.stream));     // no tag set anywhere: the state every launch leaves behind
    }
    return W.pair_ex.p;
}

void nn_reserve(Engine& e, int G) {
    NetWeights& W = *e.net;
    if (G <= W.cap_games) return;
    const size_t Gp = (size_t)((G + 7) / 8 * 8), M = Gp * 24;
    W.x16.ensure(M * 16); W.actX.ensure(M * 256); W.actH.ensure(M * 256);
    // the cluster tower's first tagged write into X relies on the sign bits it finds there being clear (plain data)
    HIPCHK(hipMemsetAsync(W.actX.p, 0, M * 256 * sizeof(uint16_t), e.stream));
    W.hp.ensure(Gp * 768); W.hv.ensure(Gp * 72); W.logits.ensure(Gp * 1352);
    W.cap_games = (int)Gp;
}

// forward_t on G device-resident states -> policy_dev [G][1352] (softmax), value_dev [G] (tanh)
// The cluster tower's workgroups wait for each other inside the launch, so all of them must be resident together.
// One launch never asks for more than the device holds, but two such launches from two contexts of this process on
// the same GPU could each get half of the CUs and wait for the other half forever (until the bounded spins give up).
// Contexts sharing a device therefore pass a per-device baton: a cluster launch waits for the previous one's event.
// With one context per GPU (the documented use) the baton is never touched.
namespace {
struct ClusterBaton { std::mutex mu; hipEvent_t ev = nullptr; int engines = 0; };
ClusterBaton g_baton[16];
}
void cluster_baton_register(int device, int delta) {
    if (device < 0 || device >= 16) return;
    std::lock_guard<std::mutex> lk(g_baton[device].mu);
    g_baton[device].engines += delta;
}

// after a timed-out hand-over (reported as DIEE_ERR_HIP): zero the cluster counters so that later launches start clean
void nn_reset_cluster(Engine& e) {
    if (e.net && e.net->cl_sync.p)
        (void)hipMemsetAsync(e.net->cl_sync.p, 0, (size_t)kClusterMaxGroups * 32 * sizeof(uint32_t), e.stream);
}

// small batches: the 38 tower layers in one launch (k_tower_cl).  false = no rule takes this batch size, or the grid
// would not be co-resident on this device: the caller runs the per-layer kernels.
// hv / logits non-null: the launch runs the head convs and the policy FC too (rows of this chunk)
static bool cluster_tower(Engine& e, NetWeights& W, int G, const void* states, uint16_t* actX, uint16_t* actH, float* hv = nullptr,
                          float* logits = nullptr, const uint32_t* n_rows_dev = nullptr, uint32_t* rows_log = nullptr) {
    const void* winit = W.wconv[0].p; const float* binit = W.bconv[0].p;
    const void* whead = logits ? W.wconv[39].p : nullptr;
    for (const auto& r : W.cluster_table) {
        if (G > r.max_games) continue;
        if (!W.cl_sync.p) {
            W.cl_sync.ensure((size_t)kClusterMaxGroups * 32);
            HIPCHK(hipMemsetAsync(W.cl_sync.p, 0, (size_t)kClusterMaxGroups * 32 * sizeof(uint32_t), e.stream));
        }
        ClusterBaton* bt = (e.device >= 0 && e.device < 16) ? &g_baton[e.device] : nullptr;
        if (bt) {
            std::unique_lock<std::mutex> lk(bt->mu);
            if (bt->engines > 1) {
                if (!bt->ev) HIPCHK(hipEventCreateWithFlags(&bt->ev, hipEventDisableTiming));
                else HIPCHK(hipStreamWaitEvent(e.stream, bt->ev, 0));
                bool took = false;                                  // (an earlier chunk of this evaluation may have grown the tree already: only ever set)
                const bool ok = launch_tower_cluster(e.stream, e.device, r.boards_per_group, actX, actH, W.wtower.p, W.btower.p, G, W.cl_sync.p, e.flags_dev.p, states, winit, binit,
                                                     whead, W.bconv[39].p, W.wfc.p, W.bfc.p, hv, logits, W.grow_done ? nullptr : W.grow_req, &took, W.cl_pack, n_rows_dev, rows_log);
                if (took) W.grow_done = true;
                if (ok) HIPCHK(hipEventRecord(bt->ev, e.stream));
                return ok;
            }
        }
        {
            bool took = false;
            const bool ok = launch_tower_cluster(e.stream, e.device, r.boards_per_group, actX, actH, W.wtower.p, W.btower.p, G, W.cl_sync.p, e.flags_dev.p, states, winit, binit,
                                                 whead, W.bconv[39].p, W.wfc.p, W.bfc.p, hv, logits, W.grow_done ? nullptr : W.grow_req, &took, W.cl_pack, n_rows_dev, rows_log);
            if (took) W.grow_done = true;
            return ok;
        }
    }
    return false;
}

static int cluster_boards_for(const NetWeights& W, int G) {
    for (const auto& r : W.cluster_table) if (G <= r.max_games) return r.boards_per_group;
    return 0;
}

// the convolutional part of the network (init block, 38-layer tower, head convs) for rows [off, off + G) of the batch:
// states -> hp / hv.  The tower kernel is picked by G alone (tower_table / cluster_table).
// Returns true when the launch produced the chunk's logits as well (cluster tower with the heads and the FC inside).
static bool nn_conv_chunk(Engine& e, const void* states_all, int off, int G, bool fused_family = false) {
    NetWeights& W = *e.net;
    hipStream_t st = e.stream;
    const void* states_dev = (const uint8_t*)states_all + (size_t)off * 32;
    uint16_t* actX = W.actX.p + (size_t)off * 24 * 256;
    uint16_t* actH = W.actH.p + (size_t)off * 24 * 256;
    uint16_t* hp = W.hp.p + (size_t)off * 768;
    float* hv = W.hv.p + (size_t)off * 72;
    const bool sample = W.sample_every > 0 && (W.forward_count++ % W.sample_every) == 0;
    int tgeom = W.tower_geometry_for(G);
    if (tgeom < 0 && fused_family) tgeom = 3;                   // the remainder of a batch above 256 boards: never the split-K family
    const bool trace_dispatch = W.trace_dispatch;               // development: which tower path a batch takes
    // sampled timing of the tower: one HIP-event pair per sampled forward (per-launch pairs cost ~4.6 us each and
    // inflate a ~30 us kernel by 14 %; the chain amortises that to < 1 %)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int kind = 1;
    auto stamp0 = [&] { if (sample) { ev0 = W.get_event(); ev1 = W.get_event(); HIPCHK(hipEventRecord(ev0, st)); } };
    // the init block reads the states and builds the input planes itself (no separate planes kernel)
    auto init_block = [&] {
        launch_conv3x3(st, 16, 0, (const uint16_t*)states_dev, W.wconv[0].p, W.bconv[0].p, nullptr, actX, nullptr, G, 256);
    };
    float* logits = W.logits.p + (size_t)off * 1352;
    auto cluster = [&](const void* states, bool with_heads) {
        const bool ok = cluster_tower(e, W, G, states, actX, actH, with_heads ? hv : nullptr, with_heads ? logits : nullptr);
        if (ok) { W.cluster_used = true; W.last_dispatch.push_back({3, cluster_boards_for(W, G), G}); }
        return ok;
    };
    bool done = false, heads_done = false, fc_done = false;
    if (tgeom < 0 && !W.cluster_table.empty() && W.cluster_init) {
        // small batches: init block + all 38 layers (+ head convs + policy FC) in ONE launch, 8-workgroup clusters per board group
        stamp0();
        if (cluster(states_dev, W.cluster_heads)) { kind = 2; done = true; heads_done = fc_done = W.cluster_heads; }
        else if (sample) { W.free_events.push_back(ev0); W.free_events.push_back(ev1); ev0 = ev1 = nullptr; }
    }
    if (!done && (tgeom == 10 || tgeom == 11)) {
        // 257 ... 512 boards: the fused tower on pairs of workgroups, 4 boards per pair (10; 11: 2 boards per pair); init block and head convs inside
        const int bpp = tgeom == 10 ? 4 : 2;
        stamp0();
        if (W.pair_tower && W.fused_heads && W.cluster_init && G <= tower_pair_max_boards(bpp) &&
            launch_tower_pair(st, bpp, W.wtower16.p, W.btower.p, G, states_dev, W.winit16.p, W.bconv[0].p, W.whead16.p, W.bconv[39].p, hp, hv, pair_exchange(e), e.flags_dev.p)) {
            done = true; heads_done = true; W.cluster_used = true;    // (its hand-overs report through the same flag bit as the cluster tower's)
            W.last_dispatch.push_back({2, bpp, G});
            if (bpp == 2) kind = 2;                                    // the sampled timings go by band: <= 256 boards is the small-batch row
        } else {
            if (sample) { W.free_events.push_back(ev0); W.free_events.push_back(ev1); ev0 = ev1 = nullptr; }
            tgeom = 3;
        }
    }
    if (!done && tgeom >= 0 && tower_geometry_has_init(tgeom) && W.cluster_init) {
        // large batches: init block + all 38 layers in one launch, activations stay in LDS
        stamp0();
        launch_tower(st, tgeom, actX, W.wtower.p, W.wtower16.p, W.btower.p, actX, G, states_dev, W.winit16.p, W.bconv[0].p,
                     W.fused_heads ? W.whead16.p : nullptr, W.bconv[39].p, hp, hv);
        done = true; heads_done = W.fused_heads;
        W.last_dispatch.push_back({1, tgeom, G});
    }
    if (!done) {
        init_block();
        if (!ev0) stamp0();
        if (tgeom >= 0) {
            launch_tower(st, tgeom, actX, W.wtower.p, W.wtower16.p, W.btower.p, actX, G);   // all 38 layers, activations stay in LDS
            W.last_dispatch.push_back({tgeom <= 1 ? 4 : 1, tgeom, G});
        } else if (cluster(nullptr, false)) {
            kind = 2;                                                   // (init block launched separately: DIEE_CLUSTER_INIT=0)
        } else {
            kind = 0;
            W.last_dispatch.push_back({0, 0, G});
            for (int i = 0; i < BLOCKS; ++i) {
                launch_conv3x3(st, 256, 0, actX, W.wl(1 + 2 * i), W.bl(1 + 2 * i), nullptr, actH, nullptr, G, 256);
                // y = relu(conv2(h) + x), written in place over x (each element is read and written by one lane)
                launch_conv3x3(st, 256, 1, actH, W.wl(2 + 2 * i), W.bl(2 + 2 * i), actX, actX, nullptr, G, 256);
            }
        }
    }
    if (trace_dispatch)
        fprintf(stderr, "[diee] forward of boards %d..%d: %s (fused geometry %d, conv variant table entries %zu / %zu)\n", off, off + G,
                kind == 1 ? "fused tower" : kind == 2 ? "cluster tower" : "per-layer kernels", tgeom, W.tower_table.size(), W.cluster_table.size());
    if (sample) {
        HIPCHK(hipEventRecord(ev1, st));
        // kind 3: the whole chunk in ONE launch of the full-chip instantiation (geometry 5, k_tower16<4,4,3>): the dominant kernel, sampled one to one
        W.pending.push_back({ev0, ev1, 38.0 * 2.0 * G * 24.0 * 2304.0 * 256.0, kind == 0 ? 38 : 1, (kind == 1 && tower_geometry_is_full_chip(tgeom)) ? 3 : kind, -1, G});
    }
    if (!heads_done) launch_conv3x3(st, 256, 2, actX, W.wconv[39].p, W.bconv[39].p, nullptr, hp, hv, G, 64);
    return fc_done;
}

static void fc_launch(Engine& e, const uint16_t* hp, float* logits, int G, const uint32_t* n_rows = nullptr) {
    NetWeights& W = *e.net;
    launch_policy_fc(e.stream, hp, W.wfc.p, W.bfc.p, logits, G, n_rows);
}

// forward_t on G device-resident states -> policy_dev [G][1352] (softmax), value_dev [G] (tanh).
// More boards than one pass of the chip holds (256 CUs x 4 boards): the whole multiples of kFullChip go through ONE
// launch of the 4-board fused tower (its workgroups run in full rounds), the remainder through the fused geometry that is
// fastest at the remainder's size -- a partial last round of 4-board workgroups would cost a full round.  Above 256
// boards every launch is of the fused 16x16x32 family, so a row's result does not depend on where in the batch it sits
// (below, the latency-optimised split-K cluster tower takes the whole batch).
//
// rows != nullptr: the batch is COMPACTED on the device (search iterations above 256 live games): only the slots with
// rows->skip[slot] == 0 are evaluated, row r of the outputs belongs to slot rows->row_slot[r]; returns true when it did so
// (the caller then reads outputs through rows->slot_row).
bool nn_forward(Engine& e, const void* states_dev, int G, float* policy_dev, float* value_dev, const NnRows* rows) {
    if (!e.net || !e.net->loaded) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    if (G <= 0) return false;
    NetWeights& W = *e.net;
    nn_reserve(e, G);
    W.last_dispatch.clear();
    hipStream_t st = e.stream;
    const int tg = W.tower_geometry_for(G);
    // (one full pass of the chip, 929 ... 1024 live games, is the opening and middle game: hardly a terminal leaf, and the row
    // map + the two remainder launches that stay empty cost 14 us per evaluation: those batches run plain)
    const bool one_full_pass = G > kFullRestBoards && G <= W.full_chip_boards;
    if (rows && W.compact && !policy_dev && G > W.compact_above && !one_full_pass && tg >= 2 && W.cluster_init && W.fused_heads) {
        const uint32_t seq = (uint32_t)(W.forward_count & (kRowsLog - 1));
        const bool sample = W.sample_every > 0 && (W.forward_count++ % W.sample_every) == 0;
        W.rows_log.ensure(kRowsLog);
        launch_row_map(st, rows->skip, (uint32_t)G, rows->row_slot, rows->slot_row, rows->n_rows, W.rows_log.p, seq);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (sample) { ev0 = W.get_event(); ev1 = W.get_event(); HIPCHK(hipEventRecord(ev0, st)); }
        uint16_t* pex = pair_exchange(e);
        if (pex) W.cluster_used = true;
        launch_tower_compact(st, W.wtower16.p, W.btower.p, G, states_dev, W.winit16.p, W.bconv[0].p, W.whead16.p, W.bconv[39].p,
                             W.hp.p, W.hv.p, rows->row_slot, rows->n_rows, pex, e.flags_dev.p);
        W.last_dispatch.push_back({1, -1, G});                      // compacted: up to three fused launches, each workgroup decides from the device-side row count
        if (sample) {
            HIPCHK(hipEventRecord(ev1, st));
            W.pending.push_back({ev0, ev1, 38.0 * 2.0 * 24.0 * 2304.0 * 256.0, 1, 1, (int)seq, G});     // flops per ROW: the row count is in rows_log[seq]
        }
        fc_launch(e, W.hp.p, W.logits.p, G, rows->n_rows);
        HIPCHK(hipGetLastError());
        return true;
    }
    const int full = W.full_chip_boards;
    int main_rows = G;
    if (full > 0 && G > full && tg >= 0) {
        main_rows = G / full * full;
        if (W.tower_geometry_for(G - main_rows) == tg) main_rows = G;   // the remainder would take the same kernel
    }
    const bool fc_main = nn_conv_chunk(e, states_dev, 0, main_rows);
    const bool fc_rest = main_rows < G ? nn_conv_chunk(e, states_dev, main_rows, G - main_rows, true) : true;
    // the policy FC for the rows whose launch did not run it itself (the cluster tower does)
    if (!fc_main && !fc_rest) fc_launch(e, W.hp.p, W.logits.p, G);
    else if (!fc_main) fc_launch(e, W.hp.p, W.logits.p, main_rows);
    else if (!fc_rest) fc_launch(e, W.hp.p + (size_t)main_rows * 768, W.logits.p + (size_t)main_rows * 1352, G - main_rows);
    if (policy_dev) launch_softmax_value(st, W.logits.p, W.hv.p, W.wv.p, policy_dev, value_dev, G);
    HIPCHK(hipGetLastError());
    return false;
}

// The tail of a batch (search_types.h, Tail): ONE cluster-tower launch (init block, 38 layers, head convs, policy FC inside) over up to
// G_upper states whose number k_tail wrote to *n_rows_dev on the device, outputs straight into the caller's ring rows.  The kernel is
// the one every batch of at most 32 boards runs (k_tower_cl<1, 8>: the same bits per row whatever shares the launch).  false = this
// ctx cannot (no such cluster rule, or the grid would not be co-resident): the caller searches launch by launch instead.
bool nn_tail_available(Engine& e, int G_upper, int n) {
    if (!e.net || !e.net->loaded) return false;
    const NetWeights& W = *e.net;
    // the rows of a tail launch must be evaluated with the arithmetic of a plain evaluation of the n live games (what the oracle's evaluator runs)
    // 512 rows: the plain evaluations of these n games are of the fused 16x16x32 family (pair tower <2>, or its fallback geometry), and so is
    // the 4-board pair tower that takes a tail launch's rows: one arithmetic per row again
    if (G_upper == kTailFusedRows)
        return W.pair_tower && W.fused_heads && W.cluster_init && tower_pair_max_boards(4) >= kTailFusedRows && W.tower_geometry_for(n) >= 2;
    if (!W.cluster_init || !W.cluster_heads || W.invariant || W.tower_geometry_for(n) >= 0) return false;
    // (1, 2 and 4 boards per cluster split K over 8 waves: one arithmetic; 8 boards per cluster splits it over 4: another)
    auto split8 = [&](int G) {
        for (const auto& r : W.cluster_table) if (G <= r.max_games) return r.boards_per_group == 1 || r.boards_per_group == 2 || r.boards_per_group == 4;
        return false;
    };
    return split8(G_upper) && split8(n);
}
bool nn_forward_tail(Engine& e, const void* states_dev, int G_upper, const uint32_t* n_rows_dev, float* hv_out, float* logits_out, int boards_band,
                     const uint32_t* n_dem_dev) {
    NetWeights& W = *e.net;
    nn_reserve(e, G_upper);
    const uint32_t seq = (uint32_t)(W.forward_count & (kRowsLog - 1));
    const bool sample = W.sample_every > 0 && (W.forward_count++ % W.sample_every) == 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t* rows_log = nullptr;
    if (sample) {
        W.rows_log.ensure(kRowsLog);
        HIPCHK(hipMemsetAsync(W.rows_log.p + seq, 0, sizeof(uint32_t), e.stream));
        rows_log = W.rows_log.p + seq;
        if (n_dem_dev) { W.dem_log.ensure(kRowsLog); HIPCHK(hipMemcpyAsync(W.dem_log.p + seq, n_dem_dev, sizeof(uint32_t), hipMemcpyDeviceToDevice, e.stream)); }
        ev0 = W.get_event(); ev1 = W.get_event(); HIPCHK(hipEventRecord(ev0, e.stream));
    }
    bool ok;
    W.last_dispatch.clear();
    if (G_upper == kTailFusedRows) {
        uint16_t* pex = pair_exchange(e);
        ok = pex != nullptr;
        if (ok) {
            launch_tower_pair_counted(e.stream, W.wtower16.p, W.btower.p, G_upper, states_dev, W.winit16.p, W.bconv[0].p, W.whead16.p, W.bconv[39].p,
                                      W.hp.p, hv_out, pex, e.flags_dev.p, n_rows_dev);
            fc_launch(e, W.hp.p, logits_out, G_upper, n_rows_dev);
            if (rows_log) HIPCHK(hipMemcpyAsync(rows_log, n_rows_dev, sizeof(uint32_t), hipMemcpyDeviceToDevice, e.stream));
            W.cluster_used = true;                                       // (its hand-overs report through the same flag bit)
            W.last_dispatch.push_back({2, 4, G_upper});
        }
    } else {
        const struct GrowReq* saved = W.grow_req; W.grow_req = nullptr;      // (k_tail grows the tree itself)
        ok = cluster_tower(e, W, G_upper, states_dev, W.actX.p, W.actH.p, hv_out, logits_out, n_rows_dev, rows_log);
        W.grow_req = saved;
        if (ok) { W.cluster_used = true; W.last_dispatch.push_back({3, cluster_boards_for(W, G_upper), G_upper}); }
    }
    if (sample) {
        if (ok) {
            HIPCHK(hipEventRecord(ev1, e.stream));
            W.pending.push_back({ev0, ev1, 38.0 * 2.0 * 24.0 * 2304.0 * 256.0, 1, 2, (int)seq, boards_band, n_dem_dev ? (int)seq : -1, W.cur_step});      // flops per ROW; a launch without rows is dropped at the harvest
        } else { W.free_events.push_back(ev0); W.free_events.push_back(ev1); }
    }
    return ok;
}

// The free-running search (search_types.h, Free): one evaluation of up to rows_upper rows whose states are gathered from the tree arena by
// index (`rows_idx[row]` = slot * node_cap + node) and whose number k_free_pack wrote to *n_rows_dev -- the fused tower's compacted-batch
// launches (launch_tower_compact: one pass of the chip for 929 ... 1024 rows, the part-filled instantiation or the pair tower below) + the
// policy FC, outputs straight into the caller's ring rows.  Every launch is of the fused 16x16x32 family: a row's bits are those of a plain
// evaluation of more than 128 boards, whichever launch computes it.
constexpr int kFreeOnePass = 1024;
bool nn_free_available(Engine& e, int n) {
    if (!e.net || !e.net->loaded) return false;
    const NetWeights& W = *e.net;
    return W.fused_heads && W.cluster_init && W.tower_geometry_for(n) >= 2;       // the plain evaluations of these n games are of the fused family too
}
void nn_forward_free(Engine& e, const void* arena_states, const uint32_t* rows_idx, const uint32_t* n_rows_dev, int rows_upper, float* hv_out, float* logits_out, int boards_band,
                     const uint32_t* n_dem_dev) {
    NetWeights& W = *e.net;
    nn_reserve(e, rows_upper);
    hipStream_t st = e.stream;
    const uint32_t seq = (uint32_t)(W.forward_count & (kRowsLog - 1));
    const bool sample = W.sample_every > 0 && (W.forward_count++ % W.sample_every) == 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (sample) {
        W.rows_log.ensure(kRowsLog);
        HIPCHK(hipMemcpyAsync(W.rows_log.p + seq, n_rows_dev, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        if (n_dem_dev) { W.dem_log.ensure(kRowsLog); HIPCHK(hipMemcpyAsync(W.dem_log.p + seq, n_dem_dev, sizeof(uint32_t), hipMemcpyDeviceToDevice, st)); }
        ev0 = W.get_event(); ev1 = W.get_event(); HIPCHK(hipEventRecord(ev0, st));
    }
    uint16_t* pex = pair_exchange(e);
    if (pex) W.cluster_used = true;                                 // (the pair tower's hand-overs report through the starved-hand-over bit)
    if (rows_upper <= kFreeOnePass)
        launch_tower_free(st, W.wtower16.p, W.btower.p, rows_upper, arena_states, W.winit16.p, W.bconv[0].p, W.whead16.p, W.bconv[39].p,
                          W.hp.p, hv_out, rows_idx, n_rows_dev, pex, e.flags_dev.p);
    else
        launch_tower_compact(st, W.wtower16.p, W.btower.p, rows_upper, arena_states, W.winit16.p, W.bconv[0].p, W.whead16.p, W.bconv[39].p,
                             W.hp.p, hv_out, rows_idx, n_rows_dev, pex, e.flags_dev.p);
    W.last_dispatch.clear();
    W.last_dispatch.push_back({1, -1, rows_upper});
    if (sample) {
        HIPCHK(hipEventRecord(ev1, st));
        W.pending.push_back({ev0, ev1, 38.0 * 2.0 * 24.0 * 2304.0 * 256.0, 1, 1, (int)seq, boards_band, n_dem_dev ? (int)seq : -1, W.cur_step});     // flops per ROW; a launch without rows is dropped at the harvest
    }
    fc_launch(e, W.hp.p, logits_out, rows_upper, n_rows_dev);
}

// ---- development probes: which kernel a (family, geometry) is, and the dispatch of a plain evaluation by its board count ----
const char* nn_kernel_name(int family, int geometry) {
    switch (family) {
        case 0: return "k_conv3x3 / k_conv3x3_sk (per-layer)";
        case 1: return tower_geometry_name(geometry);
        case 2: return geometry == 4 ? "k_tower16p<4, 6>" : geometry == 2 ? "k_tower16p<2, 6>" : "k_tower16p<?>";
        case 3: return geometry == 1 ? "k_tower_cl<1, 8>" : geometry == 2 ? "k_tower_cl<2, 8>" : geometry == 4 ? "k_tower_cl<4, 8>" : geometry == 8 ? "k_tower_cl<8, 4>" : "k_tower_cl<?>";
        case 4: return tower_geometry_name(geometry);
        default: return "?";
    }
}
// the kernel nn_conv_chunk picks for a whole batch of G boards when every launch succeeds (the same decisions, no launch)
static NetWeights::Launched dispatch_of(const NetWeights& W, int G) {
    int tgeom = W.tower_geometry_for(G);
    if (tgeom < 0 && !W.cluster_table.empty() && W.cluster_init && cluster_boards_for(W, G) > 0) return {3, cluster_boards_for(W, G), G};
    if (tgeom == 10 || tgeom == 11) {
        const int bpp = tgeom == 10 ? 4 : 2;
        if (W.pair_tower && W.fused_heads && W.cluster_init && G <= tower_pair_max_boards(bpp)) return {2, bpp, G};
        tgeom = 3;
    }
    if (tgeom >= 0) return {tgeom <= 1 ? 4 : 1, tgeom, G};
    if (cluster_boards_for(W, G) > 0) return {3, cluster_boards_for(W, G), G};
    return {0, 0, G};
}
std::vector<DispatchBand> nn_dispatch_bands(Engine& e, int upto) {
    std::vector<DispatchBand> out;
    if (!e.net) return out;
    for (int G = 1; G <= upto; ++G) {
        const NetWeights::Launched d = dispatch_of(*e.net, G);
        if (!out.empty() && out.back().family == d.family && out.back().geometry == d.geometry) out.back().boards_max = G;
        else out.push_back({G, G, d.family, d.geometry});
    }
    return out;
}

// the cluster tower ran since the last call (its hand-over flag is worth a look)
bool nn_cluster_used(Engine& e) {
    if (!e.net) return false;
    const bool u = e.net->cluster_used;
    e.net->cluster_used = false;
    return u;
}

// a cluster hand-over starved: never launch the cluster tower again in this process (batches it took run on the
// per-layer kernels), clear the flag bit and re-arm the hand-over counters
void nn_disable_cluster(Engine& e) {
    if (!e.net) return;
    e.net->starved = true;                  // the pair tower hands over inside its launch too: apply_options drops both
    e.apply_options();
    uint32_t f = 0;
    e.d2h(&f, e.flags_dev.p, 1);
    e.sync();
    f &= ~4u;
    e.h2d(e.flags_dev.p, &f, 1);
    e.sync();
    nn_reset_cluster(e);
}

NetHeads nn_heads(Engine& e, int G) {
    nn_reserve(e, G);
    NetWeights& W = *e.net;
    return NetHeads{W.logits.p, W.hv.p, W.wv.p};
}

// harvest sampled conv timings (call after a stream sync)
void nn_harvest(Engine& e, diee_stats* stats, const std::vector<unsigned long long>* step_log) {
    if (!e.net) return;
    NetWeights& W = *e.net;
    std::vector<uint32_t> rows_log, dem_log;
    for (auto& p : W.pending) if (p.rows_seq >= 0 && rows_log.empty()) {
        rows_log.resize(kRowsLog);
        e.d2h(rows_log.data(), W.rows_log.p, (size_t)kRowsLog); e.sync();
    }
    for (auto& p : W.pending) if (p.dem_seq >= 0 && dem_log.empty()) {
        dem_log.resize(kRowsLog);
        e.d2h(dem_log.data(), W.dem_log.p, (size_t)kRowsLog); e.sync();
    }
    for (auto& p : W.pending) {
        float ms = 0.f;
        if (p.rows_seq >= 0) p.flops *= (double)rows_log[(size_t)p.rows_seq];      // compacted batch: flops per row x rows evaluated
        // the rows the SEARCH used: every row of a plain or compacted evaluation; of a tail / free-running launch (rows evaluated ahead of the
        // search: which of them it goes on to expand is known only afterwards) the share its move-step's search used, expansions / rows evaluated
        double flops_dem = p.flops;
        if (p.dem_seq >= 0) {
            double share = 0.0;
            if (step_log && p.step >= 0 && 2 * (size_t)p.step + 1 < step_log->size() && (*step_log)[2 * (size_t)p.step + 1] > 0)
                share = std::min(1.0, (double)(*step_log)[2 * (size_t)p.step] / (double)(*step_log)[2 * (size_t)p.step + 1]);
            else if (rows_log[(size_t)p.rows_seq] > 0) share = std::min(1.0, (double)dem_log[(size_t)p.dem_seq] / (double)rows_log[(size_t)p.rows_seq]);   // (no log: the rows demanded when it was launched)
            flops_dem = p.flops * share;
        }
        const bool empty = p.rows_seq >= 0 && rows_log[(size_t)p.rows_seq] == 0;   // (a tail launch sent ahead of a search that was complete already)
        if (!empty && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (p.kind == 1 || p.kind == 3) {
                W.tower_seconds += ms * 1e-3; W.tower_launches += p.launches; W.tower_flops += p.flops;
                if (p.kind == 3) { W.full_seconds += ms * 1e-3; W.full_launches += p.launches; W.full_flops += p.flops; }
            } else if (p.kind == 2) { W.cluster_seconds += ms * 1e-3; W.cluster_launches += p.launches; W.cluster_flops += p.flops; }
            else { W.conv_seconds += ms * 1e-3; W.conv_launches += p.launches; W.conv_flops += p.flops; }
            static const int bounds[DIEE_BANDS - 1] = {16, 32, 64, 128, 256, 512, 928, 1024};      // DIEE_BANDS, include/diee.h
            int b = 0;
            while (b < (int)DIEE_BANDS - 1 && p.boards > bounds[b]) ++b;
            W.band_seconds[b] += ms * 1e-3; W.band_launches[b] += 1; W.band_flops[b] += p.flops; W.band_flops_demanded[b] += flops_dem;
        }
        W.free_events.push_back(p.a); W.free_events.push_back(p.b);
    }
    W.pending.clear();
    if (stats) {
        stats->conv_seconds = W.conv_seconds; stats->conv_launches = W.conv_launches; stats->conv_flops = W.conv_flops;
        stats->tower_seconds = W.tower_seconds; stats->tower_launches = W.tower_launches; stats->tower_flops = W.tower_flops;
        stats->cluster_seconds = W.cluster_seconds; stats->cluster_launches = W.cluster_launches; stats->cluster_flops = W.cluster_flops;
        stats->full_seconds = W.full_seconds; stats->full_launches = W.full_launches; stats->full_flops = W.full_flops;
        for (int b = 0; b < (int)DIEE_BANDS; ++b) { stats->band_seconds[b] = W.band_seconds[b]; stats->band_launches[b] = W.band_launches[b]; stats->band_flops[b] = W.band_flops[b]; stats->band_flops_demanded[b] = W.band_flops_demanded[b]; }
    }
}
void nn_reset_timing(Engine& e) {
    if (!e.net) return;
    e.net->conv_seconds = 0; e.net->conv_launches = 0; e.net->conv_flops = 0; e.net->forward_count = 0;
    e.net->tower_seconds = 0; e.net->tower_launches = 0; e.net->tower_flops = 0;
    e.net->cluster_seconds = 0; e.net->cluster_launches = 0; e.net->cluster_flops = 0;
    e.net->full_seconds = 0; e.net->full_launches = 0; e.net->full_flops = 0;
    for (int b = 0; b < (int)DIEE_BANDS; ++b) { e.net->band_seconds[b] = 0; e.net->band_launches[b] = 0; e.net->band_flops[b] = 0; e.net->band_flops_demanded[b] = 0; }
}

#ifdef DIEE_DEV_BUILD
// development probe: average device time of the tower conv kernel (modes 0 and 1) at batch G
void nn_conv_bench(Engine& e, int G, int variant, int reps, float* us_mode0, float* us_mode1, float* us_forward) {
    if (!e.net || !e.net->loaded) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    NetWeights& W = *e.net;
    nn_reserve(e, G);
    hipStream_t st = e.stream;
    const size_t M = (size_t)((G + 7) / 8 * 8) * 24;
    const int fill = getenv("DIEE_BENCH_ZERO") ? 0 : 0x3c;
    HIPCHK(hipMemsetAsync(W.actX.p, fill, M * 256 * 2, st));     // bf16 0x3c3c ~ 0.0115
    HIPCHK(hipMemsetAsync(W.actH.p, fill, M * 256 * 2, st));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    nn_set_conv_variant(variant);
    float ms = 0.f;
    for (int mode = 0; mode < 2; ++mode) {
        for (int r = -3; r < reps; ++r) {
            if (r == 0) HIPCHK(hipEventRecord(a, st));
            const int layer = 1 + ((r + 3) % 38);
            launch_conv3x3(st, 256, mode, mode ? W.actH.p : W.actX.p, W.wl(layer), W.bl(layer),
                           mode ? W.actX.p : nullptr, mode ? W.actX.p : W.actH.p, nullptr, G, 256);
        }
        HIPCHK(hipEventRecord(b, st));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        (mode ? *us_mode1 : *us_mode0) = ms * 1e3f / reps;
    }
    // whole forward (device-resident inputs)
    e.tmp_a.ensure((size_t)G * 32); e.tmp_b.ensure((size_t)G * 1352 * 4); e.tmp_c.ensure((size_t)G * 4);
    HIPCHK(hipMemsetAsync(e.tmp_a.p, 1, (size_t)G * 32, st));
    const int se = W.sample_every; W.sample_every = 0;
    const auto saved_table = W.tower_table;
    const auto saved_cl = W.cluster_table;
    if (variant >= 100 && variant <= 114) W.tower_table = {{0, variant - 100}};
    else if (variant != 0) W.tower_table.clear();
    if (variant == 201 || variant == 202 || variant == 204 || variant == 208) { W.cluster_table = {{1 << 30, variant - 200}}; nn_set_conv_variant(0); }   // cluster tower, 2 / 4 boards per group
    else if (variant != 0) W.cluster_table.clear();
    for (int r = -2; r < reps; ++r) {
        if (r == 0) HIPCHK(hipEventRecord(a, st));
        nn_forward(e, e.tmp_a.p, G, (float*)e.tmp_b.p, (float*)e.tmp_c.p);
    }
    W.sample_every = se; W.tower_table = saved_table; W.cluster_table = saved_cl;
    HIPCHK(hipEventRecord(b, st));
    HIPCHK(hipEventSynchronize(b));
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    *us_forward = ms * 1e3f / reps;
    if (getenv("DIEE_TOWER_CLOCK")) {      // diagnostic build (-DDIEE_TOWER_ABLATE=3): median in-kernel clock of the fused tower
        DevBuf<unsigned long long> dbg; dbg.ensure(4096);
        HIPCHK(hipMemsetAsync(dbg.p, 0, 4096 * 8, st));
        nn_set_tower_dbg(dbg.p);
        const auto saved2 = W.tower_table; W.tower_table = {{0, variant >= 100 ? variant - 100 : 5}};
        for (int r = 0; r < 200; ++r) nn_forward(e, e.tmp_a.p, G, (float*)e.tmp_b.p, (float*)e.tmp_c.p);
        W.tower_table = saved2; nn_set_tower_dbg(nullptr);
        std::vector<unsigned long long> h(4096);
        e.d2h(h.data(), dbg.p, (size_t)4096); e.sync();
        std::vector<double> mhz;
        for (int i = 0; i < 2048; ++i) if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty()) fprintf(stderr, "[diee] fused tower in-kernel clock: median %.0f MHz (min %.0f, max %.0f) over %zu workgroups; %llu shader cycles\n", mhz[mhz.size() / 2], mhz.front(), mhz.back(), mhz.size(), h[0]);
    }
    if (getenv("DIEE_CLUSTER_CLOCK") && (variant == 201 || variant == 202 || variant == 204 || variant == 208)) {   // diagnostic build: per-phase cycles of a cluster-tower layer
        DevBuf<unsigned long long> dbg; dbg.ensure(8192);
        HIPCHK(hipMemsetAsync(dbg.p, 0, 8192 * 8, st));
        nn_set_tower_dbg(dbg.p);
        const auto saved2 = W.cluster_table; W.cluster_table = {{1 << 30, variant - 200}};
        nn_forward(e, e.tmp_a.p, G, (float*)e.tmp_b.p, (float*)e.tmp_c.p);
        W.cluster_table = saved2; nn_set_tower_dbg(nullptr);
        std::vector<unsigned long long> h(8192);
        e.d2h(h.data(), dbg.p, (size_t)8192); e.sync();
        static const char* names[6] = {"poll tile in", "stage (barrier)", "MFMA loop (+ barrier)", "partials (barrier)", "reduce / epilogue + store", "end barrier"};
        double sum[6] = {0, 0, 0, 0, 0, 0}; int nwg = 0;
        for (int b2 = 0; b2 < 1024; ++b2) if (h[b2 * 8 + 2]) { ++nwg; for (int i = 0; i < 6; ++i) sum[i] += (double)h[b2 * 8 + i] / 35.0; }
        for (int i = 0; i < 6 && nwg; ++i) fprintf(stderr, "[diee] cluster tower G=%d: %-20s %8.0f cycles / layer (mean of %d workgroups)\n", G, names[i], sum[i] / nwg, nwg);
    }
    nn_set_conv_variant(0);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}
#else
void nn_conv_bench(Engine&, int, int, int, float*, float*, float*) {
    throw EngineError(DIEE_ERR_UNSUPPORTED, "diee_dev_conv_bench needs the development build (python die-e_amd/build.py --dev; load it through DIEE_LIB)");
}
#endif

void Engine::nn_forward_host(const diee_bg_state* states, uint32_t n, float* policy, float* value) {
    HIPCHK(hipSetDevice(device));
    if (!net || !net->loaded) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    if (!n) return;
    tmp_a.ensure((size_t)n * 32);
    tmp_b.ensure((size_t)n * 1352 * 4);
    tmp_c.ensure((size_t)n * 4);
    h2d(tmp_a.p, (const uint8_t*)states, (size_t)n * 32);
    const int se = net->sample_every; net->sample_every = 0;
    nn_forward(*this, tmp_a.p, (int)n, (float*)tmp_b.p, (float*)tmp_c.p);
    net->sample_every = se;
    d2h((uint8_t*)policy, tmp_b.p, (size_t)n * 1352 * 4);
    d2h((uint8_t*)value, tmp_c.p, (size_t)n * 4);
    sync();
    check_overflow();
}

}  // namespace diee
