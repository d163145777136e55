// Engine objects and the C ABI (include/whisper_mi355.h): blob parsing, weight residency and the
// kernel sequences of the three engines of the reference's Whisper example:
//   encoder        WhisperEncoder.forward   R/tensorrt_llm/models/whisper/model.py:149-172
//   cross K/V      CrossAttn_KV.forward     model.py:469-540
//   decoder step   WhisperDecoder.forward   model.py:241-299, block :61-122
// Everything is enqueued on the caller's stream into caller-owned buffers; nothing allocates or
// synchronises here (the contract of Session.run, session.py:148-178).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/whisper_mi355.h"
#include "common.h"
#include "kernels.h"

namespace wm {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int post_launch_check(hipStream_t s, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return 2; }
    static int sync_check = -1;
    if (sync_check < 0) { const char* v = getenv("WM_SYNC_CHECK"); sync_check = (v && v[0] == '1') ? 1 : 0; }
    if (sync_check) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;       // a synchronise would invalidate a capture in progress
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("kernel %s faulted: %s", what, hipGetErrorString(e)); return 2; }
    }
    return 0;
}

// ---- lab knobs (common.h) ------------------------------------------------------------------------
namespace {
std::mutex g_lab_mu;
std::string g_lab_honoured;                 // "NAME=value;..." of every knob honoured so far
std::map<std::string, int> g_lab_seen;      // 1: honoured and logged, 2: ignored and warned
}  // namespace
const char* lab_env_str(const char* name) {
    const char* v = getenv(name);
    if (!v) return nullptr;
    const char* lab = getenv("WM_LAB");
    const bool on = lab && lab[0] == '1';
    std::lock_guard<std::mutex> lk(g_lab_mu);
    int& seen = g_lab_seen[name];
    if (on && seen != 1) {
        fprintf(stderr, "whisper_mi355: lab knob %s=%s honoured (WM_LAB=1): results or timings may differ from the product's\n", name, v);
        g_lab_honoured += std::string(name) + "=" + v + ";";
        seen = 1;
    } else if (!on && seen == 0) {
        fprintf(stderr, "whisper_mi355: %s=%s is set but WM_LAB is not 1: ignored (lab knobs are for A/B runs only)\n", name, v);
        seen = 2;
    }
    return on ? v : nullptr;
}
int lab_env_int(const char* name, int dflt) {
    const char* v = lab_env_str(name);
    return v ? atoi(v) : dflt;
}

// ---- blob format (written by weight.py: write_engine_blob) ---------------------------------------
#pragma pack(push, 1)
struct BlobHeader {
    char magic[8];            // "WM355ENG"
    uint32_t version;         // 1
    uint32_t kind;            // WM_ENGINE_*
    uint32_t n_tensors;
    uint32_t flags;           // WM_FLAG_*
    int32_t dims[10];         // wm_dims order
    uint64_t data_offset;     // from blob start, 256-byte aligned
    uint64_t data_bytes;
};
struct BlobTensor {
    char name[64];
    uint32_t dtype;           // 0 f16, 1 i8, 2 f32, 3 i32, 4 packed int4 (two biased nibbles per byte)
    uint32_t ndim;
    uint64_t shape[4];
    uint64_t offset;          // from data_offset, 256-byte aligned
    uint64_t nbytes;
};
#pragma pack(pop)

struct Tensor { const unsigned char* ptr = nullptr; uint32_t dtype = 0; uint64_t shape[4] = {0, 0, 0, 0}; uint64_t nbytes = 0; };

struct Lin {                  // one Linear: weight (row-major or tile-linear), optional scale, bias
    const void* w = nullptr; const h16* s = nullptr; const h16* b = nullptr;
    int N = 0, K = 0, n_blocks = 0;
    int wcode = 0;            // weight encoding for the skinny GEMM: 0 fp16, 1 int8, 4 packed int4
};
struct EncLayer { const h16 *ln1g, *ln1b, *ln2g, *ln2b; Lin qkv, out, mlp1, mlp2; };
struct DecLayer {
    const h16 *ln1g, *ln1b, *lncg, *lncb, *ln2g, *ln2b;
    Lin qkv, out, cq, cout, mlp1, mlp2;
    float kv_scale = 1.f;
    float cross_scale = 0.f;      // > 0: int8 cross K/V (WM_FLAG_INT8_CROSS_KV)
};

}  // namespace wm

using namespace wm;

namespace wm {
// events that order a group's light stream against the shared heavy stream (wm_decoder_step_multi):
// [group][layer][0: q ready, 1: ctx ready].  Owned by the engine, i.e. created on the engine's device; the mutex
// makes concurrent calls on one engine (threads driving different streams) safe.
struct EventPool {
    std::mutex mu;
    std::vector<hipEvent_t> ev;
    hipEvent_t get(size_t idx) {
        std::lock_guard<std::mutex> lock(mu);
        while (ev.size() <= idx) {
            hipEvent_t x = nullptr;
            if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev.push_back(x);
        }
        return ev[idx];
    }
    ~EventPool() { for (hipEvent_t x : ev) (void)hipEventDestroy(x); }
};
}  // namespace wm

struct wm_engine {
    int kind = 0; uint32_t flags = 0; wm_dims dims{}; int device = 0;
    mutable wm::EventPool events;
    unsigned char* dev = nullptr; size_t dev_bytes = 0;
    std::map<std::string, Tensor> t;
    std::map<std::string, float> scalars;      // host copies of 4-byte fp32 tensors (kv scales)
    // encoder
    Lin conv1, conv2; const h16* enc_pos = nullptr; const h16 *lnpg = nullptr, *lnpb = nullptr;
    std::vector<EncLayer> enc;
    // cross
    std::vector<Lin> ckv; std::vector<float> ckv_scale;
    // decoder
    const void* emb_t = nullptr; int emb_blocks = 0; const h16 *lnfg = nullptr, *lnfb = nullptr;
    std::vector<DecLayer> dec;
    // the one-row chain's stage descriptors (gemv_chain.hip), six per layer in chain order: out, cq | cout, mlp1, mlp2, qkv of the
    // next layer -- on the device (the kernels read them there) and on the host (the launcher's checks)
    wm::ChainStage* chain_dev = nullptr; std::vector<wm::ChainStage> chain_host;      // [qkv of layer 0] + 6 per layer (gemv_chain.hip)
    wm::ChainLayerStatic* chain_lstat = nullptr;                                       // per layer: biases of the attention stages, cache scale
    // what the library knows about the chain state it keeps in a caller's workspace (granules, call counter, pointer table), keyed by the
    // table's address: the workspace_id under which the state was initialised, and the per-layer pointers last written to the table.
    // Dies with the engine; capped (a caller that hands a new workspace to every call must not grow it without bound).
    struct ChainWsSeen { uint64_t id = 0; std::vector<wm::ChainLayerIo> tab; };
    mutable std::mutex chain_io_mu;
    mutable std::unordered_map<const void*, ChainWsSeen> chain_io_seen;
    bool w8() const { return flags & WM_FLAG_WEIGHT_ONLY_INT8; }
    bool i8kv() const { return flags & WM_FLAG_INT8_KV; }
    bool i8cross() const { return flags & WM_FLAG_INT8_CROSS_KV; }
    int gelu() const { return (flags & WM_FLAG_GELU_TANH) ? 2 : 1; }
};

namespace {

int find(const wm_engine* e, const std::string& name, Tensor* out, bool required = true) {
    auto it = e->t.find(name);
    if (it == e->t.end()) {
        if (required) { set_error("engine blob has no tensor '%s'", name.c_str()); return 1; }
        *out = Tensor{};
        return 0;
    }
    *out = it->second;
    return 0;
}

// weight `base`.w (+ .s when weight-only) + optional `base`.b
int get_lin(const wm_engine* e, const std::string& base, bool tiled, bool quantisable, bool need_bias, Lin* l) {
    Tensor w, s, b;
    if (find(e, base + (tiled ? ".t" : ".w"), &w)) return 1;
    l->w = w.ptr;
    // row-major (non-tiled) matrices belong to the M >> 16 stages: a weight-only blob's int8 copies of them stay int8 at
    // rest (round 3) and are expanded per use into the caller's workspace (big())
    const bool q = quantisable && e->w8();
    const bool packed4 = w.dtype == 4;
    if ((w.dtype == 1 || packed4) != q || (packed4 && !tiled)) {
        set_error("tensor %s: dtype does not match the engine's weight-only flag", base.c_str());
        return 1;
    }
    l->wcode = packed4 ? 4 : (q ? 1 : 0);
    if (tiled) {           // shape = [n_blocks, k_tiles, 64, 16 bytes]
        l->n_blocks = (int)w.shape[0];
        l->N = l->n_blocks * 16;
        l->K = (int)w.shape[1] * (packed4 ? 128 : (q ? 64 : 32));
    } else {
        l->N = (int)w.shape[0]; l->K = (int)w.shape[1];
    }
    if (q) { if (find(e, base + ".s", &s)) return 1; l->s = (const h16*)s.ptr; }
    if (find(e, base + ".b", &b, need_bias)) return 1;
    l->b = (const h16*)b.ptr;
    return 0;
}

int get_vec(const wm_engine* e, const std::string& name, const h16** out) {
    Tensor t;
    if (find(e, name, &t)) return 1;
    *out = (const h16*)t.ptr;
    return 0;
}

int resolve(wm_engine* e) {
    const wm_dims& d = e->dims;
    char buf[96];
    if (e->kind == WM_ENGINE_ENCODER) {
        if (get_lin(e, "conv1", false, false, true, &e->conv1)) return 1;
        if (get_lin(e, "conv2", false, false, true, &e->conv2)) return 1;
        if (get_vec(e, "pos", &e->enc_pos)) return 1;
        if (get_vec(e, "ln_post.g", &e->lnpg) || get_vec(e, "ln_post.b", &e->lnpb)) return 1;
        e->enc.resize(d.n_audio_layer);
        for (int i = 0; i < d.n_audio_layer; ++i) {
            EncLayer& L = e->enc[i];
            snprintf(buf, sizeof(buf), "blocks.%d.", i);
            const std::string p(buf);
            if (get_vec(e, p + "attn_ln.g", &L.ln1g) || get_vec(e, p + "attn_ln.b", &L.ln1b) ||
                get_vec(e, p + "mlp_ln.g", &L.ln2g) || get_vec(e, p + "mlp_ln.b", &L.ln2b)) return 1;
            if (get_lin(e, p + "qkv", false, true, true, &L.qkv) || get_lin(e, p + "out", false, true, true, &L.out) ||
                get_lin(e, p + "mlp1", false, true, true, &L.mlp1) || get_lin(e, p + "mlp2", false, true, true, &L.mlp2)) return 1;
        }
    } else if (e->kind == WM_ENGINE_CROSS_KV) {
        e->ckv.resize(d.n_text_layer);
        for (int i = 0; i < d.n_text_layer; ++i) {
            snprintf(buf, sizeof(buf), "blocks.%d.kv", i);
            if (get_lin(e, buf, false, true, true, &e->ckv[i])) return 1;
            if (e->i8cross()) {
                snprintf(buf, sizeof(buf), "blocks.%d.cross_kv_scale", i);
                auto it = e->scalars.find(buf);
                if (it == e->scalars.end() || !(it->second > 0.f)) { set_error("int8 cross-K/V engine lacks a positive %s", buf); return 1; }
                e->ckv_scale.push_back(it->second);
            }
        }
    } else if (e->kind == WM_ENGINE_DECODER) {
        Tensor emb;
        if (find(e, "emb.t", &emb)) return 1;
        e->emb_t = emb.ptr; e->emb_blocks = (int)emb.shape[0];
        if (get_vec(e, "ln.g", &e->lnfg) || get_vec(e, "ln.b", &e->lnfb)) return 1;
        e->dec.resize(d.n_text_layer);
        for (int i = 0; i < d.n_text_layer; ++i) {
            DecLayer& L = e->dec[i];
            snprintf(buf, sizeof(buf), "blocks.%d.", i);
            const std::string p(buf);
            if (get_vec(e, p + "attn_ln.g", &L.ln1g) || get_vec(e, p + "attn_ln.b", &L.ln1b) ||
                get_vec(e, p + "cross_ln.g", &L.lncg) || get_vec(e, p + "cross_ln.b", &L.lncb) ||
                get_vec(e, p + "mlp_ln.g", &L.ln2g) || get_vec(e, p + "mlp_ln.b", &L.ln2b)) return 1;
            if (get_lin(e, p + "qkv", true, true, true, &L.qkv) || get_lin(e, p + "out", true, true, true, &L.out) ||
                get_lin(e, p + "cq", true, true, true, &L.cq) || get_lin(e, p + "cout", true, true, true, &L.cout) ||
                get_lin(e, p + "mlp1", true, true, true, &L.mlp1) || get_lin(e, p + "mlp2", true, true, true, &L.mlp2)) return 1;
            if (e->i8kv()) {
                auto it = e->scalars.find(p + "kv_scale");
                if (it == e->scalars.end()) { set_error("int8-KV engine lacks %skv_scale", p.c_str()); return 1; }
                L.kv_scale = it->second;
                if (!(L.kv_scale > 0.f)) { set_error("%skv_scale must be positive", p.c_str()); return 1; }
            }
            if (e->i8cross()) {
                auto it = e->scalars.find(p + "cross_kv_scale");
                if (it == e->scalars.end() || !(it->second > 0.f)) { set_error("int8 cross-K/V engine lacks a positive %scross_kv_scale", p.c_str()); return 1; }
                L.cross_scale = it->second;
            }
        }
    } else {
        set_error("unknown engine kind %d", e->kind);
        return 1;
    }
    return 0;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Device residency of a blob's tensors: one allocation, one copy, for every engine kind.  Weight-only ENCODER / CROSS-K/V
// engines keep their row-major int8 matrices and scales as stored (round 3; rounds 1-2 expanded them to fp16 at load, which
// left a weight-only engine with the memory of an fp16 one: 3.09 -> 2.35 GB where the reference's chart saves 1.7 GB,
// /root/reference/README.md:178-180).  Their GEMMs are MFMA-bound (M = 1500 x batch) and multiply fp16(fp16(q) * scale) -- the
// reference kernels' per-element dequantisation (fpA_intB_gemm_template.h:47-140 does it in registers) -- so each matrix is
// expanded right before its GEMM into a scratch block of the caller's workspace (big(): one pass over 1.6-6.6 MB, ~0.1 % of an
// encoder pass, the expansion staying in L2 / Infinity Cache for the GEMM behind it): same values, same GEMM, bit-identical.
int upload(wm_engine* e, const BlobHeader* h, const BlobTensor* tt, const unsigned char* data) {
    auto name_of = [&](uint32_t i) { char nm[65]; memcpy(nm, tt[i].name, 64); nm[64] = 0; return std::string(nm); };
    WM_CHECK_HIP(hipMalloc((void**)&e->dev, h->data_bytes ? h->data_bytes : 256));
    WM_CHECK_HIP(hipMemcpy(e->dev, data, h->data_bytes, hipMemcpyHostToDevice));
    e->dev_bytes = h->data_bytes;
    for (uint32_t i = 0; i < h->n_tensors; ++i) {
        Tensor t;
        t.ptr = e->dev + tt[i].offset; t.dtype = tt[i].dtype; t.nbytes = tt[i].nbytes;
        memcpy(t.shape, tt[i].shape, sizeof(t.shape));
        e->t[name_of(i)] = t;
    }
    return 0;
}

struct Carver {            // bump allocator over the caller's workspace
    unsigned char* base; size_t cap; size_t off = 0;
    template <typename T> T* take(size_t n) {
        off = align_up(off);
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// ---- encoder -------------------------------------------------------------------------------------
struct EncWs { h16 *melT, *c1, *x, *xn, *qkv, *ctx, *hid, *wq; size_t total; };
EncWs carve_encoder(const wm_engine* e, int B, void* ws) {
    const wm_dims& d = e->dims;
    const size_t T = d.n_audio_ctx, Tin = 2 * T, C = d.n_audio_state;
    Carver c{(unsigned char*)ws, 0};
    EncWs w;
    w.melT = c.take<h16>(B * (Tin + 2) * d.n_mels + 512);
    w.c1 = c.take<h16>(B * (Tin + 2) * C + 512);
    w.x = c.take<h16>(B * T * C);
    w.xn = c.take<h16>(B * T * C);
    w.qkv = c.take<h16>(B * T * 3 * C);
    w.ctx = c.take<h16>(B * T * C);
    w.hid = c.take<h16>(B * T * 4 * C);
    w.wq = e->w8() ? c.take<h16>(4 * C * C) : nullptr;        // fp16 expansion of the int8 matrix in use (the widest: 4C x C)
    w.total = align_up(c.off);
    return w;
}

int big(const Lin& l, const wm_engine* e, const h16* A, int lda, int M, h16* Cout, int ldc, int act,
        const h16* residual, int ldr, hipStream_t s, GemmBigParams* custom = nullptr, int max_wgs = 0, h16* wq = nullptr) {
    GemmBigParams p{};
    if (custom) p = *custom;
    const void* W = l.w;
    if (l.s) {             // int8 at rest: fp16(fp16(q) * scale) into the scratch block, then the fp16 MFMA GEMM
        WM_REQUIRE(wq, "big GEMM: int8 weights need a scratch block for their expansion");
        if (launch_dequant_w8((const int8_t*)l.w, l.s, wq, l.N, l.K, s)) return 2;
        W = wq;
    }
    p.A = A; p.lda = lda; p.M = M; p.K = l.K; p.W = W; p.N = l.N;
    p.bias = l.b; p.C = Cout; p.ldc = ldc; p.act = act;
    if (!custom) { p.residual = residual; p.ldr = ldr; }
    p.max_wgs = max_wgs;
    return launch_gemm_f16(p, s);
}

}  // namespace

extern "C" {

int wm_version(void) { return WM_ABI_VERSION; }
const char* wm_last_error(void) { return g_err; }
int wm_device_count(int* out) {
    WM_CHECK_HIP(hipGetDeviceCount(out));
    return 0;
}

int wm_engine_create(const void* blob, size_t nbytes, int device, wm_engine** out) {
    WM_REQUIRE(blob && out, "wm_engine_create: null argument");
    WM_REQUIRE(nbytes >= sizeof(BlobHeader), "engine blob too small (%zu bytes)", nbytes);
    const BlobHeader* h = (const BlobHeader*)blob;
    WM_REQUIRE(memcmp(h->magic, "WM355ENG", 8) == 0, "not a whisper_mi355 engine blob (bad magic)");
    WM_REQUIRE(h->version == 1, "unsupported engine blob version %u", h->version);
    WM_REQUIRE(sizeof(BlobHeader) + (size_t)h->n_tensors * sizeof(BlobTensor) <= h->data_offset &&
                   h->data_offset + h->data_bytes <= nbytes, "engine blob is truncated");
    wm_engine* e = new wm_engine();
    e->kind = (int)h->kind; e->flags = h->flags; e->device = device;
    memcpy(&e->dims, h->dims, sizeof(wm_dims));
    const BlobTensor* tt = (const BlobTensor*)((const unsigned char*)blob + sizeof(BlobHeader));
    for (uint32_t i = 0; i < h->n_tensors; ++i)
        if (tt[i].offset + tt[i].nbytes > h->data_bytes) {
            set_error("tensor %.64s exceeds the blob", tt[i].name);
            delete e; return 1;
        }
    hipError_t err = hipSetDevice(device);
    if (err != hipSuccess) { set_error("wm_engine_create: device %d: %s", device, hipGetErrorString(err)); delete e; return 2; }
    if (int rc = upload(e, h, tt, (const unsigned char*)blob + h->data_offset)) {
        if (e->dev) (void)hipFree(e->dev);
        delete e;
        return rc;
    }
    for (uint32_t i = 0; i < h->n_tensors; ++i)
        if (tt[i].dtype == 2 && tt[i].nbytes == 4) {
            char nm[65]; memcpy(nm, tt[i].name, 64); nm[64] = 0;
            float v;
            memcpy(&v, (const unsigned char*)blob + h->data_offset + tt[i].offset, 4);
            e->scalars[nm] = v;
        }
    if (resolve(e)) { (void)hipFree(e->dev); delete e; return 1; }
    if (e->kind == WM_ENGINE_DECODER && !e->dec.empty()) {
        const int n = (int)e->dec.size();
        e->chain_host.resize((size_t)6 * n + 1);
        auto fill = [](wm::ChainStage& st, const Lin& l, const h16* g, const h16* b, int mode) {
            st.Wt = l.w; st.scale = l.s; st.bias = l.b; st.ln_g = g; st.ln_b = b; st.K = l.K; st.n_blocks = l.n_blocks; st.mode = mode; st.pad_ = 0;
        };
        for (int i = 0; i < n; ++i) {
            const DecLayer& L = e->dec[i];
            wm::ChainStage* st = &e->chain_host[(size_t)6 * i + 1];
            fill(st[0], L.out, nullptr, nullptr, 2);
            fill(st[1], L.cq, L.lncg, L.lncb, 0);
            fill(st[2], L.cout, nullptr, nullptr, 2);
            fill(st[3], L.mlp1, L.ln2g, L.ln2b, 1);
            fill(st[4], L.mlp2, nullptr, nullptr, 2);
            if (i + 1 < n) fill(st[5], e->dec[i + 1].qkv, e->dec[i + 1].ln1g, e->dec[i + 1].ln1b, 0);
            else st[5] = wm::ChainStage{};
        }
        fill(e->chain_host[0], e->dec[0].qkv, e->dec[0].ln1g, e->dec[0].ln1b, 0);
        std::vector<wm::ChainLayerStatic> lstat((size_t)n);
        for (int i = 0; i < n; ++i) lstat[i] = wm::ChainLayerStatic{e->dec[i].qkv.b, e->dec[i].cq.b, e->dec[i].kv_scale, 0};
        const size_t bytes = e->chain_host.size() * sizeof(wm::ChainStage), lbytes = lstat.size() * sizeof(wm::ChainLayerStatic);
        if (hipMalloc((void**)&e->chain_dev, bytes) != hipSuccess || hipMalloc((void**)&e->chain_lstat, lbytes) != hipSuccess ||
            hipMemcpy(e->chain_dev, e->chain_host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e->chain_lstat, lstat.data(), lbytes, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("wm_engine_create: the decode chain's descriptor tables could not be placed on the device");
            if (e->chain_dev) (void)hipFree(e->chain_dev);
            if (e->chain_lstat) (void)hipFree(e->chain_lstat);
            (void)hipFree(e->dev); delete e; return 2;
        }
    }
    *out = e;
    return 0;
}

void wm_engine_destroy(wm_engine* e) {
    if (!e) return;
    if (e->dev) { (void)hipSetDevice(e->device); (void)hipFree(e->dev); }
    if (e->chain_dev) (void)hipFree(e->chain_dev);
    if (e->chain_lstat) (void)hipFree(e->chain_lstat);
    delete e;
}

int wm_engine_info(const wm_engine* e, int32_t* kind, uint32_t* flags, wm_dims* dims) {
    WM_REQUIRE(e, "wm_engine_info: null engine");
    if (kind) *kind = e->kind;
    if (flags) *flags = e->flags;
    if (dims) *dims = e->dims;
    return 0;
}
size_t wm_engine_weight_bytes(const wm_engine* e) { return e ? e->dev_bytes : 0; }

// ================================================================================================ encoder
size_t wm_encoder_workspace_bytes(const wm_engine* e, int batch) {
    if (!e || e->kind != WM_ENGINE_ENCODER || batch < 1) return 0;
    return carve_encoder(e, batch, nullptr).total;
}

int wm_encoder_forward(const wm_engine* e, const void* mel, int B, void* out, void* workspace,
                       size_t workspace_bytes, wm_stream_t stream_) {
    return wm_encoder_forward_shared(e, mel, B, out, workspace, workspace_bytes, 0, stream_);
}

int wm_encoder_forward_shared(const wm_engine* e, const void* mel, int B, void* out, void* workspace,
                              size_t workspace_bytes, int cu_budget, wm_stream_t stream_) {
    return wm_encoder_forward_range(e, mel, B, out, workspace, workspace_bytes, cu_budget, 0, e ? e->dims.n_audio_layer : 0, stream_);
}

// Layers [layer_begin, layer_end) of the encoder pass: the convolutions belong to a range that starts at layer 0, the final LayerNorm
// (which writes `out`) to one that ends at n_audio_layer; between the calls of one pass the residual stream lives in the workspace.
int wm_encoder_forward_range(const wm_engine* e, const void* mel, int B, void* out, void* workspace,
                             size_t workspace_bytes, int cu_budget, int layer_begin, int layer_end, wm_stream_t stream_) {
    WM_REQUIRE(e && e->kind == WM_ENGINE_ENCODER, "wm_encoder_forward: not an encoder engine");
    WM_REQUIRE(mel && out && workspace && B >= 1, "wm_encoder_forward: null argument or empty batch");
    WM_REQUIRE(cu_budget >= 0, "wm_encoder_forward_shared: cu_budget=%d", cu_budget);
    hipStream_t s = (hipStream_t)stream_;
    const wm_dims& d = e->dims;
    WM_REQUIRE(layer_begin >= 0 && layer_begin <= layer_end && layer_end <= d.n_audio_layer, "wm_encoder_forward_range: layers [%d, %d) of %d",
               layer_begin, layer_end, d.n_audio_layer);
    const int T = d.n_audio_ctx, Tin = 2 * T, C = d.n_audio_state, H = d.n_audio_head, M = B * T;
    WM_REQUIRE(C == H * 64, "head size must be 64 (n_state %d, heads %d)", C, H);
    EncWs w = carve_encoder(e, B, workspace);
    WM_REQUIRE(workspace_bytes >= w.total, "encoder workspace too small: %zu < %zu", workspace_bytes, w.total);

    if (layer_begin == 0) {
        // mel -> token-major, zero padded; slack after the last row is read by the K=256 view (x 0 weights)
        WM_CHECK_HIP(hipMemsetAsync(w.melT + (size_t)B * (Tin + 2) * d.n_mels, 0, 512 * sizeof(h16), s));
        if (launch_mel_transpose_pad((const h16*)mel, B, d.n_mels, Tin, w.melT, s)) return 2;
        // conv1 (k3 s1 p1) + GELU as a GEMM over a strided view: row t = padded rows t, t+1, t+2
        {
            GemmBigParams p{};
            p.a_rows = Tin; p.a_bstride = (long)(Tin + 2) * d.n_mels;
            p.c_rows = Tin; p.c_bstride = (long)(Tin + 2) * C;
            WM_REQUIRE(e->conv1.K >= 3 * d.n_mels, "conv1 weight K=%d < 3*n_mels", e->conv1.K);
            if (big(e->conv1, e, w.melT, d.n_mels, B * Tin, w.c1 + C, C, e->gelu(), nullptr, 0, s, &p, cu_budget)) return 2;
            if (launch_zero_pad_rows(w.c1, B, Tin + 2, C, s)) return 2;
        }
        // conv2 (k3 s2 p1) + GELU + positional embedding: row t = padded rows 2t, 2t+1, 2t+2
        {
            GemmBigParams p{};
            p.a_rows = T; p.a_bstride = (long)(Tin + 2) * C;
            p.residual = e->enc_pos; p.ldr = C; p.res_mod = T;
            if (big(e->conv2, e, w.c1, 2 * C, M, w.x, C, e->gelu(), nullptr, 0, s, &p, cu_budget)) return 2;
        }
    }
    const float qk_scale = 0.35355339059327373f;    // 64^-0.25
    for (int i = layer_begin; i < layer_end; ++i) {
        const EncLayer& L = e->enc[i];
        if (launch_layernorm(w.x, C, M, C, L.ln1g, L.ln1b, w.xn, C, s)) return 2;
        {
            GemmBigParams p{};
            p.colscale_n = 2 * C; p.colscale = qk_scale;
            if (big(L.qkv, e, w.xn, C, M, w.qkv, 3 * C, 0, nullptr, 0, s, &p, cu_budget, w.wq)) return 2;
        }
        AttnEncParams ap{w.qkv, 3 * C, B, T, H, w.ctx, C, 2 * cu_budget};      // two workgroups fill a CU's registers
        if (launch_attn_encoder(ap, s)) return 2;
        if (big(L.out, e, w.ctx, C, M, w.x, C, 0, w.x, C, s, nullptr, cu_budget, w.wq)) return 2;
        if (launch_layernorm(w.x, C, M, C, L.ln2g, L.ln2b, w.xn, C, s)) return 2;
        if (big(L.mlp1, e, w.xn, C, M, w.hid, 4 * C, e->gelu(), nullptr, 0, s, nullptr, cu_budget, w.wq)) return 2;
        if (big(L.mlp2, e, w.hid, 4 * C, M, w.x, C, 0, w.x, C, s, nullptr, cu_budget, w.wq)) return 2;
    }
    if (layer_end == d.n_audio_layer)
        if (launch_layernorm(w.x, C, M, C, e->lnpg, e->lnpb, (h16*)out, C, s)) return 2;
    return 0;
}

// ================================================================================================ cross K/V
size_t wm_cross_kv_workspace_bytes(const wm_engine* e, int batch) {
    (void)batch;
    if (!e || e->kind != WM_ENGINE_CROSS_KV) return 0;
    size_t widest = 0;                                       // fp16 expansion of one layer's int8 [2C][C] matrix
    for (const Lin& l : e->ckv) if (l.s) widest = widest > (size_t)l.N * l.K ? widest : (size_t)l.N * l.K;
    return align_up(256 + widest * sizeof(h16));
}

int wm_cross_kv(const wm_engine* e, const void* xa, int B, void* const* out_layers, void* workspace,
                size_t workspace_bytes, wm_stream_t stream_) {
    WM_REQUIRE(e && e->kind == WM_ENGINE_CROSS_KV, "wm_cross_kv: not a cross-attention K/V engine");
    WM_REQUIRE(workspace_bytes >= wm_cross_kv_workspace_bytes(e, B) && (workspace || !e->w8()), "wm_cross_kv: workspace too small: %zu < %zu",
               workspace_bytes, wm_cross_kv_workspace_bytes(e, B));
    WM_REQUIRE(xa && out_layers && B >= 1, "wm_cross_kv: null argument or empty batch");
    hipStream_t s = (hipStream_t)stream_;
    const wm_dims& d = e->dims;
    const int T = d.n_audio_ctx, C = d.n_text_state, H = d.n_text_head;
    for (int i = 0; i < d.n_text_layer; ++i) {
        WM_REQUIRE(out_layers[i], "wm_cross_kv: output %d is null", i);
        GemmBigParams p{};
        p.out_mode = 1; p.hs_T = T; p.hs_H = H; p.hs_kv = -1;
        if (e->i8cross()) p.q8_inv_scale = 1.0f / e->ckv_scale[i];     // out_layers[i] is int8 [B,2,H,T,64]
        if (big(e->ckv[i], e, (const h16*)xa, C, B * T, (h16*)out_layers[i], 0, 0, nullptr, 0, s, &p, 0, (h16*)workspace)) return 2;
    }
    return 0;
}

// ================================================================================================ decoder
namespace {
// ---- in-situ kernel timing for the roofline report (bench.py): HIP event pairs around sampled
// launches of the decode cross-attention kernel, on the stream it is launched on ------------------
// One sampler and one diagnostic timeline PER DEVICE (one process may drive several GPUs, each from its own
// thread): wm_profile_* / wm_debug_timeline act on the calling thread's current device, the decode step on its
// engine's device; the mutex covers slot allocation.
constexpr int MAX_DEVICES = 64;
struct Profiler {
    bool enabled = false; int layer_stride = 1;
    std::vector<hipEvent_t> start, stop; size_t used = 0;
    long long* timeline = nullptr; int timeline_cap = 0;
};
std::mutex g_prof_mu;
Profiler g_prof_dev[MAX_DEVICES];
int current_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    return dev;
}

// a free sample slot for an eager launch of layer `layer`, or -1; the launch itself stamps the slot's events
int prof_slot(Profiler& pr, int layer, hipStream_t s) {
    if (!pr.enabled) return -1;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (layer % pr.layer_stride != 0 || pr.used >= pr.start.size()) return -1;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return -1;   // eager launches only
    return (int)pr.used++;
}

struct DecWs { h16 *x, *xn, *ctx, *hid; float* part; float* cross_ws; size_t part_elems; int nsplit; size_t total;
               unsigned long long *gran_x, *gran_h, *gran_q, *gran_c, *gran_p, *gran_s; wm::ChainLayerIo* layer_io; unsigned* generation; };      // granule edges of the one-row chain (gemv_chain.hip) and its call counter

int cross_nsplit(int B, int H) {
    // Pieces the key range of the decode cross-attention is cut into (one workgroup per (utterance, head, piece), partial
    // softmaxes merged by attn_cross_combine_kernel).  TWO classes only, by the number of (utterance, head) pairs of the group:
    // up to 160 pairs (8 utterances of large-v2; round 5: 8 included, so that the one-launch step of up to eight rows has ONE form of the
    // cross-attention to reproduce -- r3au below: 2.15 / 2.13 ms with 1 / 4 pieces at 8 utterances) -> 4 pieces, otherwise the exact single pass without a merge launch.
    // Within a class a row's result does not depend on the batch it is in.  Rounds 1-3 used ceil(512 / pairs) <= 8 pieces
    // ("two workgroups per CU"): 8 different counts below 26 utterances, i.e. 8 different roundings of the merged softmax,
    // and no faster -- forced counts, token step in ms (profiles/r3au_cross_nsplit_forced.txt): 1 utterance 1.74 /
    // 1.65 / 1.58 / 1.58 with 1 / 2 / 4 / 8 pieces, 4 utterances 2.00 / 1.92 / 1.89 / 1.88, 6: 2.09 / 2.04 / 2.01 / 2.06,
    // 8: 2.15 / 2.16 / 2.13 (4), 12: 2.33 / 2.37 / 2.38 (1 / 2 / 3), 2 x 12: 2.86 / 2.91 / 3.01, 2 x 16: 3.41 / 3.44 / 3.47.
    // WM_CROSS_NSPLIT=n forces a count (A/B runs), -1 = 8 pieces below 512 pairs, -2 = the old rule.
    static const int forced = lab_env_int("WM_CROSS_NSPLIT", 0);
    if (forced == -1) return B * H >= 512 ? 1 : 8;
    if (forced == -2) { int n = (512 + B * H - 1) / (B * H); return n < 1 ? 1 : (n > 8 ? 8 : n); }
    if (forced > 0) return forced > 8 ? 8 : forced;
    return B * H <= 160 ? 4 : 1;
}

DecWs carve_decoder(const wm_engine* e, int B, int L, void* ws) {
    const wm_dims& d = e->dims;
    const size_t C = d.n_text_state, M = (size_t)B * L;
    Carver c{(unsigned char*)ws, 0};
    DecWs w;
    w.x = c.take<h16>(M * C); w.xn = c.take<h16>(M * C); w.ctx = c.take<h16>(M * C); w.hid = c.take<h16>(M * 4 * C);
    // split-K slabs: the widest product is ksplit * N over the six Linears; ksplit <= 24 by construction
    size_t widest = 0;
    const int Mc = (int)(M < SKINNY_MAX_M ? M : SKINNY_MAX_M);
    const int q = e->dec.empty() ? (e->w8() ? 1 : 0) : e->dec[0].qkv.wcode;     // every Linear of an engine shares one encoding
    const int Ns[6] = {(int)(3 * C), (int)C, (int)C, (int)C, (int)(4 * C), (int)C};
    const int Ks[6] = {(int)C, (int)C, (int)C, (int)C, (int)C, (int)(4 * C)};
    for (int i = 0; i < 6; ++i) {
        const int s = skinny_default_ksplit(Mc, Ks[i], Ns[i] / 16, q);
        widest = widest > (size_t)s * Ns[i] ? widest : (size_t)s * Ns[i];
    }
    w.part_elems = widest * M;
    w.part = c.take<float>(w.part_elems);
    w.nsplit = cross_nsplit(B, d.n_text_head);
    w.cross_ws = c.take<float>((size_t)B * d.n_text_head * w.nsplit * L * 66);
    const size_t R = M < (size_t)CHAIN_MAX_ROWS ? M : (size_t)CHAIN_MAX_ROWS;      // rows the one-launch step serves (gemv_chain.hip): every edge [rows][...]
    w.gran_x = c.take<unsigned long long>(R * (C / 2) + 8);
    w.gran_h = c.take<unsigned long long>(R * 2 * C + 8);
    w.gran_q = c.take<unsigned long long>(R * C + 8);
    w.gran_c = c.take<unsigned long long>(R * (C / 2) + 8);
    w.gran_p = c.take<unsigned long long>(R * d.n_text_head * 66 * 4 + 8);
    w.gran_s = c.take<unsigned long long>(R * 3 * C + 8);
    w.layer_io = c.take<wm::ChainLayerIo>((size_t)d.n_text_layer);
    w.generation = c.take<unsigned>(4);
    w.total = align_up(c.off);
    return w;
}

// skinny GEMM over all M rows in chunks of SKINNY_MAX_M; slabs laid out [ksplit][M_total][ldp]
int skinny_all(const Lin& l, const h16* A, int lda, int M, float* part, int* ksplit_out, hipStream_t s) {
    const int Mc = M < SKINNY_MAX_M ? M : SKINNY_MAX_M;
    const int ks = skinny_default_ksplit(Mc, l.K, l.n_blocks, l.wcode);
    for (int r0 = 0; r0 < M; r0 += SKINNY_MAX_M) {
        GemmSkinnyParams p{};
        p.A = A + (size_t)r0 * lda; p.lda = lda; p.M = (M - r0) < SKINNY_MAX_M ? (M - r0) : SKINNY_MAX_M; p.K = l.K;
        p.Wt = l.w; p.n_blocks = l.n_blocks; p.w8 = l.wcode; p.scale = l.s; p.ksplit = ks;
        p.part = part + (size_t)r0 * l.N; p.part_sstride = (long)M * l.N;
        if (launch_gemm_skinny(p, s)) return 2;
    }
    *ksplit_out = ks;
    return 0;
}
}  // namespace

namespace { constexpr int DEC_CHUNK = 4; }      // query tokens per pass of the decoder kernels (attn_decode.hip: MAX_L)

// n_new <= 4: one pass.  Longer token blocks (prompts / prefixes, W/decoding.py:485-513) are run as 4-token
// passes over the growing cache; their logits pass through a [batch, 4, n_vocab] staging block at the end of the
// workspace and are copied into the caller's [batch, n_new, n_vocab] tensor.
size_t wm_decoder_workspace_bytes(const wm_engine* e, int batch, int n_new) {
    if (!e || e->kind != WM_ENGINE_DECODER || batch < 1 || n_new < 1) return 0;
    if (n_new <= DEC_CHUNK) return carve_decoder(e, batch, n_new, nullptr).total;
    return carve_decoder(e, batch, DEC_CHUNK, nullptr).total + align_up((size_t)batch * DEC_CHUNK * e->dims.n_vocab * sizeof(h16));
}

namespace {
// One utterance group's decode step, cut into the phases between which the cross-attention kernel sits, so
// that several groups can be interleaved layer by layer (wm_decoder_step_multi).
// diagnostic timeline (wm_debug_timeline): Profiler::timeline of the engine's device
__global__ void stamp_kernel(long long* tl, int cap, long long tag, long long what) {
    if (threadIdx.x != 0) return;
    const int i = atomicAdd((int*)tl, 1);
    if (i < cap) { tl[1 + 3 * i] = tag; tl[2 + 3 * i] = what; tl[3 + 3 * i] = wall_clock64(); }
}

// Round 2 put the switch at 8 rows (8 rows 2.43 vs 2.56 ms per token, 2 x 8 rows 2.79 vs 2.92; 2 x 16 rows 3.70 vs 3.52 the other way).  Re-measured
// at the end of round 3 (four-wave self-attention on this side of the switch, graphs replayed node by node; profiles/r3an_small_path_switch.txt):
// 12 rows 2.38 vs 2.60 ms on the split-K chain, 2 x 12 rows 3.00 vs 3.13, 16 rows 2.68 vs 2.79, 2 x 16 rows 3.43 either way -> 16 rows (one MFMA row tile).
constexpr int SMALL_PATH_DEFAULT_ROWS = 16;
// Small batches (M = B * L <= SMALL_PATH_DEFAULT_ROWS rows) take the fused path of gemv_small.hip: every Linear is ONE launch
// that also applies the LayerNorm of its input rows and its own epilogue (bias / GELU / residual) -- 8 launches per layer
// instead of 12, no fp32 slabs, no row kernels.  WM_SMALL_PATH=<rows> moves the switch (0: the big-batch kernels for every size).
// (For more rows the same fusion was built into gemm_skinny.hip and measured: without a K split its 10-40 workgroups each
// stage the whole activation block, 70 us per launch at M = 192 against 16 + 10 for the split GEMM + row kernel, and the
// decode step at B = 576 went from 26.1 to 30.3 ms -- profiles/r2d_b576_fused_skinny_ks1_kernel_stats.csv.  Not kept.)
// exact V-row skipping of the decode cross-attention (attn_decode.hip, SKIP): on unless switched off (wm_set_cross_v_skip)
std::atomic<int> g_cross_v_skip{CROSS_V_SKIP_DEFAULT};
// the one-row chain (gemv_chain.hip): a decoder layer at batch 1 in 5 launches instead of 9 (wm_set_decode_chain)
std::atomic<int> g_decode_chain{DECODE_CHAIN_DEFAULT};
// What the one-launch forms need to know about a DEVICE.  `err` is the word a wave of a chain sets when it gives up a bounded wait (a
// workgroup of the launch was not running): it lives in pinned host memory that the device reaches directly, so the host reads it
// without a copy or a synchronisation -- every decoder call looks at it.  `declined` is sticky: once a launch has given up, the device
// is not trusted to hold a chain's workgroups together any more and every later call takes the launch-per-kernel path
// (wm_set_decode_chain re-arms).  `resident` caches the occupancy verdict per kernel variant and LDS footprint.
struct ChainDev {
    std::mutex mu;
    int n_cu = 0;
    unsigned* err_host = nullptr; unsigned* err_dev = nullptr;
    bool declined = false; std::string reason, footprint;
    std::map<long long, bool> resident;
    std::atomic<long long> launches{0};
    std::atomic<long long> declined_calls{0};
};
ChainDev g_chain_dev[64];
static ChainDev& chain_dev_slot(int device) { return g_chain_dev[device >= 0 && device < 64 ? device : 0]; }
// first use on a device: the CU count and the error word (the calling thread's current device is the engine's)
int chain_dev_init(ChainDev& cd, int device) {
    std::lock_guard<std::mutex> lk(cd.mu);
    if (cd.n_cu > 0 && cd.err_host) return 0;
    int v = 0;
    WM_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device));
    void* host = nullptr; void* dev = nullptr;
    WM_CHECK_HIP(hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(host, 0, 64);
    WM_CHECK_HIP(hipHostGetDevicePointer(&dev, host, 0));
    cd.err_host = (unsigned*)host; cd.err_dev = (unsigned*)dev;
    cd.n_cu = v > 0 ? v : 1;
    return 0;
}
inline unsigned chain_err_peek(const ChainDev& cd) {
    return cd.err_host ? __atomic_load_n(cd.err_host, __ATOMIC_RELAXED) : 0u;
}
// a stream created with a CU mask (wm_stream_create_cu_mask, hipExtStreamCreateWithCUMask) that leaves it fewer CUs than the device
// has cannot hold a chain's workgroups together.  A query that fails says nothing: the stream is taken for an ordinary one.
bool stream_has_all_cus(hipStream_t s, int n_cu) {
    uint32_t mask[32];
    memset(mask, 0, sizeof(mask));
    if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) { (void)hipGetLastError(); return true; }
    int bits = 0;
    for (uint32_t m : mask) bits += __builtin_popcount(m);
    return bits == 0 || bits >= n_cu;        // (no bit at all: the runtime reports no mask)
}
std::atomic<int> g_small_rows{-1};        // -1: not yet read from the environment
int small_path_max_rows() {               // WM_SMALL_PATH=<rows> / wm_set_small_batch_rows: the fused path serves M <= rows (0: never)
    int r = g_small_rows.load(std::memory_order_relaxed);
    if (r < 0) {
        r = lab_env_int("WM_SMALL_PATH", SMALL_PATH_DEFAULT_ROWS);
        r = r < 0 ? 0 : (r > GEMV_SMALL_MAX_M ? GEMV_SMALL_MAX_M : r);
        g_small_rows.store(r, std::memory_order_relaxed);
    }
    return r;
}

// Groups of at least rows_path_min_rows() rows (above the small-batch switch) take the row-split form of the same fused Linears
// (gemm_rows.hip) for every projection whose input is n_state wide: 10 launches per layer instead of 12, fp32 slabs only
// behind the MLP's second Linear.  Default 40 rows (WM_ROWS_MIN; WM_ROWS_PATH=0 or wm_set_rows_path(0): never -- the split-K
// chain, gemm_skinny + row kernel, for every Linear as in rounds 1-2).  Measured per token step, large-v2 int8, split-K chain vs
// row-split form (profiles/r3k_batch_sweep.txt): 12 rows 2.81 / 2.99 ms, 2 x 16 rows 3.55 / 3.69, 2 x 32 rows 4.69 / 5.09,
// 3 x 43 rows 7.42 / 7.21, 3 x 192 rows 25.2 / 23.4 -- below ~40 rows a group's step is a chain of latencies and the split-K
// GEMM + row kernel pair (two short launches on 40-160 workgroups) is the quicker link; above, the hand-overs' bytes decide.
std::atomic<int> g_rows_min{-1};          // -1: not yet read from the environment; 0: never
int rows_path_min_rows() {
    int r = g_rows_min.load(std::memory_order_relaxed);
    if (r < 0) {
        r = lab_env_int("WM_ROWS_PATH", 1) == 0 ? 0 : lab_env_int("WM_ROWS_MIN", 40);
        if (r < 0) r = 0;
        g_rows_min.store(r, std::memory_order_relaxed);
    }
    return r;
}

// Waves per (b, h) of the decode self-attention: 0 = by size (the small-batch side of the switch above runs the four-wave
// workgroup form, attn_decode.hip), 1 / 4 = forced (WM_SELF_WAVES, wm_set_self_attn_waves).
std::atomic<int> g_self_waves{-1};
int self_attn_waves(int rows) {
    int w = g_self_waves.load(std::memory_order_relaxed);
    if (w < 0) {
        w = lab_env_int("WM_SELF_WAVES", 0);
        if (w != 1 && w != 4) w = 0;
        g_self_waves.store(w, std::memory_order_relaxed);
    }
    return w ? w : (rows <= small_path_max_rows() ? 4 : 1);
}

// rows (utterances) of a group the one-launch step serves: CHAIN_MAX_ROWS (8) unless WM_CHAIN_ROWS=n keeps it to fewer (lab: A/B runs)
int chain_max_rows() {
    static const int r = [] { const int v = lab_env_int("WM_CHAIN_ROWS", CHAIN_MAX_ROWS); return v < 1 ? 1 : (v > CHAIN_MAX_ROWS ? CHAIN_MAX_ROWS : v); }();
    return r;
}

struct GroupStep {
    const wm_engine* e; const wm_decoder_io* io; DecWs w;
    int B, L, T, C, H, M;
    bool small = false;                          // the fused small-batch path (gemv_small.hip)
    bool chain = false;                          // ... with its Linears chained inside one launch (one row: gemv_chain.hip)
    int chain_wgs = 0; unsigned* chain_err = nullptr; ChainDev* chain_cd = nullptr;
    bool rows = false;                           // the fused row-split path (gemm_rows.hip)
    bool fused() const { return small || rows; }

    // one Linear of the small-batch path.  mode as epilogue.h; ln_g != null: LayerNorm of the input rows (the residual stream)
    // inside the kernel
    int gemv(const Lin& l, const h16* A, int lda, int mode, const h16* ln_g, const h16* ln_b, h16* out16, int ld16, hipStream_t s) {
        GemvSmallParams p{};
        p.A = A; p.lda = lda; p.M = M; p.K = l.K; p.Wt = l.w; p.n_blocks = l.n_blocks; p.w8 = l.wcode; p.scale = l.s;
        p.ln_g = ln_g; p.ln_b = ln_b;                // the kernel normalises the (few) input rows itself
        p.mode = mode; p.bias = l.b; p.gelu_kind = e->gelu();
        p.out32 = w.part; p.ld32 = l.N;
        p.out16 = out16; p.ld16 = ld16; p.n_valid = l.N;
        p.x = w.x; p.ldx = C;
        return small ? launch_gemv_small(p, s) : launch_gemm_rows(p, s);
    }

    Profiler* prof = nullptr;

    // diagnostic (WM_TIMELINE_FINE=1 with wm_debug_timeline): a stamp behind EVERY kernel of the chain, code 1000 + 32 * layer + position
    // (scripts/chain_probe.py turns them into in-situ durations per chain position); the K/V launch keeps its own pair of stamps
    void mark(int layer, int pos, hipStream_t s) {
        static const bool fine = lab_env_int("WM_TIMELINE_FINE", 0) == 1;
        if (fine && prof->timeline)
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, prof->timeline, prof->timeline_cap, (long long)(uintptr_t)io->logits, (long long)(1000 + 32 * layer + pos));
    }

    // alone: no other group's step is issued beside this one.  The one-launch forms need their 256 workgroups resident TOGETHER (one
    // fills a CU's LDS); two such launches dispatched side by side on two streams could each get half of the chip and wait for the
    // other half until the bounded waits give up -- so only a step that runs alone takes them.
    int init(const wm_engine* e_, const wm_decoder_io* io_, hipStream_t stream, bool alone = true) {
        e = e_; io = io_;
        prof = &g_prof_dev[e->device >= 0 && e->device < MAX_DEVICES ? e->device : 0];
        WM_REQUIRE(io && io->tokens && io->positional_embedding && io->present && io->cross && io->logits && io->workspace,
                   "wm_decoder_step: null argument");
        const wm_dims& d = e->dims;
        B = io->batch; L = io->n_new; T = io->n_past; C = d.n_text_state; H = d.n_text_head; M = B * L;
        WM_REQUIRE(B >= 1 && L >= 1 && L <= 4 && T >= 0, "wm_decoder_step: bad batch/n_new/n_past (%d, %d, %d)", B, L, T);
        WM_REQUIRE(T + L <= d.n_text_ctx, "wm_decoder_step: T+L=%d exceeds n_text_ctx=%d", T + L, d.n_text_ctx);
        WM_REQUIRE(T == 0 || io->past, "wm_decoder_step: n_past > 0 needs past buffers");
        WM_REQUIRE(T == 0 || io->past_capacity >= T, "wm_decoder_step: past capacity %d < n_past %d", io->past_capacity, T);
        WM_REQUIRE(io->present_capacity >= T + L, "wm_decoder_step: present capacity %d < n_past + n_new = %d", io->present_capacity, T + L);
        WM_REQUIRE(!io->n_past_dev || (L == 1 && io->past), "wm_decoder_step: a device step counter needs n_new == 1 and past buffers");
        WM_REQUIRE(C == H * 64, "head size must be 64");
        w = carve_decoder(e, B, L, io->workspace);
        WM_REQUIRE(io->workspace_bytes >= w.total, "decoder workspace too small: %zu < %zu", io->workspace_bytes, w.total);
        small = M <= small_path_max_rows();
        rows = !small && rows_path_min_rows() > 0 && M >= rows_path_min_rows() && !e->dec.empty() && gemm_rows_supports(C, e->dec[0].qkv.wcode);
        chain = false;
        hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cap_st) == hipSuccess && cap_st != hipStreamCaptureStatusNone;
        ChainDev& cd = chain_dev_slot(e->device);
        // A chain launch on this device has given up a wait and nobody has acknowledged it (wm_decode_chain_error): the results of that
        // step -- and of everything decoded from it -- are invalid.  Every decoder call fails loudly until the caller has looked.
        // (Not while a stream is being captured: the call issues no work then, and failing it would abort the caller's capture.)
        if (!capturing && chain_err_peek(cd) != 0) {
            set_error("wm_decoder_step: a one-launch decode step on device %d gave up waiting for its workgroups (they were not resident "
                      "together): the results since then are invalid.  Call wm_decode_chain_error() to acknowledge and decode again -- the device "
                      "takes the launch-per-kernel path from then on (wm_set_decode_chain re-arms the one-launch forms)", e->device);
            return 1;
        }
        if (alone && small && L == 1 && M <= chain_max_rows() && !e->dec.empty() && e->chain_dev && g_decode_chain.load(std::memory_order_relaxed) && !io->qkv_amax) {
            if (chain_dev_init(cd, e->device)) return 2;
            const int n_cu = cd.n_cu;
            // one-row groups run a decoder layer (or the whole step) as one launch -- with the in-place cache (past[i] == present[i]), four
            // key-range pieces, fp16 cross K/V, the four-wave self-attention form; anything else takes the launch-per-kernel path
            chain_wgs = n_cu > 256 ? 256 : n_cu; chain_err = cd.err_dev; chain_cd = &cd;
            bool ok = gemv_chain_supports(C, e->dec[0].qkv.wcode, n_cu) && w.nsplit == 4 && !e->i8cross() && self_attn_waves(M) == 4 &&
                      io->present_capacity <= 512 && !(M > 2 && e->dec[0].qkv.wcode == 4) &&
                      (M > 4 ? (M * H <= chain_wgs && M * H * w.nsplit <= 4 * chain_wgs)                   // (5-8 rows: two items per workgroup and round, <= 2 rounds, beside the self-attention's heads)
                             : M * H + (M > 2 ? (M * H * w.nsplit + 1) / 2 : M * H * w.nsplit) <= chain_wgs);      // (3, 4 rows: two cross-attention items per workgroup)
            for (int i = 0; ok && i < e->dims.n_text_layer; ++i)
                ok = io->present[i] && io->cross[i] &&
                     (T == 0 ? io->n_past_dev == nullptr : (io->past[i] == io->present[i] && io->past_capacity == io->present_capacity));
            // ... and only where the launch's workgroups can be resident TOGETHER (they wait for each other): a device that has let a chain
            // down before is not asked again, the runtime's occupancy figure must cover the grid, and the stream must own every CU
            if (ok) {
                std::lock_guard<std::mutex> lk(cd.mu);
                ok = !cd.declined;
                if (ok) {
                    const long long key = ((long long)M << 48) | ((long long)e->dec[0].qkv.wcode << 40) | ((long long)(e->i8kv() ? 1 : 0) << 32) | ((long long)e->dims.n_audio_ctx << 8) | w.nsplit;
                    auto it = cd.resident.find(key);
                    if (it == cd.resident.end()) {
                        bool fits = false;
                        char why[200] = "";
                        if (gemv_chain_resident(e->dec[0].qkv.wcode, e->i8kv() ? 1 : 0, M, e->dims.n_audio_ctx, w.nsplit, chain_wgs, n_cu, &fits, why, sizeof(why))) return 2;
                        it = cd.resident.emplace(key, fits).first;
                        cd.footprint = why;
                        if (!fits) cd.reason = why;
                    }
                    ok = it->second;
                }
            }
            if (ok && !stream_has_all_cus(stream, n_cu)) ok = false;
            if (!ok) cd.declined_calls.fetch_add(1, std::memory_order_relaxed);
            chain = ok;
        }
        return 0;
    }

    // the arguments every chain launch shares
    void chain_common(GemvChainParams& p) {
        const wm_dims& d = e->dims;
        p.cross_Tk = d.n_audio_ctx; p.cross_heads = H; p.cross_nsplit = w.nsplit; p.gran_q = w.gran_q; p.cross_at = 1;
        p.gran_p = w.gran_p; p.merge_at = 2; p.merge_nsplit = w.nsplit; p.merge_heads = H;
        p.self_cap = io->present_capacity; p.self_T = T; p.self_t_dev = io->n_past_dev; p.self_heads = H; p.self_i8 = e->i8kv() ? 1 : 0;
        p.gran_c = w.gran_c;
        p.out32 = w.part; p.w8 = e->dec[0].out.wcode; p.gelu_kind = e->gelu();
        p.x = w.x; p.gran_x = w.gran_x; p.gran_h = w.gran_h; p.err = chain_err;
        p.generation = w.generation;
        p.rows = M;                                   // (L == 1: one row per utterance; the utterances' buffers are slices of [B, 2, H, T, 64])
        p.live = io->live_rows;                       // rows still decoding (optional): the others' attention stages read and append nothing
        p.cross_row_bytes = (long)2 * H * d.n_audio_ctx * 64 * 2;
        p.self_row_bytes = (long)2 * H * io->present_capacity * 64 * (e->i8kv() ? 1 : 2);
    }

    // decoder layer i of a one-row group as ONE launch (gemv_chain.hip): self-attention, out, cq, cross-attention, merge + cout, mlp1,
    // mlp2, the qkv sums of the next layer
    int run_layer(int i, hipStream_t s) {
        const DecLayer& Lr = e->dec[i];
        const bool more = i + 1 < e->dims.n_text_layer;
        GemvChainParams p{};
        chain_common(p);
        p.n_stages = more ? 6 : 5; p.st = e->chain_dev + 1 + (size_t)6 * i;
        p.cross_kv = (const h16*)io->cross[i]; p.cross_qbias = Lr.cq.b;
        p.self_part = w.part; p.self_bias = Lr.qkv.b; p.self_cache = io->present[i]; p.self_kv_scale = Lr.kv_scale;
        p.launch_id = i;
        chain_cd->launches.fetch_add(1, std::memory_order_relaxed);
        return launch_gemv_chain(p, &e->chain_host[1 + (size_t)6 * i], chain_wgs, s);
    }

    int finish(const Lin& l, int ks, int mode, const h16* g, const h16* bta, h16* out, int ldo, int N, hipStream_t s) {
        RowFinishParams p{};
        p.part = w.part; p.ksplit = ks; p.M = M; p.N = N; p.ldp = l.N; p.part_sstride = (long)M * l.N;
        p.bias = l.b; p.mode = mode; p.gelu_kind = e->gelu(); p.x = w.x; p.ldx = C; p.ln_g = g; p.ln_b = bta;
        p.out = out; p.ldo = ldo;
        return launch_row_finish(p, s);
    }

    // token + positional embedding, first LayerNorm
    int begin(hipStream_t s) {
        const wm_dims& d = e->dims;
        EmbedParams ep{io->tokens, io->tokens_ld > 0 ? io->tokens_ld : L, M, L, e->emb_t, C,
                       (const h16*)io->positional_embedding, w.x, C, d.n_vocab, io->n_past_dev, nullptr};
        ep.generation = chain ? w.generation : nullptr;      // the chain's granule epochs count the calls on this workspace
        if (chain) if (int rc = chain_prepare_workspace(s)) return rc;
        if (launch_embed(ep, s)) return 2;
        if (small) return 0;                         // the first LayerNorm happens inside the qkv projection
        return launch_layernorm(w.x, C, M, C, e->dec[0].ln1g, e->dec[0].ln1b, w.xn, C, s);
    }

    // self-attention block and the cross-attention query projection of layer i
    // The whole token step of a one-row group as ONE launch: [LayerNorm + qkv of layer 0], then per layer self-attention, out, cq,
    // cross-attention, merge + cout, mlp1, mlp2, qkv of the next layer -- gemv_chain.hip walks over the layers itself; the per-layer
    // cross K/V and cache pointers reach it through a table in the workspace, rewritten (small launches on this stream) only when
    // the caller's pointers differ from the ones last written there.
    // The chain's state in the caller's workspace -- tagged granules, the call counter, the per-layer pointer table -- is the LIBRARY's to
    // initialise: the workspace arrives as the allocator left it (the reference's plugins get theirs the same way; INTEGRATION.md's stub
    // uses torch.empty).  A state the library has not initialised under this (address, workspace_id) is cleared by a memset ON THE
    // STREAM OF THE CALL, ahead of the embedding kernel that opens the call (which counts the cleared call counter up to 1: tag 0 is
    // never valid, and bit 31 of every epoch is set besides).  workspace_id == 0 vouches for nothing: cleared on every call.  Under
    // stream capture the nodes are recorded but not run, so a captured call is never REMEMBERED: one that meets a state no eager
    // call has initialised carries the initialisation inside its graph (every replay clears and rewrites: correct, slower), one that
    // follows an eager call on the same (address, id) -- decoding.py's order -- relies on the state that call left.
    bool ws_fresh = false, ws_capturing = false;
    int chain_prepare_workspace(hipStream_t s) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        ws_capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        ws_fresh = true;
        if (io->workspace_id != 0) {
            std::lock_guard<std::mutex> lk(e->chain_io_mu);
            auto it = e->chain_io_seen.find(w.layer_io);
            // known: an EAGER call has initialised this state under this id (it persists in the workspace: a call captured later
            // relies on it, like an eager one -- a memset node and four table launches in every replay cost the batch-1 step 2 %)
            ws_fresh = it == e->chain_io_seen.end() || it->second.id != io->workspace_id;
            if (ws_fresh && !ws_capturing) {         // (a captured call enqueues nothing: it initialises inside its graph and is not remembered)
                if (e->chain_io_seen.size() > 4096) e->chain_io_seen.clear();
                wm_engine::ChainWsSeen& seen = e->chain_io_seen[w.layer_io];
                seen.id = io->workspace_id; seen.tab.clear();
            }
        }
        if (ws_fresh) {
            unsigned char* lo = (unsigned char*)w.gran_x;
            unsigned char* hi = (unsigned char*)(w.generation + 4);
            WM_CHECK_HIP(hipMemsetAsync(lo, 0, (size_t)(hi - lo), s));
        }
        return 0;
    }

    bool step_done = false;
    int whole_step(hipStream_t s) {
        const int n = e->dims.n_text_layer;
        std::vector<ChainLayerIo> tab((size_t)n);
        for (int i = 0; i < n; ++i) tab[i] = ChainLayerIo{io->cross[i], io->present[i]};
        if (io->workspace_id == 0) {                 // nothing is vouched for: the table is rewritten on every call
            if (launch_chain_io_table(w.layer_io, tab.data(), n, s)) return 2;
        } else {
            std::lock_guard<std::mutex> lk(e->chain_io_mu);
            auto it = e->chain_io_seen.find(w.layer_io);
            const bool have = it != e->chain_io_seen.end() && it->second.id == io->workspace_id;
            const bool same = !ws_fresh && have && it->second.tab.size() == tab.size() &&
                              memcmp(it->second.tab.data(), tab.data(), tab.size() * sizeof(ChainLayerIo)) == 0;
            if (!same) {
                if (launch_chain_io_table(w.layer_io, tab.data(), n, s)) return 2;
                if (have) {
                    if (ws_capturing) it->second.tab.clear();      // the replay will rewrite the table: what an eager call wrote is no longer known to be there
                    else it->second.tab = tab;
                }
            }
        }
        GemvChainParams p{};
        chain_common(p);
        p.n_layers = n; p.lstat = e->chain_lstat; p.lio = w.layer_io; p.gran_s = w.gran_s;
        p.st = e->chain_dev; p.n_stages = 0; p.launch_id = 0;
        if (launch_gemv_chain(p, e->chain_host.data(), chain_wgs, s)) return 2;
        chain_cd->launches.fetch_add(1, std::memory_order_relaxed);
        step_done = true;
        return 0;
    }

    int pre_cross(int i, hipStream_t s) {
        const DecLayer& Lr = e->dec[i];
        int ks = 0;
        if (i == 0) {
            step_done = false;
            if (chain && g_decode_chain.load(std::memory_order_relaxed) >= 2 && e->dims.n_text_layer <= 32 && e->chain_lstat)
                if (int rc = whole_step(s)) return rc;
        }
        if (step_done) return 0;
        mark(i, 0, s);
        if (small) {
            // (chained: layers > 0 got their qkv sums from the chain that closed the layer before)
            if (!(chain && i > 0) && gemv(Lr.qkv, w.x, C, 0, Lr.ln1g, Lr.ln1b, nullptr, 0, s)) return 2;      // LN + qkv sums -> w.part [M][3C]
            ks = 1;
        } else if (rows) {     // xn = LayerNorm(x) came with the row kernel that closed the previous layer (or from begin())
            if (gemv(Lr.qkv, w.xn, C, 0, nullptr, nullptr, nullptr, 0, s)) return 2;
            ks = 1;
        } else if (skinny_all(Lr.qkv, w.xn, C, M, w.part, &ks, s)) return 2;
        mark(i, 1, s);
        AttnSelfParams p{};
        p.part = w.part; p.ksplit = ks; p.ldp = Lr.qkv.N; p.part_sstride = (long)M * Lr.qkv.N; p.bias = Lr.qkv.b;
        p.B = B; p.L = L; p.T = T; p.H = H;
        WM_REQUIRE(io->present[i], "wm_decoder_step: present[%d] is null", i);
        p.present = io->present[i]; p.present_cap = io->present_capacity; p.present_bstride = (long)2 * H * io->present_capacity * 64;
        if (T > 0) {
            WM_REQUIRE(io->past[i], "wm_decoder_step: past[%d] is null", i);
            p.past = io->past[i]; p.past_cap = io->past_capacity; p.past_bstride = (long)2 * H * io->past_capacity * 64;
        } else { p.past = p.present; p.past_cap = p.present_cap; p.past_bstride = p.present_bstride; }
        p.int8_kv = e->i8kv(); p.kv_scale = Lr.kv_scale; p.out = w.ctx; p.ldo = C;
        p.amax = io->qkv_amax ? io->qkv_amax + i : nullptr;
        p.t_dev = io->n_past_dev;
        p.live = io->live_rows;
        p.waves = self_attn_waves(M);
        layer_done = false;
        if (chain) {                   // the whole layer in one launch (a live-row list is not consulted: a one-row group is stepped while its
            layer_done = true;         // row decodes; the steps between the row's EOT and the host's next poll compute values nobody reads)
            if (run_layer(i, s)) return 2;
            mark(i, 12, s);
            return 0;
        }
        if (launch_attn_self(p, s)) return 2;
        mark(i, 2, s);
        if (fused()) {
            if (gemv(Lr.out, w.ctx, C, 2, nullptr, nullptr, nullptr, 0, s)) return 2;    // x += out(ctx)
            mark(i, 3, s);
            if (gemv(Lr.cq, w.x, C, 0, Lr.lncg, Lr.lncb, nullptr, 0, s)) return 2;       // LN + q sums -> w.part [M][C]
            mark(i, 5, s);
            cq_ks = 1;
            return 0;
        }
        if (skinny_all(Lr.out, w.ctx, C, M, w.part, &ks, s)) return 2;
        mark(i, 3, s);
        if (finish(Lr.out, ks, 0, Lr.lncg, Lr.lncb, w.xn, C, C, s)) return 2;
        mark(i, 4, s);
        if (skinny_all(Lr.cq, w.xn, C, M, w.part, &cq_ks, s)) return 2;
        mark(i, 5, s);
        return 0;
    }
    int cq_ks = 0;
    bool layer_done = false;                     // this layer ran as one chain launch (cross() and post_cross() have nothing left to do)

    // the HBM-bound kernel: K and V of every utterance of the group, once
    int cross(int i, hipStream_t s) {
        if (step_done || layer_done) return 0;           // done by the chain launch
        if (!prof->timeline) return cross_launch(i, s);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, prof->timeline, prof->timeline_cap, (long long)(uintptr_t)io->logits, (long long)(2 * i));
        const int rc = cross_launch(i, s);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, prof->timeline, prof->timeline_cap, (long long)(uintptr_t)io->logits, (long long)(2 * i + 1));
        return rc;
    }

    int cross_launch(int i, hipStream_t s) {
        const DecLayer& Lr = e->dec[i];
        const wm_dims& d = e->dims;
        AttnCrossParams p{};
        p.part = w.part; p.ksplit = cq_ks; p.ldp = Lr.cq.N; p.part_sstride = (long)M * Lr.cq.N; p.bias = Lr.cq.b;
        p.B = B; p.L = L; p.H = H; p.Tk = d.n_audio_ctx;
        WM_REQUIRE(io->cross[i], "wm_decoder_step: cross[%d] is null", i);
        p.kv = (const h16*)io->cross[i]; p.kv_bstride = (long)2 * H * d.n_audio_ctx * 64;
        p.kv_q8_scale = e->i8cross() ? Lr.cross_scale : 0.f;
        p.out = w.ctx; p.ldo = C; p.nsplit = w.nsplit; p.ws = w.cross_ws;
        p.live = io->live_rows;
        p.skip_zero_rows = g_cross_v_skip.load(std::memory_order_relaxed);
        const int slot = (L == 1) ? prof_slot(*prof, i, s) : -1;
        if (launch_attn_cross(p, s, slot >= 0 ? prof->start[slot] : nullptr, slot >= 0 ? prof->stop[slot] : nullptr)) return 2;
        return 0;
    }

    // cross-attention output projection and the MLP of layer i (ends with the next layer's LayerNorm)
    int post_cross(int i, hipStream_t s) {
        const DecLayer& Lr = e->dec[i];
        const wm_dims& d = e->dims;
        int ks = 0;
        mark(i, 6, s);
        if (step_done || layer_done) return 0;
        if (fused()) {
            if (gemv(Lr.cout, w.ctx, C, 2, nullptr, nullptr, nullptr, 0, s)) return 2;
            mark(i, 7, s);
            if (gemv(Lr.mlp1, w.x, C, 1, Lr.ln2g, Lr.ln2b, w.hid, 4 * C, s)) return 2;     // LN + GELU
            mark(i, 10, s);
            if (small) return gemv(Lr.mlp2, w.hid, 4 * C, 2, nullptr, nullptr, nullptr, 0, s);
            // K = 4 n_state: the hidden rows of a 32-row block do not fit one workgroup's LDS -- K slices over workgroups (and row
            // splits), the row kernel adds them to the residual stream and applies the next layer's first LayerNorm, whose
            // output 60-80 column groups of the qkv projection would otherwise each recompute
            if (skinny_all(Lr.mlp2, w.hid, 4 * C, M, w.part, &ks, s)) return 2;
            mark(i, 11, s);
            const bool last_ = (i + 1 == d.n_text_layer);
            const int rc = finish(Lr.mlp2, ks, 0, last_ ? e->lnfg : e->dec[i + 1].ln1g, last_ ? e->lnfb : e->dec[i + 1].ln1b, w.xn, C, C, s);
            mark(i, 12, s);
            return rc;
        }
        if (skinny_all(Lr.cout, w.ctx, C, M, w.part, &ks, s)) return 2;
        mark(i, 7, s);
        if (finish(Lr.cout, ks, 0, Lr.ln2g, Lr.ln2b, w.xn, C, C, s)) return 2;
        mark(i, 8, s);
        if (skinny_all(Lr.mlp1, w.xn, C, M, w.part, &ks, s)) return 2;
        mark(i, 9, s);
        if (finish(Lr.mlp1, ks, 1, nullptr, nullptr, w.hid, 4 * C, 4 * C, s)) return 2;
        mark(i, 10, s);
        if (skinny_all(Lr.mlp2, w.hid, 4 * C, M, w.part, &ks, s)) return 2;
        mark(i, 11, s);
        const bool last = (i + 1 == d.n_text_layer);
        const int rc = finish(Lr.mlp2, ks, 0, last ? e->lnfg : e->dec[i + 1].ln1g, last ? e->lnfb : e->dec[i + 1].ln1b, w.xn, C, C, s);
        mark(i, 12, s);
        return rc;
    }

    // logits = ln(x) . E^T (fp16 out, whisper/model.py:288-290)
    int end(hipStream_t s) {
        const wm_dims& d = e->dims;
        if (small) {
            // the final LayerNorm as a launch of its own: 3 242 workgroups (16 vocabulary entries each) would all redo it
            if (launch_layernorm(w.x, C, M, C, e->lnfg, e->lnfb, w.xn, C, s)) return 2;
            GemvSmallParams p{};
            p.A = w.xn; p.lda = C; p.M = M; p.K = C; p.Wt = e->emb_t; p.n_blocks = e->emb_blocks; p.w8 = 0;
            p.mode = 3; p.out16 = (h16*)io->logits; p.ld16 = d.n_vocab; p.n_valid = d.n_vocab;
            return launch_gemv_small(p, s);
        }
        for (int r0 = 0; r0 < M; r0 += SKINNY_MAX_M) {
            GemmSkinnyParams p{};
            p.A = w.xn + (size_t)r0 * C; p.lda = C; p.M = (M - r0) < SKINNY_MAX_M ? (M - r0) : SKINNY_MAX_M; p.K = C;
            p.Wt = e->emb_t; p.n_blocks = e->emb_blocks; p.w8 = 0; p.ksplit = 1;
            p.out = (h16*)io->logits + (size_t)r0 * d.n_vocab; p.ldc = d.n_vocab; p.n_valid = d.n_vocab;
            if (launch_gemm_skinny(p, s)) return 2;
        }
        return 0;
    }
};

}  // namespace

int wm_decoder_step(const wm_engine* e, const wm_decoder_io* io, wm_stream_t stream_) {
    WM_REQUIRE(e && e->kind == WM_ENGINE_DECODER, "wm_decoder_step: not a decoder engine");
    hipStream_t s = (hipStream_t)stream_;
    if (io && io->n_new > DEC_CHUNK) {              // a long token block: 4-token passes over the growing cache
        WM_REQUIRE(!io->n_past_dev, "wm_decoder_step: a device step counter needs n_new == 1");
        WM_REQUIRE(io->tokens && io->positional_embedding && io->logits && io->workspace && io->present, "wm_decoder_step: null argument");
        const int L = io->n_new, B = io->batch, C = e->dims.n_text_state, V = e->dims.n_vocab;
        WM_REQUIRE(B >= 1 && io->n_past >= 0 && io->n_past + L <= e->dims.n_text_ctx, "wm_decoder_step: n_past+n_new=%d exceeds n_text_ctx=%d",
                   io->n_past + L, e->dims.n_text_ctx);
        const size_t base = carve_decoder(e, B, DEC_CHUNK, nullptr).total;
        const size_t need = base + align_up((size_t)B * DEC_CHUNK * V * sizeof(h16));
        WM_REQUIRE(io->workspace_bytes >= need, "decoder workspace too small: %zu < %zu", io->workspace_bytes, need);
        h16* stage = (h16*)((unsigned char*)io->workspace + base);
        for (int off = 0; off < L; off += DEC_CHUNK) {
            const int l = (L - off) < DEC_CHUNK ? (L - off) : DEC_CHUNK;
            wm_decoder_io sub = *io;
            sub.n_new = l; sub.n_past = io->n_past + off;
            sub.tokens = io->tokens + off; sub.tokens_ld = io->tokens_ld > 0 ? io->tokens_ld : L;
            sub.positional_embedding = (const h16*)io->positional_embedding + (size_t)off * C;
            sub.logits = stage; sub.workspace_bytes = base;
            if (off > 0) { sub.past = (const void* const*)io->present; sub.past_capacity = io->present_capacity; }   // the cache so far lives in `present`
            if (int rc = wm_decoder_step(e, &sub, stream_)) return rc;
            // [B, l, V] -> rows off .. off+l of the caller's [B, L, V]
            WM_CHECK_HIP(hipMemcpy2DAsync((h16*)io->logits + (size_t)off * V, (size_t)L * V * sizeof(h16), stage, (size_t)l * V * sizeof(h16),
                                          (size_t)l * V * sizeof(h16), (size_t)B, hipMemcpyDeviceToDevice, s));
        }
        return 0;
    }
    GroupStep g;
    if (int rc = g.init(e, io, s, !(io && io->not_alone))) return rc;
    if (g.begin(s)) return 2;
    for (int i = 0; i < e->dims.n_text_layer; ++i) {
        if (g.pre_cross(i, s)) return 2;
        if (g.cross(i, s)) return 2;
        if (g.post_cross(i, s)) return 2;
    }
    return g.end(s);
}

int wm_decoder_step_multi(const wm_engine* e, int n_groups, const wm_decoder_io* const* ios,
                          const wm_stream_t* light_streams, wm_stream_t heavy_stream) {
    WM_REQUIRE(e && e->kind == WM_ENGINE_DECODER, "wm_decoder_step_multi: not a decoder engine");
    WM_REQUIRE(n_groups >= 1 && n_groups <= 8 && ios && light_streams && heavy_stream, "wm_decoder_step_multi: bad arguments");
    hipStream_t hs = (hipStream_t)heavy_stream;
    GroupStep g[8];
    for (int k = 0; k < n_groups; ++k) {
        WM_REQUIRE(light_streams[k] && light_streams[k] != heavy_stream, "wm_decoder_step_multi: group %d needs its own stream", k);
        if (int rc = g[k].init(e, ios[k], (hipStream_t)light_streams[k], n_groups == 1 && !(ios[k] && ios[k]->not_alone))) return rc;
    }
    const int n_layer = e->dims.n_text_layer;
    for (int k = 0; k < n_groups; ++k)
        if (g[k].begin((hipStream_t)light_streams[k])) return 2;
    for (int i = 0; i < n_layer; ++i) {
        for (int k = 0; k < n_groups; ++k) {
            hipStream_t ls = (hipStream_t)light_streams[k];
            hipEvent_t q_ready = e->events.get(((size_t)k * n_layer + i) * 2), ctx_ready = e->events.get(((size_t)k * n_layer + i) * 2 + 1);
            WM_REQUIRE(q_ready && ctx_ready, "wm_decoder_step_multi: hipEventCreate failed");
            if (g[k].pre_cross(i, ls)) return 2;
            WM_CHECK_HIP(hipEventRecord(q_ready, ls));
            WM_CHECK_HIP(hipStreamWaitEvent(hs, q_ready, 0));
            if (g[k].cross(i, hs)) return 2;
            WM_CHECK_HIP(hipEventRecord(ctx_ready, hs));
            WM_CHECK_HIP(hipStreamWaitEvent(ls, ctx_ready, 0));
        }
        for (int k = 0; k < n_groups; ++k)
            if (g[k].post_cross(i, (hipStream_t)light_streams[k])) return 2;
    }
    for (int k = 0; k < n_groups; ++k)
        if (g[k].end((hipStream_t)light_streams[k])) return 2;
    return 0;
}

int wm_stream_create_cu_mask(const uint32_t* mask, int n_words, wm_stream_t* out) {
    WM_REQUIRE(mask && n_words >= 1 && out, "wm_stream_create_cu_mask: bad arguments");
    hipStream_t s = nullptr;
    WM_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
    *out = (wm_stream_t)s;
    return 0;
}

int wm_stream_destroy(wm_stream_t stream) {
    if (stream) WM_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

// ================================================================================================ profiling
int wm_profile_configure(int enabled, int layer_stride, int max_samples) {
    WM_REQUIRE(layer_stride >= 1 && max_samples >= 0, "wm_profile_configure: bad arguments");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    Profiler& g_prof = g_prof_dev[current_device_index()];
    g_prof.enabled = false;
    for (size_t i = 0; i < g_prof.start.size(); ++i) { (void)hipEventDestroy(g_prof.start[i]); (void)hipEventDestroy(g_prof.stop[i]); }
    g_prof.start.clear(); g_prof.stop.clear(); g_prof.used = 0; g_prof.layer_stride = layer_stride;
    if (!enabled) return 0;
    g_prof.start.resize(max_samples); g_prof.stop.resize(max_samples);
    for (int i = 0; i < max_samples; ++i) {
        WM_CHECK_HIP(hipEventCreate(&g_prof.start[i]));
        WM_CHECK_HIP(hipEventCreate(&g_prof.stop[i]));
    }
    g_prof.enabled = true;
    return 0;
}

int wm_debug_timeline(void* buf, int capacity) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    Profiler& pr = g_prof_dev[current_device_index()];
    pr.timeline = (long long*)buf;
    pr.timeline_cap = buf ? capacity : 0;
    return 0;
}

int wm_profile_read(double* total_ms, int64_t* count, int reset) {
    WM_REQUIRE(total_ms && count, "wm_profile_read: null argument");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    Profiler& g_prof = g_prof_dev[current_device_index()];
    double sum = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        WM_CHECK_HIP(hipEventSynchronize(g_prof.stop[i]));
        WM_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]));
        sum += ms;
    }
    *total_ms = sum; *count = (int64_t)g_prof.used;
    if (reset) g_prof.used = 0;
    return 0;
}

// ================================================================================================ greedy
int wm_greedy_step(const wm_greedy_io* io, wm_stream_t stream) {
    WM_REQUIRE(io && io->logits && io->tokens && io->sum_logprobs, "wm_greedy_step: null argument");
    GreedyParams p{};
    p.logits = (h16*)io->logits; p.ld_row = io->row_stride; p.B = io->batch; p.V = io->n_vocab;
    p.tokens = io->tokens; p.ld_tok = io->tokens_ld; p.cur_len = io->cur_len; p.sum_logprobs = io->sum_logprobs;
    p.suppress = io->suppress; p.n_suppress = io->n_suppress; p.blank = io->blank; p.n_blank = io->n_blank;
    p.sample_begin = io->sample_begin; p.eot = io->eot; p.timestamp_begin = io->timestamp_begin;
    p.max_initial_ts = io->max_initial_timestamp_index; p.apply_rules = io->apply_rules; p.n_done = io->n_done;
    p.t_dev = io->n_past_dev;
    p.done = io->done; p.row_limit = io->row_limit;
    WM_REQUIRE(io->temperature >= 0.f, "wm_greedy_step: negative temperature");
    p.temperature = io->temperature; p.seed_lo = (uint32_t)io->seed; p.seed_hi = (uint32_t)(io->seed >> 32);
    p.seed_dev = io->seed_dev; p.row0 = io->row0;
    return launch_greedy(p, (hipStream_t)stream);
}

int wm_step_advance(int32_t* counter, wm_stream_t stream) { return launch_step_advance(counter, (hipStream_t)stream); }
int wm_step_finish(int32_t* counter, const int32_t* done, int batch, int32_t* live, wm_stream_t stream) {
    return launch_step_finish(counter, done, batch, live, (hipStream_t)stream);
}

// ================================================================================================ kernel-level
int wm_gemm(const void* A, int lda, int M, int K, const void* W, int N, int w8, const void* scale,
            const void* bias, const void* residual, int ldr, int act, void* C, int ldc,
            void* workspace, size_t workspace_bytes, wm_stream_t stream) {
    GemmBigParams p{};
    p.A = (const h16*)A; p.lda = lda; p.M = M; p.K = K; p.W = W; p.N = N;
    p.bias = (const h16*)bias; p.C = (h16*)C; p.ldc = ldc;
    p.residual = (const h16*)residual; p.ldr = ldr; p.act = act;
    if (w8) {
        // exactly what the engines do with a weight-only matrix of an M >> 16 stage (expand_lin): one expansion to
        // fp16(fp16(q) * scale), then the fp16 MFMA GEMM -- here into the caller's workspace, per call
        WM_REQUIRE(scale && workspace, "wm_gemm: int8 weights need their scales and a workspace for the fp16 expansion");
        WM_REQUIRE(workspace_bytes >= (size_t)N * K * sizeof(h16), "wm_gemm: workspace too small: %zu < %zu", workspace_bytes, (size_t)N * K * sizeof(h16));
        if (launch_dequant_w8((const int8_t*)W, (const h16*)scale, (h16*)workspace, N, K, (hipStream_t)stream)) return 2;
        p.W = workspace;
    }
    return launch_gemm_f16(p, (hipStream_t)stream);
}

int wm_conv1d_gelu(const void* x_pad, int B, int T_in, int C_in, const void* W, int K, const void* bias, int C_out,
                   int stride, int gelu, void* out, wm_stream_t stream) {
    WM_REQUIRE(x_pad && W && out && B >= 1 && T_in >= 1, "wm_conv1d_gelu: null argument or empty input");
    WM_REQUIRE(stride == 1 || stride == 2, "wm_conv1d_gelu: stride %d (1 or 2)", stride);
    WM_REQUIRE(T_in % stride == 0 && K >= 3 * C_in, "wm_conv1d_gelu: T_in=%d, K=%d < 3*C_in=%d", T_in, K, 3 * C_in);
    const int T_out = T_in / stride;
    GemmBigParams p{};
    // output row t of utterance b = GELU(W . [x_pad[b][s*t], x_pad[b][s*t+1], x_pad[b][s*t+2]] + bias): a strided view
    p.A = (const h16*)x_pad; p.lda = stride * C_in; p.M = B * T_out; p.K = K; p.W = W; p.N = C_out;
    p.a_rows = T_out; p.a_bstride = (long)(T_in + 2) * C_in;
    p.bias = (const h16*)bias; p.C = (h16*)out; p.ldc = C_out; p.act = gelu;
    return launch_gemm_f16(p, (hipStream_t)stream);
}

int wm_argmax(const void* logits, int64_t row_stride, int batch, int n_vocab, int32_t* ids, wm_stream_t stream) {
    WM_REQUIRE(logits && ids && batch >= 1 && n_vocab >= 1, "wm_argmax: null argument or empty input");
    return launch_argmax((const h16*)logits, (long)row_stride, batch, n_vocab, ids, (hipStream_t)stream);
}

int wm_gemm_skinny(const void* A, int lda, int M, int K, const void* Wt, int n_blocks, int w8,
                   const void* scale, int ksplit, float* part, wm_stream_t stream) {
    GemmSkinnyParams p{};
    p.A = (const h16*)A; p.lda = lda; p.M = M; p.K = K; p.Wt = Wt; p.n_blocks = n_blocks; p.w8 = w8;
    p.scale = (const h16*)scale; p.ksplit = ksplit; p.part = part;
    return launch_gemm_skinny(p, (hipStream_t)stream);
}
int wm_gemm_skinny_default_ksplit(int M, int K, int n_blocks, int w8) { return skinny_default_ksplit(M, K, n_blocks, w8); }

int wm_set_rows_path(int min_rows) {
    const int prev = rows_path_min_rows();
    g_rows_min.store(min_rows < 0 ? 0 : min_rows, std::memory_order_relaxed);
    return prev;
}

int wm_gemm_rows(const wm_gemv_io* io, wm_stream_t stream) {
    WM_REQUIRE(io && io->a && io->wt, "wm_gemm_rows: null argument");
    GemvSmallParams p{};
    p.A = (const h16*)io->a; p.lda = io->lda; p.M = io->m; p.K = io->k; p.Wt = io->wt; p.n_blocks = io->n_blocks; p.w8 = io->w8;
    p.scale = (const h16*)io->scale;
    p.ln_g = (const h16*)io->ln_gamma; p.ln_b = (const h16*)io->ln_beta;
    p.mode = io->mode; p.bias = (const h16*)io->bias; p.gelu_kind = io->gelu_kind;
    p.out32 = io->out32; p.ld32 = io->ld32; p.out16 = (h16*)io->out16; p.ld16 = io->ld16; p.n_valid = io->n_valid;
    p.x = (h16*)io->x; p.ldx = io->ldx;
    return launch_gemm_rows(p, (hipStream_t)stream);
}

int wm_lab_knobs(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_lab_mu);
    if (buf && cap > 0) { strncpy(buf, g_lab_honoured.c_str(), cap - 1); buf[cap - 1] = 0; }
    return (int)g_lab_honoured.size();
}

int wm_set_gemm_small_tiles(int tiles) {
    const int prev = get_gemm_small_tiles();
    set_gemm_small_tiles(tiles);
    return prev;
}

int wm_set_self_attn_waves(int waves) {
    self_attn_waves(1);                          // (reads the environment once)
    const int prev = g_self_waves.load(std::memory_order_relaxed);
    g_self_waves.store(waves == 1 || waves == 4 ? waves : 0, std::memory_order_relaxed);
    return prev;
}

int wm_set_small_batch_rows(int rows) {
    const int prev = small_path_max_rows();
    g_small_rows.store(rows < 0 ? 0 : (rows > GEMV_SMALL_MAX_M ? GEMV_SMALL_MAX_M : rows), std::memory_order_relaxed);
    return prev;
}

int wm_gemv_fused(const wm_gemv_io* io, wm_stream_t stream) {
    WM_REQUIRE(io && io->a && io->wt, "wm_gemv_fused: null argument");
    GemvSmallParams p{};
    p.A = (const h16*)io->a; p.lda = io->lda; p.M = io->m; p.K = io->k; p.Wt = io->wt; p.n_blocks = io->n_blocks; p.w8 = io->w8;
    p.scale = (const h16*)io->scale;
    p.ln_g = (const h16*)io->ln_gamma; p.ln_b = (const h16*)io->ln_beta;
    p.mode = io->mode; p.bias = (const h16*)io->bias; p.gelu_kind = io->gelu_kind;
    p.out32 = io->out32; p.ld32 = io->ld32; p.out16 = (h16*)io->out16; p.ld16 = io->ld16; p.n_valid = io->n_valid;
    p.x = (h16*)io->x; p.ldx = io->ldx;
    return launch_gemv_small(p, (hipStream_t)stream);
}

int wm_layernorm(const void* x, int ldx, int M, int N, const void* gamma, const void* beta, void* out, int ldo,
                 wm_stream_t stream) {
    return launch_layernorm((const h16*)x, ldx, M, N, (const h16*)gamma, (const h16*)beta, (h16*)out, ldo, (hipStream_t)stream);
}

int wm_attn_encoder(const void* qkv, int ld, int B, int T, int H, void* out, int ldo, wm_stream_t stream) {
    AttnEncParams p{(const h16*)qkv, ld, B, T, H, (h16*)out, ldo};
    return launch_attn_encoder(p, (hipStream_t)stream);
}

int wm_attn_decode_cross(const float* q, int B, int L, int H, int Tk, const void* kv, void* out, int nsplit,
                         float* ws, wm_stream_t stream) {
    AttnCrossParams p{};
    p.part = q; p.ksplit = 1; p.ldp = H * 64; p.bias = nullptr;
    p.B = B; p.L = L; p.H = H; p.Tk = Tk; p.kv = (const h16*)kv; p.kv_bstride = (long)2 * H * Tk * 64;
    p.out = (h16*)out; p.ldo = H * 64; p.nsplit = nsplit; p.ws = ws;
    p.skip_zero_rows = g_cross_v_skip.load(std::memory_order_relaxed);
    return launch_attn_cross(p, (hipStream_t)stream);
}

int wm_set_decode_chain(int on) {
    const int prev = g_decode_chain.load(std::memory_order_relaxed);
    g_decode_chain.store(on < 0 ? DECODE_CHAIN_DEFAULT : (on > 2 ? 2 : on), std::memory_order_relaxed);
    for (ChainDev& cd : g_chain_dev) {               // an explicit choice re-arms devices that had let a chain down
        std::lock_guard<std::mutex> lk(cd.mu);
        if (cd.declined) { cd.declined = false; cd.reason.clear(); }
    }
    return prev;
}

int wm_decode_chain_error(int* out) {
    WM_REQUIRE(out, "wm_decode_chain_error: null argument");
    ChainDev& cd = chain_dev_slot(current_device_index());
    // No synchronisation here (round 5: a device-wide one also waited for whatever ran beside the loop -- the next batch's encoder on its
    // own stream -- and cost the pipelined batch-1 job 3.5 %): the word is in host memory and current for every step whose results the
    // caller has already waited for.
    const unsigned v = chain_err_peek(cd);
    *out = (int)v;
    if (v) {
        std::lock_guard<std::mutex> lk(cd.mu);
        __atomic_store_n(cd.err_host, 0u, __ATOMIC_RELAXED);
        cd.declined = true;
        cd.reason = "a one-launch decode step gave up waiting for its workgroups (they were not resident together)";
    }
    return 0;
}

int wm_decode_chain_status(wm_chain_status* out) {
    WM_REQUIRE(out, "wm_decode_chain_status: null argument");
    ChainDev& cd = chain_dev_slot(current_device_index());
    std::lock_guard<std::mutex> lk(cd.mu);
    memset(out, 0, sizeof(*out));
    out->mode = g_decode_chain.load(std::memory_order_relaxed);
    out->launches = cd.launches.load(std::memory_order_relaxed);
    out->declined_calls = cd.declined_calls.load(std::memory_order_relaxed);
    out->declined = cd.declined ? 1 : 0;
    out->error_pending = chain_err_peek(cd) ? 1 : 0;
    snprintf(out->reason, sizeof(out->reason), "%s", cd.reason.c_str());
    snprintf(out->footprint, sizeof(out->footprint), "%s", cd.footprint.c_str());
    return 0;
}

int wm_debug_occupy(int n_workgroups, size_t lds_bytes, int64_t microseconds, wm_stream_t stream) {
    return launch_occupy(n_workgroups, lds_bytes, (long long)microseconds, (hipStream_t)stream);
}

int wm_set_cross_v_skip(int on) {
    const int prev = g_cross_v_skip.load(std::memory_order_relaxed);
    g_cross_v_skip.store(on < 0 ? CROSS_V_SKIP_DEFAULT : (on ? 1 : 0), std::memory_order_relaxed);
    return prev;
}

int wm_attn_decode_cross_i8(const float* q, int B, int L, int H, int Tk, const void* kv_i8, float kv_scale, void* out, int nsplit,
                            float* ws, wm_stream_t stream) {
    WM_REQUIRE(kv_scale > 0.f, "wm_attn_decode_cross_i8: the scale must be positive");
    AttnCrossParams p{};
    p.part = q; p.ksplit = 1; p.ldp = H * 64; p.bias = nullptr;
    p.B = B; p.L = L; p.H = H; p.Tk = Tk; p.kv = (const h16*)kv_i8; p.kv_bstride = (long)2 * H * Tk * 64;
    p.kv_q8_scale = kv_scale;
    p.out = (h16*)out; p.ldo = H * 64; p.nsplit = nsplit; p.ws = ws;
    return launch_attn_cross(p, (hipStream_t)stream);
}

int wm_attn_decode_self(const float* qkv, int B, int L, int T, int H, const void* past, int past_cap,
                        void* present, int present_cap, int int8_kv, float kv_scale, void* out, wm_stream_t stream) {
    AttnSelfParams p{};
    p.part = qkv; p.ksplit = 1; p.ldp = 3 * H * 64; p.bias = nullptr;
    p.B = B; p.L = L; p.T = T; p.H = H;
    p.present = present; p.present_cap = present_cap; p.present_bstride = (long)2 * H * present_cap * 64;
    if (T > 0 && past) { p.past = past; p.past_cap = past_cap; p.past_bstride = (long)2 * H * past_cap * 64; }
    else { p.past = present; p.past_cap = present_cap; p.past_bstride = p.present_bstride; }
    p.int8_kv = int8_kv; p.kv_scale = kv_scale; p.out = (h16*)out; p.ldo = H * 64;
    p.waves = self_attn_waves(B * L);
    return launch_attn_self(p, (hipStream_t)stream);
}

int wm_quantize_i8(const void* x, void* q, int64_t n, float inv_scale, wm_stream_t stream) {
    return launch_quantize_i8((const h16*)x, (int8_t*)q, (long)n, inv_scale, (hipStream_t)stream);
}

size_t wm_log_mel_workspace_bytes(int batch, int n_samples, int n_mels) {
    return log_mel_workspace_bytes(batch, n_samples, n_mels);
}

int wm_log_mel(const float* audio, int batch, int n_samples, int64_t audio_ld, const float* filters, int n_mels,
               void* mel_f16, float* mel_f32, void* workspace, size_t workspace_bytes, wm_stream_t stream) {
    return launch_log_mel(audio, batch, n_samples, (long)audio_ld, filters, n_mels, (h16*)mel_f16, mel_f32, workspace,
                          workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
