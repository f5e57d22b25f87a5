// Host side of libmcpc.so: the C ABI declared in include/mcpc.h.
// Allocation, weight packing, launch orchestration (step segments + Hebbian flushes); no torch.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is loaded with dlopen on first use (no link dependency)

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcpc_build.h"
#include "mcpc_kernels.h"

using namespace mcpc;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return fail(MCPC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

inline int pad16(int n) { return (n + 15) / 16 * 16; }
inline int kblocks(int k_pad) { return (k_pad + kKB - 1) / kKB; }      // k-blocks of the GEMM core that cover k_pad (a multiple of 16)
inline int grid_for(size_t n, int block = 256) { return (int)std::min<size_t>((n + block - 1) / block, 4096); }
// which Linears (et x at tiles of 16) take the LDS-tiled Hebbian kernels (mcpc_hebbian.h): wide ones, and 256 outputs with a narrow input
inline bool heb_wide(int et, int at) { return et >= 8 && at % 8 == 0 && (at <= 16 || at % 16 == 0); }
inline bool heb_narrow_in(int et, int at) { return !heb_wide(et, at) && et == 16 && (at == 1 || at == 2 || at == 4); }

struct Lin {
    const float* W = nullptr;      // borrowed, torch layout [out][in]
    const float* bias = nullptr;   // borrowed or null
    bool bound = false;
    int n_out = 0, n_in = 0, out_pad = 0, in_pad = 0;
    float* Wf = nullptr;           // packed forward  [out_tiles][in_tiles][64][4]
    float* Wb = nullptr;           // packed backward [in_tiles][out_tiles][64][4]
    float* bias_pad = nullptr;     // [out_pad]
    float* G = nullptr;            // gradient sums [out_pad][g_ld]
    float* Gb = nullptr;           // [out_pad]
    int g_ld = 0;
    bool spill_tm = false;         // its Hebbian operands are spilled tile-major (the fp16 kernel mcpc_heb7_kernel reads them)
    size_t slab_off = 0;           // float offset of this Linear's split-K slabs inside mcpc_engine::slab
    size_t slab_floats = 0;        // ... and their size
};

// Developer overrides of the schedule heuristics, parsed ONCE from mcpc_net_desc::tuning at mcpc_create
// ("key=value,key=value"; see include/mcpc.h).  The library itself reads no environment variables.
constexpr int kMaxRingParts = 8;
struct Knobs {
    int ws = -1;              // -1: automatic; 0: barrier kernel; 2: in-place wave-specialised kernel everywhere; 3: insist on the unified-wave
                              // kernel (mcpc_steps_u.h) for the runs it serves (create fails when its LDS plan does not fit)
    int no_overlap = 0;       // 1: Hebbian flushes run serially on the caller's stream (one ring segment = the whole ring)
    int slot_cap = 384;       // spill-ring slots at most (3 parts of 128 steps)
    int spill_gb = 0;         // > 0: spill budget in GiB (overrides mcpc_net_desc::spill_budget_bytes)
    int cu_slack = 0;         // CUs the round schedule leaves free (launches of at most n_cu - cu_slack workgroups)
    int flush_streams = 2;    // low-priority streams the GEMMs of an overlapped flush are spread over (1 or 2)
    int flush_tail = 0;       // > 0: the last accumulating segment of a stretch is cut to this many steps (its flush is the one nothing overlaps)
    int ring_parts = 3;       // parts of the spill ring: one is filled by the step kernel, one is being flushed, one is slack -- with two
                              // halves the step kernel waited at every boundary for a flush that takes as long as its own segment
                              // (96.8 -> 95.2 us per step of the learning call; parts of 64 steps beat 48, 96 and 128)
    int dw_ksplit = 0;        // > 0: K-splits per workgroup tile of the Hebbian GEMM (0: one wave of workgroups over the chip)
    int ws_prio = 0;          // 1: epilogue waves at raised priority, 2: GEMM waves, 0: neither (the GEMM waves need most of the issue port
                              // themselves: with the epilogue waves at raised priority a step of cfg-M took 94.2 us against 87.7, round 3)
    int stagger = 0;          // barrier kernel: start cycles of the second workgroup of a CU
    int no_lean = 0;          // 1: the in-place kernel's E waves use the generic epilogues everywhere (A/B, parity tests)
    int no_ybits = 0;         // 1: 0/1 targets are read as fp32 like any other target (A/B, parity tests)
    int overlay16 = 0;        // 1: 16-chain plans share the LDS of the ring and the E_l like 32-chain plans do (A/B, parity tests)
    int no_xl = 0;            // 1: 16-chain plans keep the state and the per-step constants in global memory even when the LDS has the room (A/B, parity tests)
    int rr = 1;               // 0: shards of more 16-chain units than CUs run as one launch in hardware rounds instead of the round schedule (setup_rounds)
    int rr_qmax = 100;        // round schedule: most steps per launch in stretches without Hebbian accumulation
    int heb171 = 0;           // 1: the 17-tile group of a read-out on <17, 1> with twice the activation groups instead of <17, 2> (A/B)
    int heb_fp32 = 0;         // 1: the tiled Hebbian GEMM runs on the fp32 MFMA (mcpc_heb_kernel) instead of the fp16x6 form (A/B, parity tests)
    // unified-wave kernel: the cost model its rows are dealt by (build_phases_u) -- developer knobs for its calibration.  The defaults started
    // from in-kernel stamps (shader cycles: 2000 / 1200 / 330 / 68 / 1000 / 1300 / 500) and were then moved by a grid search on the step time
    // itself (scripts/u_cost_search.py, profiles/r06_small_net.txt: 16.9 -> 16.1 us per MCPC step at batch 256): what the model has to get
    // right is the ORDER of the jobs' costs and where a split stops paying, and a row's fixed cost and the read-out's epilogue weigh more in
    // that than their stamps say (they sit on the level's critical path)
    int u_row = 5000;         // per row: descriptor, the next row's fragment requests, the epilogue's fixed part
    int u_gemm0 = 1200;       // per GEMM, before its first k-block
    int u_kb = 200;           // per k-block beyond its tiles' MFMAs and requests (the B split; nothing with the operand in planes)
    int u_kbt = 68;           // per k-block and tile
    int u_eh = 3000, u_eb = 1300, u_ef = 500;      // per tile of an epilogue: read-out / x update / prediction error
};

int parse_tuning(const char* str, Knobs& k) {
    if (!str) return 0;
    std::string s(str);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find_first_of(",;", pos);
        if (end == std::string::npos) end = s.size();
        std::string item = s.substr(pos, end - pos);
        pos = end + 1;
        while (!item.empty() && item.front() == ' ') item.erase(item.begin());
        while (!item.empty() && item.back() == ' ') item.pop_back();
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        const std::string key = item.substr(0, eq);
        const int val = eq == std::string::npos ? 1 : atoi(item.c_str() + eq + 1);
        struct { const char* name; int* dst; } table[] = {
            {"ws", &k.ws}, {"no_overlap", &k.no_overlap},
            {"slot_cap", &k.slot_cap}, {"spill_gb", &k.spill_gb}, {"cu_slack", &k.cu_slack}, {"ring_parts", &k.ring_parts}, {"flush_tail", &k.flush_tail}, {"flush_streams", &k.flush_streams}, {"dw_ksplit", &k.dw_ksplit},
            {"ws_prio", &k.ws_prio}, {"stagger", &k.stagger}, {"no_lean", &k.no_lean}, {"no_ybits", &k.no_ybits}, {"overlay16", &k.overlay16}, {"heb_fp32", &k.heb_fp32}, {"heb171", &k.heb171}, {"rr", &k.rr}, {"rr_qmax", &k.rr_qmax}, {"no_xl", &k.no_xl},
            {"u_row", &k.u_row}, {"u_gemm0", &k.u_gemm0}, {"u_kb", &k.u_kb}, {"u_kbt", &k.u_kbt}, {"u_eh", &k.u_eh}, {"u_eb", &k.u_eb}, {"u_ef", &k.u_ef}};
        bool found = false;
        for (auto& t : table)
            if (key == t.name) { *t.dst = val; found = true; }
        if (!found) return fail(MCPC_EINVAL, "unknown tuning key '%s' in mcpc_net_desc::tuning", key.c_str());
    }
    if (k.ws != -1 && k.ws != 0 && k.ws != 2 && k.ws != 3) return fail(MCPC_EINVAL, "tuning ws=%d: 0 (barrier kernel), 2 (in-place kernel) or 3 (unified-wave kernel)", k.ws);
    if (k.slot_cap < 2) k.slot_cap = 2;
    if (k.cu_slack < 0) k.cu_slack = 0;
    if (k.ring_parts < 2 || k.ring_parts > kMaxRingParts) return fail(MCPC_EINVAL, "tuning ring_parts=%d: 2..%d", k.ring_parts, kMaxRingParts);
    return 0;
}

}  // namespace

// LDS plan and step table of the unified-wave kernel (mcpc_steps_u.h), kept BESIDE the engine's main plan: mcpc_run picks the kernel per
// run (lean runs: fused SGD update with or without the Philox kick, Adam without noise), everything else stays on the main plan's kernel.
struct UPlan {
    bool ok = false;                // the plan fits the LDS (whole read-out error + state rows resident)
    bool on = false;                // ... and the engine holds its table (automatic tuning, or ws=3)
    bool prefer = false;            // ... and uses it for every lean run (tuning ws=3, or the automatic choice: choose_unified); otherwise
                                    // only for zero-loss runs, whose read-out this kernel alone skips on the steps nobody records
    int lds_a[kMaxLatent]{}, lds_e[kMaxLatent]{}, lds_eo = 0, lds_red = 0, lds_ws_sync = 0, lds_zero = 0, lds_spillmax = 0, lds_rowexp = 0;
    int lds_x[kMaxLatent]{}, lds_bias[kMaxLatent]{}, lds_hbias = 0, lds_yw = 0, lds_bytes = 0;
    KPhase* phases = nullptr;
    int n_phases = 0;
};

struct mcpc_engine {
    mcpc_net_desc d{};
    Knobs knobs;
    ncclComm_t comm = nullptr;      // mcpc_comm_init: the shards' communicator (RCCL), one rank per engine
    int comm_ranks = 0;
    int L = 0, Bpad = 0, nwg = 0, has_head = 0;
    int nwg_live = 0;               // 16-chain in-place plans: workgroups that hold at least one chain of the batch (Bpad is a multiple of 32, so
                                    // the last 16-chain unit may be all padding: it is never launched -- its spill rows and energy slots stay zero)
    static constexpr int ct = 16;   // chains per workgroup: one MFMA column tile (the 32-chain forms of rounds 1-3 left the tree in round 4)
    int nw = kWaves;                // waves per workgroup: 4 (barrier kernel) or 8 (in-place kernel: 4 GEMM + 4 epilogue waves)
    int ws = 0;                     // 0: barrier kernel (the fallback, two workgroups per CU); 2: in-place wave-specialised kernel
    int ws2_chunk = 0, ws2_ring = 0; // in-place variant: read-out tiles per chunk, chunks in the LDS ring
    bool ws2_overlay = true;        // the ring of read-out error chunks shares LDS with E_1 .. E_{L-1}; plans that fit keep them apart,
                                    // which frees the order of the forward entries (build_phases_ws2)
    int lds_ws_sync = 0;
    int npad[kMaxLatent]{};
    int out_pad = 0;
    Lin lin[kMaxLatent + 1];
    float* x[kMaxLatent]{};
    float* m[kMaxLatent]{};
    float* v[kMaxLatent]{};
    float* e0sum = nullptr;
    float* mu1 = nullptr;
    float* ypad = nullptr;
    float* ytile = nullptr;         // tile-major copy of ypad (KHead::ytile)
    uint32_t* ybits = nullptr;      // bit-packed copy of a 0/1 target (see mcpc_pack_target_bits_kernel)
    int* y_binary = nullptr;        // device flag: the bound target is exactly 0/1 everywhere
    int ywords = 0;
    bool target_bound = false;
    const float* inputs = nullptr;
    // Hebbian spill ring: two halves, the flush of one half runs on `aux` while the step kernel fills the other
    int slots = 0, half_slots = 0;
    hipStream_t aux = nullptr;
    hipEvent_t ev_steps[kMaxRingParts] = {}, ev_flush[kMaxRingParts] = {};
    hipStream_t spacer = nullptr;   // never used: see ensure_spill
    hipStream_t aux3 = nullptr;     // second low-priority stream of the overlapped flush: the GEMMs of a flush alternate between the two,
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // so that the tail of one launch is filled by the next (same-stream kernels serialise)
    bool flush_pending[kMaxRingParts] = {};
    float* spill_a[kMaxLatent]{};
    float* spill_e[kMaxLatent]{};
    float* spill_eo = nullptr;
    float* slab = nullptr;
    size_t slab_floats = 0;
    // energies
    double* epart = nullptr;
    size_t epart_rows = 0;
    float* adam_coef = nullptr;     // device table [steps][2]; grown buffers are retired, never freed under a running kernel
    size_t adam_cap = 0;
    float* adam_host[2] = {nullptr, nullptr};     // pinned staging, double-buffered: a run never rewrites a table whose upload may be pending
    size_t adam_host_cap[2] = {0, 0};
    hipEvent_t adam_ev[2] = {nullptr, nullptr};
    int adam_next = 0;
    struct Retired { void* p; hipEvent_t ev; };
    std::vector<Retired> retired;   // device buffers replaced by larger ones while earlier launches may still use them: each carries an event
                                    // recorded on the stream at retirement and is freed by the first later run that finds the event complete
    bool spill_ready = false;       // the spill ring, its stream/events and the slabs exist (allocated by the first accumulating run)
    // LDS plan
    int lds_a[kMaxLatent]{}, lds_e[kMaxLatent]{}, lds_eo = 0, lds_red = 0, lds_bytes = 0;
    int lds_zero = 0;               // 16 floats nothing writes: the GEMM core's over-reading lanes read them (KParams::lds_zero)
    bool xl = false;                // 16-chain in-place plan with room: state rows, biases, mu_1 rows and target words live in LDS (KParams::xl)
    int lds_x[kMaxLatent]{}, lds_bias[kMaxLatent]{}, lds_hbias = 0, lds_yw = 0;
    // per-step phase table (device copy)
    KPhase* phases = nullptr;
    int n_phases = 0;
    int* err = nullptr;             // device error word written by the kernels
    int* wexp = nullptr;            // [kMaxLatent + 1] per Linear: the power of two its packed weights are scaled by (mcpc_wexp_kernel)
    unsigned* spillmax = nullptr;   // [kMaxRingParts][kSpillTensors]: per ring part, the largest |value| per spilled tensor of the segment
                                    // that filled it (bit patterns; written by the step kernels, read by the Hebbian GEMMs of that part)
    int lds_spillmax = 0;
    int g_first = -1;               // in-place table: first entry with work for the GEMM waves (build_phases_ws2)
    UPlan u;                        // unified-wave kernel: its own LDS plan and table
    int lds_rowexp = 0;             // in-place plan: kRowExpFloats words of row exponents (mcpc_kernels.h: rowexp_track)
    unsigned long long* clk = nullptr;   // profiling: {shader cycles, 100 MHz ticks} of one wave per launch (KParams::clk)
    float* dummy = nullptr;         // 4 KiB of zeros (KParams::dummy)
    // Round schedule (setup_rounds): a shard of more 16-chain units than CUs as `rr_k` launches per cycle, each unit in `rr_m` of them
    bool rr = false;
    int rr_k = 0, rr_m = 0;
    std::string rr_name;                 // mcpc_step_kernel_name of an engine on the round schedule
    std::string u_rr_name;               // ... when its fused calls run on the unified-wave kernel
    std::vector<int> rr_count, rr_off;   // per launch of a cycle: workgroups, offset of its [ids][rel] rows in rr_tab
    int* rr_tab = nullptr;
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // HIP events around every launch of the step kernel
    size_t events_used = 0;
    double prof_steps = 0;          // whole-shard steps covered by the bracketed launches
#ifdef MCPC_STAMPS
    unsigned long long* dbg = nullptr;
#endif
};

namespace {

// RCCL, bound at first use: a host that never shards the chains never loads it; inside a torch process dlopen returns the copy
// torch already mapped (same soname), so both share one HIP runtime.
struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.so) return MCPC_OK;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) return fail(MCPC_EHIP, "librccl.so.1 could not be loaded: %s", dlerror());
    Rccl r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(so, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(so, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(so, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(so, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(so, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
        dlclose(so);
        return fail(MCPC_EHIP, "librccl.so.1 lacks an entry point this library needs");
    }
    r.so = so;
    g_rccl = r;
    return MCPC_OK;
}

int free_all(mcpc_engine* e) {
    if (e->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(e->comm); e->comm = nullptr; e->comm_ranks = 0; }
    auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
    for (int l = 0; l < kMaxLatent; ++l) { F(e->x[l]); F(e->m[l]); F(e->v[l]); F(e->spill_a[l]); F(e->spill_e[l]); }
    F(e->e0sum); F(e->mu1); F(e->ypad); F(e->ytile); F(e->ybits); F(e->y_binary); F(e->spill_eo); F(e->slab); F(e->epart); F(e->adam_coef); F(e->phases); F(e->err); F(e->wexp); F(e->spillmax); F(e->clk); F(e->dummy); F(e->u.phases);
    for (auto& ln : e->lin) { F(ln.Wf); F(ln.Wb); F(ln.bias_pad); F(ln.G); F(ln.Gb); }
    for (auto& ev : e->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (int h = 0; h < kMaxRingParts; ++h) { if (e->ev_steps[h]) (void)hipEventDestroy(e->ev_steps[h]); if (e->ev_flush[h]) (void)hipEventDestroy(e->ev_flush[h]); e->ev_steps[h] = e->ev_flush[h] = nullptr; }
    if (e->aux) { (void)hipStreamDestroy(e->aux); e->aux = nullptr; }
    if (e->aux3) { (void)hipStreamDestroy(e->aux3); e->aux3 = nullptr; }
    if (e->spacer) { (void)hipStreamDestroy(e->spacer); e->spacer = nullptr; }
    if (e->ev_fork) { (void)hipEventDestroy(e->ev_fork); e->ev_fork = nullptr; }
    if (e->ev_join) { (void)hipEventDestroy(e->ev_join); e->ev_join = nullptr; }
    F(e->rr_tab);
    for (auto& q : e->retired) { (void)hipFree(q.p); if (q.ev) (void)hipEventDestroy(q.ev); }
    e->retired.clear();
    for (int i = 0; i < 2; ++i) {
        if (e->adam_host[i]) { (void)hipHostFree(e->adam_host[i]); e->adam_host[i] = nullptr; }
        if (e->adam_ev[i]) { (void)hipEventDestroy(e->adam_ev[i]); e->adam_ev[i] = nullptr; }
    }
    e->events.clear();
    return 0;
}

template <typename T>
int dmalloc(T*& p, size_t count) {
    void* q = nullptr;
    hipError_t err = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (err != hipSuccess) return fail(MCPC_ENOMEM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(err));
#ifdef MCPC_POISON_ALLOC      // diagnostic build: every fresh device allocation is filled with 0xFF bytes (NaN as fp32 / fp64, -1 as int), so that a
                              // read of memory nobody initialised shows on every box, not only on one whose fresh pages are not zero
    if (hipMemset(q, 0xFF, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return fail(MCPC_EHIP, "hipMemset (poison) failed");
#endif
    p = (T*)q;
    return 0;
}

// A buffer that launches already in the stream may still use is not freed but retired behind an event.
void retire(mcpc_engine* e, void* p, hipStream_t stream) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;                                  // no event: the buffer waits for mcpc_destroy
    }
    e->retired.push_back({p, ev});
}
// ... and freed by a later run once that event has completed (never under a kernel that uses it)
void free_completed_retired(mcpc_engine* e) {
    size_t keep = 0;
    for (size_t i = 0; i < e->retired.size(); ++i) {
        auto& q = e->retired[i];
        if (q.ev && hipEventQuery(q.ev) == hipSuccess) { (void)hipFree(q.p); (void)hipEventDestroy(q.ev); }
        else e->retired[keep++] = q;
    }
    e->retired.resize(keep);
    (void)hipGetLastError();                           // hipEventQuery's hipErrorNotReady is not an error of this run
}

// LDS plan: activations ping-pong between two buffers (FX_l in buffer l&1), the read-out error
// chunk takes the buffer FX_{L-1} is NOT in, errors E_l (l>=1) get their own rows.
int plan_lds(mcpc_engine* e) {
    int buf[2] = {0, 0};
    const int CT = e->ct;
    for (int l = 0; l < e->L; ++l) buf[l & 1] = std::max(buf[l & 1], CT * (e->npad[l] + kLdPad));
    const int eo_floats = e->has_head ? CT * (kChunkTiles * 16 + kLdPad) : 0;
    const int eo_buf = ((e->L - 1) & 1) ^ 1;
    buf[eo_buf] = std::max(buf[eo_buf], eo_floats);
    int off = 0;
    const int base0 = off; off += buf[0];
    const int base1 = off; off += buf[1];
    for (int l = 0; l < e->L; ++l) e->lds_a[l] = (l & 1) ? base1 : base0;
    e->lds_eo = eo_buf ? base1 : base0;
    for (int l = 1; l < e->L; ++l) { e->lds_e[l] = off; off += CT * (e->npad[l] + kLdPad); }
    e->lds_e[0] = 0;
    e->lds_red = off; off += 2 * (kMaxLatent + 1) * kMaxWaves;
    e->lds_zero = off; off += 16;            // (mcpc_gemm_f16.h: what lanes beyond a ragged k range read)
    e->lds_spillmax = off; off += kSpillTensors;     // the workgroup's largest |value| per spilled tensor (mcpc_kernels.h: spill_track)
    e->lds_bytes = off * (int)sizeof(float);
    if (e->lds_bytes > 160 * 1024)
        return fail(MCPC_ENOMEM, "network needs %d bytes of LDS per workgroup (> 163840): latent widths too large for the fused kernel", e->lds_bytes);
    return 0;
}

// LDS plan of the in-place wave-specialised kernel: every FX_l has its own rows (no staging slots).  With a read-out,
// the prediction errors E_1 .. E_{L-1} and the ring of read-out error chunks share ONE region: the ring is only live
// during the read-out phase at the start of a step, the E_l only from the forward entries (scheduled behind the
// read-out) to the x updates at its end.  At cfg-M that pays for a ring of THREE chunks of 13 tiles (49 tiles =
// 13+12+12+12; a table entry hands out up to 16 tiles, four per GEMM wave): two GEMMs of slack between a chunk's
// epilogue and its back-projection, 15 table entries per step.
int plan_lds_ws2(mcpc_engine* e, bool allow_xl = true) {
    const int CT = e->ct, L = e->L;
    int off = 0;
    for (int l = 0; l < L; ++l) { e->lds_a[l] = off; off += CT * (e->npad[l] + kLdPad); }
    e->lds_e[0] = 0;
    e->lds_red = off; off += 2 * (kMaxLatent + 1) * kMaxWaves;
    e->lds_ws_sync = off; off += 16;
    // E_1 .. E_{L-1} stacked; the ring of read-out error chunks behind them when both fit (16-chain workgroups: the forward
    // entries are then free to run between the read-out chunks), else on top of them (shared region)
    int e_sum = 0;
    for (int l = 1; l < L; ++l) { e->lds_e[l] = off + e_sum; e_sum += CT * (e->npad[l] + kLdPad); }
    e->lds_eo = off;
    e->ws2_chunk = 0; e->ws2_ring = 0; e->ws2_overlay = true;
    int ring_floats = 0;
    if (e->has_head) {
        // fewest chunks of at most 16 tiles whose ring of two still fits; chunks equalised; ring of three if that fits too
        const int ht = std::max(e->out_pad / 16, 1);
        const int span = kWs2Pairs * ws2_nt<1>();     // tiles a table entry hands out
        auto ring_of = [&](int hc, int nb) { return nb * CT * (hc * 16 + kLdPad); };
        // (+ tail: what is added behind the operand regions below -- a plan that close to the limit takes a smaller chunk or ring here
        // instead of failing the final size check)
        constexpr int tail = 16 + kSpillTensors + kRowExpFloats;      // (zero region, spill maxima, row exponents: added below)
        auto fits = [&](int hc, int nb) { return (off + tail + std::max(ring_of(hc, nb), e_sum)) * (int)sizeof(float) <= 160 * 1024; };
        auto fits_apart = [&](int hc, int nb) { return (off + tail + ring_of(hc, nb) + e_sum) * (int)sizeof(float) <= 160 * 1024; };
        // (chunks are whole k-blocks of the back-projection GEMM: tq tiles)
        const int tq = kKB / 16, hb = (ht + tq - 1) / tq;
        int hcb_fit = 0;
        for (int hcb = std::min(span / tq, hb); hcb >= 1; --hcb)
            if (fits(hcb * tq, 2)) { hcb_fit = hcb; break; }
        if (!hcb_fit) return fail(MCPC_ENOMEM, "in-place schedule does not fit the LDS");
        const int nch = (hb + hcb_fit - 1) / hcb_fit;
        const int hc = tq * ((hb + nch - 1) / nch);          // equalised: the widest chunk of the split (tiles)
        const int nb = (nch >= 3 && fits(hc, 3)) ? 3 : 2;
        e->ws2_chunk = hc; e->ws2_ring = nb; ring_floats = ring_of(hc, nb);
        if (L >= 2 && fits_apart(hc, nb) && !e->knobs.overlay16) { e->ws2_overlay = false; e->lds_eo = off + e_sum; }
    }
    off += e->ws2_overlay ? std::max(ring_floats, e_sum) : ring_floats + e_sum;
    // The GEMM core reads the LDS operand in whole 32-deep k-blocks; the lanes whose k values lie beyond a row whose width is not a
    // multiple of 32 read THESE 16 floats instead of what lies behind the row (mcpc_gemm_f16.h): zero-filled at launch, never written.
    e->lds_zero = off;
    off += 16;
    e->lds_spillmax = off; off += kSpillTensors;     // the workgroup's largest |value| per spilled tensor (mcpc_kernels.h: spill_track)
    e->lds_rowexp = off; off += kRowExpFloats;       // per B operand and chain row: generation and exponent of the row's maximum (rowexp_track)
    // with room to spare (16-chain plans: 45 KB at cfg-M) the lean epilogues keep what they read every step in LDS: the state rows
    // X_l (layout of FX_l), the bias rows, the mu_1 rows (layout of FX_0), the read-out bias and the bit-packed target rows
    e->xl = false;
    if (allow_xl && !e->knobs.no_xl) {
        int extra = 0;
        for (int l = 0; l < L; ++l) extra += CT * (e->npad[l] + kLdPad) + (l >= 1 ? e->npad[l] : CT * (e->npad[0] + kLdPad));
        const int ywords = (e->out_pad + 31) / 32;
        if (e->has_head) extra += e->out_pad + (CT * ywords + 3) / 4 * 4;
        if ((off + extra) * (int)sizeof(float) <= 160 * 1024) {
            for (int l = 0; l < L; ++l) { e->lds_x[l] = off; off += CT * (e->npad[l] + kLdPad); }
            for (int l = 0; l < L; ++l) { e->lds_bias[l] = off; off += l >= 1 ? e->npad[l] : CT * (e->npad[0] + kLdPad); }
            if (e->has_head) { e->lds_hbias = off; off += e->out_pad; e->lds_yw = off; off += (CT * ywords + 3) / 4 * 4; }
            e->xl = true;
        }
    }
    e->lds_bytes = off * (int)sizeof(float);
    if (e->lds_bytes > 160 * 1024) return fail(MCPC_ENOMEM, "in-place schedule does not fit the LDS (%d bytes)", e->lds_bytes);
    return 0;
}

// Table of the in-place wave-specialised kernel (mcpc_steps_ws2.h), R = ring size.  One step =
//   read-out:  HF(0) .. HF(R-1) HB(0) HF(R) HB(1) ...   (FWD_0, which has no GEMM, slipped in behind the first HB);
//   forward:   FWD_{L-1} ... FWD_1   (their outputs E_l share LDS with the ring, so they follow the last HB);
//   updates:   BWD_{L-1} (accb hand-off), BWD_0 ... BWD_{L-2};  energy reduction.
// Read dependencies, waited for in front of the GEMM (dep_e, "all E waves past entry"):
//   HF(c) <- last BWD_{L-1} of the PREVIOUS step (FX_{L-1});  HB(c) <- HF(c);  FWD_l <- last BWD_{l-1} of the previous
//   step (FX_{l-1});  BWD_l GEMM <- last FWD_{l+1} (E_{l+1}).
// Write-after-read dependencies, waited for behind the GEMM, before the block is stored:
//   dep_g ("all G waves past entry"):  HF(c) <- HB(c-R) (ring slot);  HF(c) in a slot that overlaps the E_l <- last GEMM
//     of the previous step that reads an E_l (BWD_{L-2});  FWD_l <- HB(last);
//   dep_se ("all E waves past entry"): HF(c) in a slot that overlaps the E_l <- the last BWD entry of the previous step
//     (its epilogue loads read E_l).
// The others are implied: a G wave that stores into FX_l has just waited for epilogues that can only have run after every
// G wave finished the GEMMs that read the old contents (the BWD_{L-1} hand-off after HB(last) <- HF(last) epilogues, BWD_l
// after the FWD_{l+1} epilogues, which follow FWD_{l+1}'s GEMM over FX_l).
//
// 16-chain plans whose LDS holds the ring AND the E_l side by side (plan_lds_ws2: ws2_overlay == false) order a step differently:
//   read-out:  HF(0) .. HF(R-1) HB(0) FWD_1 HF(R) HB(1) FWD_0 HB(2) FWD_2 ...   (every forward entry fills a gap of the read-out)
//   updates:   as above.
// With 16 chains a GEMM is half as long, the epilogues are not, and the shared region cost twice: HF(0) / HF(1) waited for the LAST
// epilogue of the previous step (their slots overlapped the E_l it reads) and the forward GEMMs sat behind the read-out where
// nothing hid their epilogues.  Apart, HF(c < R) needs no write-after-read wait at all, and FWD_l waits -- behind its GEMM -- for
// the previous step's readers of E_l: dep_g <- last back-projection GEMM, dep_se <- last BWD entry.
int build_phases_ws2(mcpc_engine* e) {
    const int L = e->L;
    const int span = kWs2Pairs * ws2_nt<1>();     // tiles per table entry
    auto tiles = [&](int l) { return e->npad[l] / 16; };
    auto blank = [&]() { KPhase k{}; k.dep_e = -1; k.dep_g = -1; k.dep_se = -1; k.b_row = -1; k.o_row = -1; return k; };
    enum { REF_LAST_BWD = -1000, REF_LAST_FWD = -2000, REF_LAST_HB = -3000, REF_LAST_BWD_GEMM = -4000, REF_LAST_BWD_ANY = -5000 };   // symbolic deps
    auto fwd_entries = [&](int l, std::vector<KPhase>& out) {
        for (int base = 0; base < tiles(l); base += span) {
            KPhase k = blank();
            k.type = PH_FWD; k.layer = l; k.tile0 = base; k.ntiles = std::min(span, tiles(l) - base);
            if (l == 0) {
                k.flags = PHF_MU1 | PHF_WS_EPI;
            } else {
                k.A = e->lin[l].Wf; k.a_lin = l; k.kw = 16 * tiles(l - 1); k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
                k.b_lds = e->lds_a[l - 1]; k.ldb = e->npad[l - 1] + kLdPad;
                k.out_lds = e->lds_e[l]; k.out_ld = e->npad[l] + kLdPad;
                k.b_row = rowexp_fx(l - 1); k.o_row = rowexp_e(l);
                k.flags = PHF_WS_GEMM | PHF_WS_EPI; k.dep_e = REF_LAST_BWD - (l - 1);
                if (e->has_head && e->ws2_overlay) k.dep_g = REF_LAST_HB;       // E_l shares LDS with the ring
                else if (e->has_head) {
                    // E_l has rows of its own and the entry runs between the read-out chunks: what still reads the OLD E_l are
                    // the previous step's back-projection GEMMs (all G waves: E_l is a B operand) and x updates (E waves)
                    k.dep_g = REF_LAST_BWD_GEMM; k.dep_se = REF_LAST_BWD_ANY;
                }
            }
            out.push_back(k);
        }
    };
    // FWD_0 has no GEMM and no LDS output: it fills a gap of the read-out; the others follow the read-out
    std::vector<KPhase> fill, after;
    if (e->has_head && !e->ws2_overlay) {
        // every forward entry fills a gap of the read-out, in the order their inputs become ready: the previous step ends with
        // the x updates BWD_{L-1}, BWD_0, BWD_1 ... BWD_{L-2}, so FWD_1 (needs f(x_0)) first, FWD_0 (no GEMM), then FWD_2 ... FWD_{L-1}
        if (L >= 2) fwd_entries(1, fill);
        fwd_entries(0, fill);
        for (int l = 2; l < L; ++l) fwd_entries(l, fill);
    } else {
        for (int l = L - 1; l >= 0; --l) fwd_entries(l, (l >= 1 && e->has_head) ? after : fill);
    }
    std::vector<KPhase> ph;
    size_t nf = 0;
    if (e->has_head) {
        const int hc = e->ws2_chunk, R = e->ws2_ring;        // hc: widest chunk = ring slot width
        const int ht = e->out_pad / 16;
        const int nch = (ht + hc - 1) / hc;
        // chunk c = tiles [c_start[c], c_start[c+1]): whole k-blocks of the back-projection (tq tiles each) dealt out evenly, the last
        // chunk clipped to the read-out's width
        const int tq = kKB / 16, hb = (ht + tq - 1) / tq;
        std::vector<int> c_start(nch + 1, 0);
        for (int c = 0; c < nch; ++c) c_start[c + 1] = std::min(ht, c_start[c] + tq * (hb / nch + (c < hb % nch ? 1 : 0)));
        const int chunk_floats = e->ct * (hc * 16 + kLdPad);
        int e_sum = 0;
        for (int l = 1; l < L; ++l) e_sum += e->ct * (e->npad[l] + kLdPad);
        std::vector<int> idx_f(nch, -1), idx_b(nch, -1);
        auto add_f = [&](int c) {
            KPhase f = blank();
            f.type = PH_HEADF; f.layer = L - 1; f.tile0 = c_start[c]; f.ntiles = c_start[c + 1] - c_start[c]; f.rot = c & (kWs2Pairs - 1);
            f.A = e->lin[L].Wf; f.a_lin = L; f.kw = 16 * tiles(L - 1); f.nkb = kblocks(f.kw); f.a_tile_stride = f.nkb * kFragBlock;
            f.b_lds = e->lds_a[L - 1]; f.ldb = e->npad[L - 1] + kLdPad;
            f.out_lds = e->lds_eo + (c % R) * chunk_floats; f.out_ld = hc * 16 + kLdPad;
            f.b_row = rowexp_fx(L - 1); f.o_row = rowexp_ring(c % R);
            f.flags = PHF_WS_GEMM | PHF_WS_EPI; f.dep_e = REF_LAST_BWD - (L - 1);
            if (c >= R) f.dep_g = idx_b[c - R];
            else if (e->ws2_overlay && (c % R) * chunk_floats < e_sum) { f.dep_g = REF_LAST_BWD_GEMM; f.dep_se = REF_LAST_BWD_ANY; }   // slot overlaps the E_l
            idx_f[c] = (int)ph.size(); ph.push_back(f);
        };
        auto add_b = [&](int c) {
            KPhase b = blank();
            b.type = PH_HEADB; b.layer = L - 1; b.tile0 = 0; b.ntiles = tiles(L - 1);
            b.A = e->lin[L].Wb; b.a_lin = L; b.a_tile_stride = kblocks(e->out_pad) * kFragBlock; b.a_off0 = (c_start[c] / tq) * kFragBlock;
            b.kw = 16 * (c_start[c + 1] - c_start[c]); b.nkb = (c_start[c + 1] - c_start[c] + tq - 1) / tq;
            b.b_lds = e->lds_eo + (c % R) * chunk_floats; b.ldb = hc * 16 + kLdPad;
            b.b_row = rowexp_ring(c % R);
            b.flags = PHF_WS_GEMM; b.dep_e = idx_f[c];
            idx_b[c] = (int)ph.size(); ph.push_back(b);
            if (nf < fill.size()) ph.push_back(fill[nf++]);      // one forward entry behind every back-projection
        };
        for (int c = 0; c < nch; ++c) {
            add_f(c);
            if (c >= R - 1) add_b(c - (R - 1));
        }
        for (int c = std::max(nch - (R - 1), 0); c < nch; ++c) add_b(c);
    }
    while (nf < fill.size()) ph.push_back(fill[nf++]);
    // bottom-up (FWD_1 first): the small GEMMs go first, so that the epilogue of FWD_1 has FWD_2's GEMM to hide behind
    for (auto it = after.rbegin(); it != after.rend(); ++it) ph.push_back(*it);
    // x updates: BWD_{L-1} first (its back-projection is complete: accb), then bottom-up from BWD_0, so that BWD_{L-2},
    // which reads the E_{L-1} produced last, comes last
    for (int base = 0; base < tiles(L - 1); base += span) {
        KPhase k = blank();
        k.type = PH_BWD; k.layer = L - 1; k.tile0 = base; k.ntiles = std::min(span, tiles(L - 1) - base);
        k.flags = PHF_WS_EPI | (e->has_head ? PHF_WS2_HANDOFF : 0);
        k.sign = e->has_head ? 1.0f : 0.0f;
        k.out_lds = e->lds_a[L - 1]; k.out_ld = e->npad[L - 1] + kLdPad;
        k.o_row = rowexp_fx(L - 1);
        ph.push_back(k);
    }
    for (int l = 1; l <= L - 1; ++l)
        for (int base = 0; base < tiles(l - 1); base += span) {
            KPhase k = blank();
            k.type = PH_BWD; k.layer = l - 1; k.tile0 = base; k.ntiles = std::min(span, tiles(l - 1) - base);
            k.A = e->lin[l].Wb; k.a_lin = l; k.kw = 16 * tiles(l); k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
            k.b_lds = e->lds_e[l]; k.ldb = e->npad[l] + kLdPad; k.sign = -1.0f;
            k.out_lds = e->lds_a[l - 1]; k.out_ld = e->npad[l - 1] + kLdPad;
            k.b_row = rowexp_e(l); k.o_row = rowexp_fx(l - 1);
            k.flags = PHF_WS_GEMM | PHF_WS_EPI; k.dep_e = REF_LAST_FWD - l;
            ph.push_back(k);
        }
    { KPhase k = blank(); k.type = PH_ENERGY; k.flags = PHF_WS_EPI; ph.push_back(k); }
    // resolve the symbolic dependencies
    std::vector<int> last_fwd(L, -1), last_bwd(L, -1);
    int last_hb = -1, last_bwd_gemm = -1, last_bwd_any = -1;
    for (size_t i = 0; i < ph.size(); ++i) {
        if (ph[i].type == PH_FWD) last_fwd[ph[i].layer] = (int)i;
        if (ph[i].type == PH_BWD) { last_bwd[ph[i].layer] = (int)i; last_bwd_any = (int)i; if (ph[i].flags & PHF_WS_GEMM) last_bwd_gemm = (int)i; }
        if (ph[i].type == PH_HEADB) last_hb = (int)i;
    }
    auto resolve = [&](int d) {
        if (d == REF_LAST_BWD_ANY) return last_bwd_any;
        if (d == REF_LAST_BWD_GEMM) return last_bwd_gemm;
        if (d == REF_LAST_HB) return last_hb;
        if (d <= REF_LAST_FWD && d > REF_LAST_HB) return last_fwd[REF_LAST_FWD - d];
        if (d <= REF_LAST_BWD && d > REF_LAST_FWD) return last_bwd[REF_LAST_BWD - d];
        return d;
    };
    for (auto& k : ph) { k.dep_e = resolve(k.dep_e); k.dep_g = resolve(k.dep_g); k.dep_se = resolve(k.dep_se); }
    // the GEMM waves walk only the entries they have work in (a GEMM, or the hand-off of the read-out's back-projection): FWD_0, the
    // energy entry and x updates without a back-projection cost them a table round each for nothing
    auto g_works = [&](const KPhase& k) { return (k.flags & (PHF_WS_GEMM | PHF_WS2_HANDOFF)) != 0; };
    e->g_first = -1;
    for (size_t i = 0; i < ph.size(); ++i) if (g_works(ph[i])) { e->g_first = (int)i; break; }
    for (size_t i = 0; i < ph.size(); ++i) {
        ph[i].next_g = e->g_first;
        for (size_t d = 1; d <= ph.size(); ++d) {
            const size_t j = (i + d) % ph.size();
            if (g_works(ph[j])) { ph[i].next_g = (int)j; break; }
        }
    }
    int rc = dmalloc(e->phases, ph.size());
    if (rc) return rc;
    if (hipMemcpy(e->phases, ph.data(), ph.size() * sizeof(KPhase), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MCPC_EHIP, "hipMemcpy of the phase table failed");
    e->n_phases = (int)ph.size();
    return 0;
}


// ---- unified-wave kernel (mcpc_steps_u.h) ---------------------------------------------------------------------------------------------
// LDS plan: FX_l, E_l, the WHOLE read-out error e_o [16][out_pad], the state rows X_l and the per-step constants of the workgroup's 16
// chains.  No plan (u.ok == false, not an error) when that does not fit 160 KiB: such networks run on the in-place kernel.
int plan_lds_u(mcpc_engine* e) {
    UPlan& u = e->u;
    u.ok = false;
    const int CT = e->ct, L = e->L;
    // two tries: everything the epilogues read per step in LDS; else the bit-packed target rows stay in global memory (the read-out's
    // epilogue requests its words in front of its row's GEMM: mcpc_steps_u.h) -- cfg-M's plan is 1 792 bytes over the 160 KiB with them
    // and fits to the byte without
    for (int yw_in_lds = 1; yw_in_lds >= 0 && !u.ok; --yw_in_lds) {
        int off = 0;
        for (int l = 0; l < L; ++l) { u.lds_a[l] = off; off += CT * (e->npad[l] + kLdPad); }
        u.lds_e[0] = 0;
        for (int l = 1; l < L; ++l) { u.lds_e[l] = off; off += CT * (e->npad[l] + kLdPad); }
        u.lds_red = off; off += 2 * (kMaxLatent + 1) * kMaxWaves;
        u.lds_ws_sync = off;                                // (no progress counters in this kernel)
        u.lds_eo = off;
        if (e->has_head) off += CT * (e->out_pad + kLdPad);
        u.lds_zero = off; off += 16;
        u.lds_spillmax = off; off += kSpillTensors;
        u.lds_rowexp = off; off += (rowexp_ring(0) + 1) * 16;      // the ids this kernel uses: FX_l, E_l and ONE read-out row word
        for (int l = 0; l < L; ++l) { u.lds_x[l] = off; off += CT * (e->npad[l] + kLdPad); }
        for (int l = 0; l < L; ++l) { u.lds_bias[l] = off; off += l >= 1 ? e->npad[l] : CT * (e->npad[0] + kLdPad); }
        u.lds_yw = -1;
        if (e->has_head) {
            const int ywords = (e->out_pad + 31) / 32;
            u.lds_hbias = off; off += e->out_pad;
            if (yw_in_lds) { u.lds_yw = off; off += (CT * ywords + 3) / 4 * 4; }
        }
        u.lds_bytes = off * (int)sizeof(float);
        u.ok = u.lds_bytes <= 160 * 1024;
    }
    return 0;
}

// Tables of the unified-wave kernel: every wave walks its OWN rows (u.phases[w * n_phases + p]).  One step = two levels, each opened by a
// workgroup barrier:
//   forward:  read-out tiles (HEADF: out, loss error -> e_o), FWD_{L-1} .. FWD_1 (prediction errors E_l), FWD_0 -- they read the FX_l the
//             previous step's x updates left and write e_o / E_l;
//   updates:  BWD_{L-1} (GEMM over the whole e_o), BWD_{L-2} .. BWD_0 (GEMM over E_{l+1}) -- they read e_o / E_l and write X_l, FX_l.
// Inside a level the jobs are independent, and with a barrier on either side no tile belongs to a wave: a JOB is up to four consecutive
// unit tiles of one entry (one GEMM call + one epilogue call of a wave), and the jobs of a level are dealt to the eight waves by cost,
// longest first (a cost model in cycles: fixed cost per row, k-blocks x (operand split + MFMAs per tile), epilogue per tile).  Why four
// tiles where the work allows: the operand split (24 VALU instructions per k-block) and the row's fixed costs are shared by the job's
// tiles -- a GEMM of 8 tiles as 2 jobs of 4 splits its B operand twice, as 8 jobs of 1 eight times -- and heavy entries (the read-out's
// back-projection: K = n_out) are cut finer only as far as the level's balance needs.  (The one exception: the running sum of e_1 lives in
// registers of wave w for tile w of the top layer -- lean_load_e0 -- so FWD_0's tiles are pinned when that layer has at most 8.)
int build_phases_u(mcpc_engine* e) {
    UPlan& u = e->u;
    const int L = e->L;
    auto tiles = [&](int l) { return e->npad[l] / 16; };
    auto blank = [&]() { KPhase k{}; k.dep_e = -1; k.dep_g = -1; k.dep_se = -1; k.b_row = -1; k.o_row = -1; k.next_g = -1; k.rot = 1; return k; };
    struct Job { KPhase k; double cost; int pin; };
    // cost model (shader cycles per wave; calibrated on profiles/r06_small_net.txt).  Per k-block of a row's GEMM: 1 tile ~400, 2 tiles ~470,
    // 4 tiles ~600 -- the B split, the fragment requests and, with one tile, three dependent MFMAs -- and ~200 less when the operand arrives
    // in planes (`ps`: the read-out's back-projection; the table is built before the loss is known and assumes the Bernoulli read-out the
    // reference trains with); ~1200 before the first block; per row ~2000 for its descriptor, the next row's fragment requests and the
    // epilogue's fixed part; per tile of an epilogue: read-out ~1000, x update with the Philox kick ~1300, prediction error ~500.
    const Knobs& kn = e->knobs;
    auto gemm_cost = [&](int nt, int nkb, bool ps) { return nkb > 0 ? (double)kn.u_gemm0 + nkb * ((double)std::max(kn.u_kb - (ps ? 200 : 0), 0) + (double)kn.u_kbt * nt) : 0.0; };
    const double row_cost = (double)kn.u_row;
    auto make_jobs = [&](const KPhase& proto, int nt_total, double epi_tile, int g, std::vector<Job>& out, bool pinned, bool ps) {
        if (pinned) g = 1;
        for (int t = 0; t < nt_total; t += g) {
            Job j; j.k = proto; j.k.tile0 = t; j.k.ntiles = std::min(g, nt_total - t); j.k.rot = 1;
            if (proto.type == PH_HEADF) j.k.out_lds = proto.out_lds + 16 * t;       // (the epilogue writes columns relative to its row's first tile)
            j.cost = row_cost + gemm_cost(j.k.ntiles, (proto.flags & PHF_WS_GEMM) ? proto.nkb : 0, ps) + j.k.ntiles * epi_tile;
            j.pin = pinned ? t : -1;
            out.push_back(j);
        }
    };
    struct Entry { KPhase k; int ntiles; double epi_tile; bool pinned; bool ps; };
    std::vector<Entry> level[2];
    if (e->has_head) {
        KPhase f = blank();
        f.type = PH_HEADF; f.layer = L - 1;
        f.A = e->lin[L].Wf; f.a_lin = L; f.kw = 16 * tiles(L - 1); f.nkb = kblocks(f.kw); f.a_tile_stride = f.nkb * kFragBlock;
        f.b_lds = u.lds_a[L - 1]; f.ldb = e->npad[L - 1] + kLdPad;
        f.out_lds = u.lds_eo; f.out_ld = e->out_pad + kLdPad;            // (row-relative columns: the epilogue adds 16 (tile - tile0) to its row's base)
        f.b_row = rowexp_fx(L - 1); f.o_row = rowexp_ring(0);
        f.flags = PHF_WS_GEMM | PHF_WS_EPI;
        level[0].push_back({f, e->out_pad / 16, (double)kn.u_eh, false, false});
    }
    for (int l = L - 1; l >= 0; --l) {
        KPhase k = blank();
        k.type = PH_FWD; k.layer = l;
        if (l == 0) {
            k.flags = PHF_MU1 | PHF_WS_EPI;
        } else {
            k.A = e->lin[l].Wf; k.a_lin = l; k.kw = 16 * tiles(l - 1); k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
            k.b_lds = u.lds_a[l - 1]; k.ldb = e->npad[l - 1] + kLdPad;
            k.out_lds = u.lds_e[l]; k.out_ld = e->npad[l] + kLdPad;
            k.b_row = rowexp_fx(l - 1); k.o_row = rowexp_e(l);
            k.flags = PHF_WS_GEMM | PHF_WS_EPI;
        }
        level[0].push_back({k, tiles(l), (double)kn.u_ef, l == 0 && tiles(0) <= kUWaves, false});
    }
    for (int l = L - 1; l >= 0; --l) {
        KPhase k = blank();
        k.type = PH_BWD; k.layer = l;
        k.out_lds = u.lds_a[l]; k.out_ld = e->npad[l] + kLdPad; k.o_row = rowexp_fx(l);
        k.flags = PHF_WS_EPI;
        if (l == L - 1) {
            k.sign = e->has_head ? 1.0f : 0.0f;
            if (e->has_head) {
                k.A = e->lin[L].Wb; k.a_lin = L; k.kw = e->out_pad; k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
                k.b_lds = u.lds_eo; k.ldb = e->out_pad + kLdPad; k.b_row = rowexp_ring(0);
                k.flags |= PHF_WS_GEMM;
            }
        } else {
            k.A = e->lin[l + 1].Wb; k.a_lin = l + 1; k.kw = 16 * tiles(l + 1); k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
            k.b_lds = u.lds_e[l + 1]; k.ldb = e->npad[l + 1] + kLdPad; k.b_row = rowexp_e(l + 1); k.sign = -1.0f;
            k.flags |= PHF_WS_GEMM;
        }
        level[1].push_back({k, tiles(l), (double)kn.u_eb, false, l == L - 1 && e->has_head && e->out_pad > kShortK * kKB});
    }
    std::vector<KPhase> rows[kUWaves];
    for (int lv = 0; lv < 2; ++lv) {
        // the grain of every entry (4, 2 or 1 tiles per job) by exhaustive search: the combination whose longest-first deal has the
        // shortest makespan (at most 7 entries per level: 3^7 deals of a few dozen jobs)
        const int ne = (int)level[lv].size();
        std::vector<int> grain(ne, 4), best_grain(ne, 4);
        double best_span = 1e300;
        std::vector<Job> jobs;
        auto deal = [&](const std::vector<int>& gr, std::vector<KPhase>* mine) {
            jobs.clear();
            for (int i = 0; i < ne; ++i) make_jobs(level[lv][i].k, level[lv][i].ntiles, level[lv][i].epi_tile, gr[i], jobs, level[lv][i].pinned, level[lv][i].ps);
            std::stable_sort(jobs.begin(), jobs.end(), [](const Job& a, const Job& b) { return (a.pin >= 0) != (b.pin >= 0) ? a.pin >= 0 : a.cost > b.cost; });
            double load[kUWaves] = {0};
            for (auto& j : jobs) {
                int w = 0;
                if (j.pin >= 0) w = j.pin;
                else for (int i = 1; i < kUWaves; ++i) if (load[i] < load[w]) w = i;
                load[w] += j.cost;
                if (mine) mine[w].push_back(j.k);
            }
            double span = 0, sum = 0;
            for (double v : load) { span = std::max(span, v); sum += v; }
            return span + 1e-3 * sum;               // (ties: the deal with less work in total)
        };
        int combos = 1;
        for (int i = 0; i < ne; ++i) combos *= 3;
        for (int cidx = 0; cidx < combos; ++cidx) {
            int c = cidx;
            for (int i = 0; i < ne; ++i) { grain[i] = 4 >> (c % 3); c /= 3; }
            const double span = deal(grain, nullptr);
            if (span < best_span) { best_span = span; best_grain = grain; }
        }
        std::vector<KPhase> mine[kUWaves];
        (void)deal(best_grain, mine);
        for (int w = 0; w < kUWaves; ++w) {
            if (mine[w].empty()) { KPhase k = blank(); k.type = PH_NOP; mine[w].push_back(k); }
            mine[w][0].flags |= PHF_SYNC;                        // the level's barrier
            rows[w].insert(rows[w].end(), mine[w].begin(), mine[w].end());
        }
    }
    // the four fragment slots a row's GEMM starts from (u_prefetch), resolved here: offsets in 16-byte units from the row's A, -1 = none.
    // (dep_e, dep_g, dep_se, next_g carry them: the unified-wave kernel has no other use for those fields)
    for (int w = 0; w < kUWaves; ++w)
        for (auto& k : rows[w]) {
            int slot[4] = {-1, -1, -1, -1};
            const int nt = std::min(k.ntiles, kUNT);
            if (nt > 0 && (k.flags & PHF_WS_GEMM) && k.nkb > 0)
                for (int sl = 0; sl < 4; ++sl) {
                    const bool deep = nt <= 2;
                    const int ti = deep ? (nt == 2 ? (sl & 1) : 0) : sl, kb = deep ? (nt == 2 ? (sl >> 1) : sl) : 0;
                    if (ti < nt && kb < k.nkb) slot[sl] = (k.tile0 + k.rot * ti) * k.a_tile_stride + k.a_off0 + kb * kFragBlock;
                }
            k.dep_e = slot[0]; k.dep_g = slot[1]; k.dep_se = slot[2]; k.next_g = slot[3];
        }
    size_t R = 0;
    for (int w = 0; w < kUWaves; ++w) R = std::max(R, rows[w].size());
    std::vector<KPhase> ph;
    for (int w = 0; w < kUWaves; ++w) {
        while (rows[w].size() < R) { KPhase k = blank(); k.type = PH_NOP; rows[w].push_back(k); }
        ph.insert(ph.end(), rows[w].begin(), rows[w].end());
    }
    int rc = dmalloc(u.phases, ph.size());
    if (rc) return rc;
    if (hipMemcpy(u.phases, ph.data(), ph.size() * sizeof(KPhase), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MCPC_EHIP, "hipMemcpy of the phase table failed");
    u.n_phases = (int)R;
#ifdef MCPC_STAMPS
    {
        static const char* tn[6] = {"FWD", "HEADF", "HEADB", "BWD", "ENERGY", "NOP"};
        for (int w = 0; w < kUWaves; ++w) {
            fprintf(stderr, "[u-table] wave %d:", w);
            for (size_t i = 0; i < R; ++i) {
                const KPhase& k = rows[w][i];
                fprintf(stderr, " %s%s(l%d t%d+%d kb%d)", (k.flags & PHF_SYNC) ? "|" : "", tn[k.type], k.layer, k.tile0, k.ntiles, (k.flags & PHF_WS_GEMM) ? k.nkb : 0);
            }
            fprintf(stderr, "\n");
        }
    }
#endif
    return 0;
}

// The automatic choice between the in-place and the unified-wave kernel for an engine whose unified plan fits (tuning ws=2 / ws=3 force
// either).  Measured on one MI355X (profiles/r06_small_net.txt, us per 16-chain unit-step, MCPC / MAP / learning call):
//   20-128-128-784 (476 tile-blocks of GEMM per step)   in-place 21.3 / 23.1 / 23.9    unified 16.8 / 18.0 / 19.3
//   30-200-200-784 (877)                                  in-place 25.3                  unified 23.2            (MCPC)
//   30-224-224-784 (917)                                  in-place 25.5                  unified 26.8
//   30-256-256-784 (1 080 tile-blocks: cfg-M)            in-place 26.4 / 30.0 / 29.4    unified 27.9-29.3 / 31.0 / 32.4
// A step's fixed costs per table entry are what the unified form removes; the overlap of GEMM and epilogue waves is what it gives up, and
// at cfg-M's width that overlap is worth more.  The unit is what both scale with: (unit tile, 32-deep k-block) pairs of all GEMMs of a step.
// A ZERO-LOSS call (unclamped generation) runs on the unified kernel whatever the width: only that kernel skips the read-out on the steps
// nobody records (cfg-M's net: 16.6 against 25.3 us per step) -- mcpc_run decides that per run.
int gemm_tile_blocks(const mcpc_engine* e) {
    int n = 0;
    for (int l = 1; l < e->L; ++l) n += (e->npad[l] / 16) * kblocks(e->npad[l - 1]) + (e->npad[l - 1] / 16) * kblocks(e->npad[l]);
    if (e->has_head) n += (e->out_pad / 16) * kblocks(e->npad[e->L - 1]) + (e->npad[e->L - 1] / 16) * kblocks(e->out_pad);
    return n;
}
bool choose_unified(const mcpc_engine* e) { return gemm_tile_blocks(e) <= 900; }

// The per-step schedule: every GEMM of a Langevin step with its operands, the epilogue that follows
// it and the barrier it needs.  Output tiles are handed out 16 at a time (4 waves x kNT tiles).
int build_phases(mcpc_engine* e) {
    std::vector<KPhase> ph;
    const int L = e->L;
    const int span = kNT * kWaves;
    auto tiles = [&](int l) { return e->npad[l] / 16; };
    // top latent layer: its prediction is the constant mu1, no GEMM
    for (int base = 0; base < tiles(0); base += span) {
        KPhase k{};
        k.type = PH_FWD; k.layer = 0; k.tile0 = base; k.ntiles = std::min(span, tiles(0) - base);
        k.flags = PHF_MU1 | (base + span >= tiles(0) ? PHF_SYNC : 0);
        ph.push_back(k);
    }
    for (int l = 1; l < L; ++l)
        for (int base = 0; base < tiles(l); base += span) {
            KPhase k{};
            k.type = PH_FWD; k.layer = l; k.tile0 = base; k.ntiles = std::min(span, tiles(l) - base);
            k.A = e->lin[l].Wf; k.a_lin = l; k.kw = 16 * tiles(l - 1); k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
            k.b_lds = e->lds_a[l - 1]; k.ldb = e->npad[l - 1] + kLdPad;
            k.flags = base + span >= tiles(l) ? PHF_SYNC : 0;
            ph.push_back(k);
        }
    if (e->has_head) {
        const int ht = e->out_pad / 16;
        for (int c0 = 0; c0 < ht; c0 += kChunkTiles) {
            const int ntc = std::min(kChunkTiles, ht - c0);
            KPhase f{};
            f.type = PH_HEADF; f.layer = L - 1; f.tile0 = c0; f.ntiles = ntc;
            f.A = e->lin[L].Wf; f.a_lin = L; f.kw = 16 * tiles(L - 1); f.nkb = kblocks(f.kw); f.a_tile_stride = f.nkb * kFragBlock;
            f.b_lds = e->lds_a[L - 1]; f.ldb = e->npad[L - 1] + kLdPad; f.flags = PHF_SYNC;
            f.out_lds = e->lds_eo; f.out_ld = kChunkTiles * 16 + kLdPad; f.dep_e = f.dep_g = -1;
            ph.push_back(f);
            KPhase b{};
            b.type = PH_HEADB; b.layer = L - 1; b.tile0 = 0; b.ntiles = tiles(L - 1);
            b.A = e->lin[L].Wb; b.a_lin = L; b.a_tile_stride = kblocks(e->out_pad) * kFragBlock; b.a_off0 = (c0 * 16 / kKB) * kFragBlock;
            b.kw = ntc * 16; b.nkb = (ntc * 16 + kKB - 1) / kKB;
            b.b_lds = e->lds_eo; b.ldb = kChunkTiles * 16 + kLdPad;
            b.flags = PHF_ACC_FROM_B | PHF_ACC_TO_B | PHF_SYNC;
            ph.push_back(b);
        }
    }
    { KPhase k{}; k.type = PH_ENERGY; ph.push_back(k); }
    // x updates, bottom-up: the last latent layer first (its back-projection sits in accb)
    for (int base = 0; base < tiles(L - 1); base += span) {
        KPhase k{};
        k.type = PH_BWD; k.layer = L - 1; k.tile0 = base; k.ntiles = std::min(span, tiles(L - 1) - base);
        k.flags = e->has_head ? PHF_ACC_FROM_B : 0;
        k.sign = e->has_head ? 1.0f : 0.0f;
        ph.push_back(k);
    }
    for (int l = L - 1; l >= 1; --l)
        for (int base = 0; base < tiles(l - 1); base += span) {
            KPhase k{};
            k.type = PH_BWD; k.layer = l - 1; k.tile0 = base; k.ntiles = std::min(span, tiles(l - 1) - base);
            k.A = e->lin[l].Wb; k.a_lin = l; k.kw = 16 * tiles(l); k.nkb = kblocks(k.kw); k.a_tile_stride = k.nkb * kFragBlock;
            k.b_lds = e->lds_e[l]; k.ldb = e->npad[l] + kLdPad; k.sign = -1.0f;
            ph.push_back(k);
        }
    int rc = dmalloc(e->phases, ph.size());
    if (rc) return rc;
    if (hipMemcpy(e->phases, ph.data(), ph.size() * sizeof(KPhase), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MCPC_EHIP, "hipMemcpy of the phase table failed");
    e->n_phases = (int)ph.size();
    return 0;
}

}  // namespace

namespace {

// Round schedule of the in-place kernel for a shard of U 16-chain units on C < U CUs.  One unit per CU is what the kernel is built
// for (a step is a chain of dependent hand-overs inside ONE workgroup: a second workgroup per CU does not fit the LDS, a launch of
// U > C workgroups runs as ceil(U / C) hardware rounds, the last one mostly empty).  Instead the units are dealt into k groups and a
// CYCLE is k launches of q steps; launch i runs groups i .. i+m-1 (mod k): every unit takes part in m of the k launches, in
// order, and after the cycle every unit has advanced m q steps -- k / m launch times per m q steps where the hardware rounds need
// ceil(U / C).  6000 chains = 375 units on 256 CUs: k = 3, m = 2, 250 workgroups per launch, 1.5 launch times per step instead of 2.
// (k, m): the smallest k within 3 % of the smallest k / m over k <= 16 whose launches fit C workgroups.  Chains are independent, so the trajectories are those
// of any other schedule, bitwise; in a Hebbian segment (m q <= slots of a ring part) every unit fills its own rows of all m q slots
// before the flush, which therefore sees what the plain schedule would have written.
int setup_rounds(mcpc_engine* e, int n_cu) {
    const int U = e->nwg_live, C = n_cu - e->knobs.cu_slack;
    if (U <= C || C < 1) return 0;
    auto gsize = [&](int k, int g) { return (int)((int64_t)(g + 1) * U / k - (int64_t)g * U / k); };
    // best m for every k <= 16, then the SMALLEST k within 3 % of the best k / m: short cycles mean long launches (a Hebbian segment is
    // one cycle of at most `half_slots` steps) -- 300 units: (6, 5) at 1.200 rather than (13, 11) at 1.182
    int bk = 0, bm = 0, mk[17] = {0};
    for (int k = 2; k <= 16; ++k)
        for (int m = k - 1; m >= 1; --m) {
            int worst = 0;
            for (int i = 0; i < k; ++i) { int s = 0; for (int j = 0; j < m; ++j) s += gsize(k, (i + j) % k); worst = std::max(worst, s); }
            if (worst > C) continue;
            mk[k] = m;
            if (!bk || (int64_t)k * bm < (int64_t)bk * m) { bk = k; bm = m; }
            break;
        }
    for (int k = 2; bk && k < bk; ++k)
        if (mk[k] && 100.0 * k * bm <= 103.0 * bk * mk[k]) { bk = k; bm = mk[k]; break; }
    if (!bk) { bk = (U + C - 1) / C; bm = 1; }            // more than 16 rounds: plain rounds of at most C units
    const int k = bk, m = bm;
    std::vector<int> tab;
    e->rr_count.assign(k, 0); e->rr_off.assign(k, 0);
    std::vector<int> done(k, 0);                              // launches of this cycle a group has taken part in
    for (int i = 0; i < k; ++i) {
        std::vector<int> ids, rel;
        for (int j = 0; j < m; ++j) {
            const int g = (i + j) % k;
            for (int u = (int)((int64_t)g * U / k); u < (int)((int64_t)(g + 1) * U / k); ++u) { ids.push_back(u); rel.push_back(done[g]); }
        }
        for (int j = 0; j < m; ++j) ++done[(i + j) % k];
        e->rr_off[i] = (int)tab.size(); e->rr_count[i] = (int)ids.size();
        tab.insert(tab.end(), ids.begin(), ids.end());
        tab.insert(tab.end(), rel.begin(), rel.end());
    }
    for (int g = 0; g < k; ++g)
        if (done[g] != m) return fail(MCPC_EINVAL, "round schedule: unbalanced cycle");
    int rc = dmalloc(e->rr_tab, tab.size());
    if (rc) return rc;
    if (hipMemcpy(e->rr_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MCPC_EHIP, "hipMemcpy of the round-schedule tables failed");
    if (hipFuncSetAttribute((const void*)mcpc_steps_ws2_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes) != hipSuccess)
        return fail(MCPC_EHIP, "hipFuncSetAttribute failed for the round schedule");
    e->rr_k = k; e->rr_m = m; e->rr = true;
    e->u_rr_name = "mcpc::mcpc_steps_u_kernel<true> (round schedule: k=" + std::to_string(k) + " launches per cycle, every 16-chain unit in m=" + std::to_string(m) + " of them)";
    e->rr_name = "mcpc::mcpc_steps_ws2_kernel<1, true> (round schedule: k=" + std::to_string(k) + " launches per cycle, every 16-chain unit in m=" + std::to_string(m) + " of them)";
    return 0;
}

}  // namespace

extern "C" {

int mcpc_abi_version(void) { return MCPC_ABI_VERSION; }

const char* mcpc_build_info(void) {
    // exp: the umbrella of mcpc_build.h, and -- should a switch ever be added without being listed there -- any -DMCPC_EXP / MCPC_HEB_EXP
    // in the flags the Makefile recorded
    static const bool exp = MCPC_TIMING_BUILD != 0 || std::strstr(MCPC_BUILD_FLAGS, "MCPC_EXP") != nullptr ||
                            std::strstr(MCPC_BUILD_FLAGS, "MCPC_HEB_EXP") != nullptr;
    static char buf[1024];
    static const int n = std::snprintf(buf, sizeof buf, "libmcpc abi=%d arch=gfx950 csrc=%s commit=%s exp=%d stamps=%d flags=[%s]",
                                       MCPC_ABI_VERSION, MCPC_BUILD_CSRC_SHA, MCPC_BUILD_COMMIT, exp ? 1 : 0, MCPC_STAMPS_BUILD, MCPC_BUILD_FLAGS);
    (void)n;
    return buf;
}
const char* mcpc_last_error(void) { return g_err.c_str(); }

int mcpc_create(const mcpc_net_desc* d, mcpc_engine** out) {
    if (!d || !out) return fail(MCPC_EINVAL, "null argument");
    *out = nullptr;
    if (d->abi_version != MCPC_ABI_VERSION) return fail(MCPC_EINVAL, "ABI version mismatch: header %d, library %d", d->abi_version, MCPC_ABI_VERSION);
    if (d->n_latent < 1 || d->n_latent > kMaxLatent) return fail(MCPC_EINVAL, "n_latent=%d out of range 1..%d", d->n_latent, kMaxLatent);
    if (d->batch < 1 || d->n_in < 1 || d->n_out < 0) return fail(MCPC_EINVAL, "bad batch/n_in/n_out (%d/%d/%d)", d->batch, d->n_in, d->n_out);
    for (int l = 0; l < d->n_latent; ++l) {
        if (d->sizes[l] < 1) return fail(MCPC_EINVAL, "sizes[%d]=%d", l, d->sizes[l]);
        if (d->acts[l] < 0 || d->acts[l] > 2) return fail(MCPC_EINVAL, "acts[%d]=%d", l, d->acts[l]);
        if (!(d->ecoef[l] > 0.f)) return fail(MCPC_EINVAL, "ecoef[%d] must be positive", l);
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) return fail(MCPC_EINVAL, "device %d not present (%d devices)", d->device, ndev);
    HIP_TRY(hipSetDevice(d->device));

    mcpc_engine* e = new mcpc_engine();
    e->d = *d;
    e->L = d->n_latent;
    e->has_head = d->n_out > 0;
    e->Bpad = (d->batch + kCT - 1) / kCT * kCT;
    { const int rc = parse_tuning(d->tuning, e->knobs); if (rc) { delete e; return rc; } }
    e->d.tuning = nullptr;                   // the caller's string is not kept
    const Knobs& kn = e->knobs;
    // Default schedule: the in-place wave-specialised kernel (4 GEMM + 4 epilogue waves, 16 chains, one workgroup per CU; shards of
    // more units than CUs on the round schedule, setup_rounds).  The barrier kernel (16 chains, 4 waves, two workgroups per CU, generic
    // epilogues) is the fallback when the in-place LDS plan does not fit, and the independent form the parity checks replay the default
    // against (bench.py self_check, tests): tuning ws=0 forces it, ws=2 insists on the in-place kernel.
    int n_cu = 256;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess || n_cu <= 0) n_cu = 256;
    e->ws = kn.ws == 0 ? 0 : 2;
    e->nw = e->ws == 2 ? 2 * kWs2Pairs : kWaves;
    e->nwg = e->Bpad / e->ct;
    for (int l = 0; l < e->L; ++l) e->npad[l] = pad16(d->sizes[l]);
    e->out_pad = pad16(d->n_out);
    // The back-projection of the read-out error is accumulated in registers over the whole read-out: 16 tiles per workgroup
    // (a last latent layer of up to 256 units) in both kernels (cap_ws2 = 4 GEMM waves x 4 tiles, cap_bar = 4 waves x 4 tiles).
    const int last_tiles = e->has_head ? e->npad[e->L - 1] / 16 : 0;
    const int cap_ws2 = kWs2Pairs * ws2_nt<1>(), cap_bar = kNT * kWaves;
    if (last_tiles > std::max(cap_ws2, cap_bar) || (last_tiles > cap_bar && e->ws != 2)) {
        delete e;
        return fail(MCPC_ENOMEM, "last latent layer wider than %d units is not supported by the fused read-out (its back-projection is held in register tiles)", (e->ws == 2 ? std::max(cap_ws2, cap_bar) : cap_bar) * 16);
    }
    int rc = e->ws == 2 ? plan_lds_ws2(e) : plan_lds(e);
    if (rc && e->ws == 2 && kn.ws == -1 && last_tiles <= cap_bar) {   // no in-place plan fits: the barrier schedule
        g_err.clear();
        e->ws = 0; e->nw = kWaves;
        rc = plan_lds(e);
    }
    if (rc) { delete e; return rc; }
    e->nwg_live = e->ws == 2 ? (d->batch + 15) / 16 : e->nwg;

    auto bail = [&](int code) { free_all(e); delete e; return code; };
    for (int l = 0; l < e->L; ++l) {
        const size_t n = (size_t)e->Bpad * e->npad[l];
        if ((rc = dmalloc(e->x[l], n)) || (rc = dmalloc(e->m[l], n)) || (rc = dmalloc(e->v[l], n))) return bail(rc);
        if (hipMemset(e->x[l], 0, n * 4) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    }
    if ((rc = dmalloc(e->e0sum, (size_t)e->Bpad * e->npad[0])) || (rc = dmalloc(e->mu1, (size_t)e->Bpad * e->npad[0]))) return bail(rc);
    if (hipMemset(e->e0sum, 0, (size_t)e->Bpad * e->npad[0] * 4) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    if (e->has_head) {
        if ((rc = dmalloc(e->ypad, (size_t)e->Bpad * e->out_pad)) || (rc = dmalloc(e->ytile, (size_t)e->Bpad * e->out_pad))) return bail(rc);
        if (hipMemset(e->ypad, 0, (size_t)e->Bpad * e->out_pad * 4) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
        e->ywords = (e->out_pad + 31) / 32;
        if ((rc = dmalloc(e->ybits, (size_t)e->Bpad * e->ywords)) || (rc = dmalloc(e->y_binary, 3))) return bail(rc);
        if (hipMemset(e->y_binary, 0, 3 * sizeof(int)) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));   // [1] stays 0; [2]: target in [-1, 2]
    }
    // Linear descriptors + gradient sums
    const int nlin = e->L + (e->has_head ? 1 : 0);
    for (int j = 0; j < nlin; ++j) {
        Lin& ln = e->lin[j];
        ln.n_in = j == 0 ? d->n_in : d->sizes[j - 1];
        ln.n_out = j < e->L ? d->sizes[j] : d->n_out;
        ln.in_pad = j == 0 ? d->n_in : e->npad[j - 1];
        ln.out_pad = pad16(ln.n_out);
        ln.g_ld = ln.in_pad;
        if ((rc = dmalloc(ln.G, (size_t)ln.out_pad * ln.g_ld)) || (rc = dmalloc(ln.Gb, (size_t)ln.out_pad))) return bail(rc);
        if (hipMemset(ln.G, 0, (size_t)ln.out_pad * ln.g_ld * 4) != hipSuccess || hipMemset(ln.Gb, 0, (size_t)ln.out_pad * 4) != hipSuccess)
            return bail(fail(MCPC_EHIP, "hipMemset failed"));
        if (j >= 1) {
            // packed fragments (floats): tiles x k-blocks x kFragBlock 16-byte units, forward and backward
            const size_t pkf = (size_t)(ln.out_pad / 16) * kblocks(ln.in_pad) * kFragBlock * 4;
            const size_t pkb = (size_t)(ln.in_pad / 16) * kblocks(ln.out_pad) * kFragBlock * 4;
            if ((rc = dmalloc(ln.Wf, pkf)) || (rc = dmalloc(ln.Wb, pkb)) || (rc = dmalloc(ln.bias_pad, (size_t)ln.out_pad))) return bail(rc);
        }
    }
    // spill ring
    size_t per_slot = 0;
    for (int l = 0; l < e->L; ++l) per_slot += (size_t)e->Bpad * e->npad[l] * (l >= 1 ? 2 : 1);
    per_slot += (size_t)e->Bpad * e->out_pad;
    per_slot *= sizeof(float);
    // defaults sized for 288 GB of HBM per GPU: room for `slot_cap` steps (384: Hebbian segments of 128 steps; 17 GB at cfg-M, capped
    // by the quarter of the device's memory for a shard of 48 000 chains), at least 6 GiB.  Round 2: a 2 GiB ring (segments of 24)
    // cost 1.4 % more per step at cfg-M; 6 GiB instead of 24 cost 13 % at 24 000 chains (segments of 17 steps).  Round 3 (step kernel
    // on every CU, the flush a phase of its own): 192 / 384 / 576 slots = 12 290 / 12 930 / 13 070 steps/s on the headline call.
    int64_t budget = d->spill_budget_bytes;
    if (budget <= 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
        budget = std::max<int64_t>((int64_t)6 << 30, std::min<int64_t>((int64_t)kn.slot_cap * (int64_t)per_slot, (int64_t)(total_b / 4)));
    }
    if (kn.spill_gb > 0) budget = (int64_t)kn.spill_gb << 30;
    e->slots = (int)std::max<int64_t>(1, std::min<int64_t>(kn.slot_cap, budget / (int64_t)per_slot));
    // `half_slots` = slots of one PART of the ring = steps of one Hebbian segment
    if (kn.no_overlap || e->slots < 2) e->half_slots = e->slots;          // serial flushes on the caller's stream: one part
    else if (e->slots >= kn.ring_parts) { e->half_slots = e->slots / kn.ring_parts; e->slots = e->half_slots * kn.ring_parts; }
    else { e->slots &= ~1; e->half_slots = e->slots / 2; }                   // fewer slots than parts: two halves
    // (the ring itself is allocated by the first run that accumulates Hebbian sums: ensure_spill)
    // layout of the spilled operands per Linear: tile-major for the fp16 tiled kernel, row-major for the others (plan_hebbian's
    // choice of kernel depends on the shapes only)
    for (int j = 1; j < nlin; ++j) {
        const int et = e->lin[j].out_pad / 16, at = e->lin[j].in_pad / 16;
        e->lin[j].spill_tm = (heb_wide(et, at) || heb_narrow_in(et, at)) && !kn.heb_fp32;
    }

    if ((rc = e->ws == 2 ? build_phases_ws2(e) : build_phases(e))) return bail(rc);
    // the unified-wave kernel beside the in-place kernel, where its plan fits (mcpc_steps_u.h)
    if (e->ws == 2 && (kn.ws == -1 || kn.ws == 3) && !kn.no_lean && !kn.no_xl) {
        (void)plan_lds_u(e);
        e->u.on = e->u.ok;
        e->u.prefer = e->u.ok && (kn.ws == 3 || choose_unified(e));
        if (e->u.on && (rc = build_phases_u(e))) return bail(rc);
    }
    if (kn.ws == 3 && !e->u.on) return bail(fail(MCPC_ENOMEM, "tuning ws=3: the unified-wave kernel's LDS plan does not fit this network (%d bytes)", e->u.lds_bytes));
    if ((rc = dmalloc(e->err, 1))) return bail(rc);
    if (hipMemset(e->err, 0, sizeof(int)) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    if ((rc = dmalloc(e->spillmax, (size_t)kMaxRingParts * kSpillTensors))) return bail(rc);
    if (hipMemset(e->spillmax, 0, (size_t)kMaxRingParts * kSpillTensors * sizeof(unsigned)) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    if ((rc = dmalloc(e->wexp, kMaxLatent + 1))) return bail(rc);
    if (hipMemset(e->wexp, 0, (kMaxLatent + 1) * sizeof(int)) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    if ((rc = dmalloc(e->clk, 2))) return bail(rc);
    if (hipMemset(e->clk, 0, 2 * sizeof(unsigned long long)) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    if ((rc = dmalloc(e->dummy, 1024))) return bail(rc);
    if (hipMemset(e->dummy, 0, 4096) != hipSuccess) return bail(fail(MCPC_EHIP, "hipMemset failed"));
    const void* kfn = e->ws == 2 ? (const void*)mcpc_steps_ws2_kernel<1> : (const void*)mcpc_steps_kernel<1, 4>;
    hipError_t herr = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes);
    if (herr != hipSuccess) return bail(fail(MCPC_EHIP, "hipFuncSetAttribute(%d bytes LDS) failed: %s", e->lds_bytes, hipGetErrorString(herr)));
    if (e->ws == 2 && e->nwg_live > n_cu && kn.rr) {
        if ((rc = setup_rounds(e, n_cu))) return bail(rc);
    }
    if (e->u.on) {
        if (hipFuncSetAttribute((const void*)mcpc_steps_u_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, e->u.lds_bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)mcpc_steps_u_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, e->u.lds_bytes) != hipSuccess)
            return bail(fail(MCPC_EHIP, "hipFuncSetAttribute failed for the unified-wave kernel (%d bytes LDS)", e->u.lds_bytes));
    }
    *out = e;
    return MCPC_OK;
}

int mcpc_destroy(mcpc_engine* e) {
    if (!e) return MCPC_OK;
    (void)hipSetDevice(e->d.device);
    (void)hipDeviceSynchronize();
    free_all(e);
    delete e;
    return MCPC_OK;
}

int mcpc_bind_params(mcpc_engine* e, int j, const float* W, const float* bias) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    const int nlin = e->L + (e->has_head ? 1 : 0);
    if (j < 0 || j >= nlin) return fail(MCPC_EINVAL, "Linear index %d out of range 0..%d", j, nlin - 1);
    if (!W) return fail(MCPC_EINVAL, "null weight pointer for Linear %d", j);
    e->lin[j].W = W;
    e->lin[j].bias = bias;
    e->lin[j].bound = true;
    return MCPC_OK;
}

int mcpc_params_changed(mcpc_engine* e, void* stream_) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(e->d.device));
    const int nlin = e->L + (e->has_head ? 1 : 0);
    for (int j = 0; j < nlin; ++j)
        if (!e->lin[j].bound) return fail(MCPC_ESTATE, "Linear %d has no bound parameters", j);
    for (int j = 1; j < nlin; ++j) {
        Lin& ln = e->lin[j];
        const size_t total = (size_t)ln.out_pad * ln.in_pad;
        const int grid = std::max(grid_for(total), (ln.out_pad + 255) / 256);
        // the exponent the fp16 planes of this Linear are scaled by (from max |W|), then the planes themselves
        hipLaunchKernelGGL(mcpc_wexp_kernel, dim3(1), dim3(1024), 0, stream, ln.W, (size_t)ln.n_out * ln.n_in, e->wexp + j);
        hipLaunchKernelGGL(mcpc_pack_kernel, dim3(grid), dim3(256), 0, stream, ln.W, ln.bias, ln.Wf, ln.Wb, ln.bias_pad,
                           ln.n_out, ln.n_in, ln.out_pad / 16, ln.in_pad / 16, (const int*)(e->wexp + j));
    }
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

int mcpc_bind_inputs(mcpc_engine* e, const float* inputs, void* /*stream*/) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    e->inputs = inputs;
    return MCPC_OK;
}

int mcpc_bind_target(mcpc_engine* e, const float* target, void* stream_) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    if (!e->has_head) return fail(MCPC_EINVAL, "network has no read-out: no target to bind");
    if (!target) return fail(MCPC_EINVAL, "null target");
    HIP_TRY(hipSetDevice(e->d.device));
    const size_t total = (size_t)e->Bpad * e->out_pad;
    hipLaunchKernelGGL(mcpc_pad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream_, target, e->ypad,
                       e->d.batch, e->d.n_out, e->Bpad, e->out_pad);
    hipLaunchKernelGGL(mcpc_tile_major_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream_, e->ypad, e->ytile, e->Bpad, e->out_pad);
    // a 0/1 target (the Bernoulli read-out's usual one) is also kept bit-packed; the flag tells the step kernel which to read
    HIP_TRY(hipMemsetAsync(e->y_binary, 0xff, sizeof(int), (hipStream_t)stream_));
    HIP_TRY(hipMemsetAsync(e->y_binary + 2, 0xff, sizeof(int), (hipStream_t)stream_));      // cleared by the first value outside [-1, 2] (headb_fixed_exp)
    hipLaunchKernelGGL(mcpc_pack_target_bits_kernel, dim3(grid_for((size_t)e->Bpad * e->ywords)), dim3(256), 0, (hipStream_t)stream_,
                       e->ypad, e->ybits, e->y_binary, e->Bpad, e->out_pad, e->ywords);
    HIP_TRY(hipGetLastError());
    e->target_bound = true;
    return MCPC_OK;
}

int mcpc_load_state(mcpc_engine* e, const float* const* x, void* stream_) {
    if (!e || !x) return fail(MCPC_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    for (int l = 0; l < e->L; ++l) {
        if (!x[l]) return fail(MCPC_EINVAL, "null state pointer for layer %d", l);
        const size_t total = (size_t)e->Bpad * e->npad[l];
        hipLaunchKernelGGL(mcpc_pad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream_, x[l], e->x[l],
                           e->d.batch, e->d.sizes[l], e->Bpad, e->npad[l]);
    }
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

int mcpc_store_state(mcpc_engine* e, float* const* x, void* stream_) {
    if (!e || !x) return fail(MCPC_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    for (int l = 0; l < e->L; ++l) {
        if (!x[l]) return fail(MCPC_EINVAL, "null state pointer for layer %d", l);
        const size_t total = (size_t)e->d.batch * e->d.sizes[l];
        hipLaunchKernelGGL(mcpc_unpad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream_, e->x[l], x[l],
                           e->d.batch, e->d.sizes[l], e->npad[l]);
    }
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

int mcpc_store_adam_state(mcpc_engine* e, float* const* m, float* const* v, void* stream_) {
    if (!e || !m || !v) return fail(MCPC_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    for (int l = 0; l < e->L; ++l) {
        if (!m[l] || !v[l]) return fail(MCPC_EINVAL, "null Adam state pointer for layer %d", l);
        const size_t total = (size_t)e->d.batch * e->d.sizes[l];
        hipLaunchKernelGGL(mcpc_unpad_adam_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream_, e->m[l], m[l],
                           e->d.batch, e->d.sizes[l], e->npad[l]);
        hipLaunchKernelGGL(mcpc_unpad_adam_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream_, e->v[l], v[l],
                           e->d.batch, e->d.sizes[l], e->npad[l]);
    }
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

}  // extern "C"

namespace {

// How the Hebbian sums of Linear j are computed for `rows` spilled rows: the LDS-tiled kernel (mcpc_hebbian.h) for wide
// Linears -- as one or two launches whose error-tile groups cover the output exactly (49 tiles = 17 + 16 + 16) -- the same
// kernel with the operands swapped for a wide Linear with a narrow input (256 x 32: the narrow side takes the TE slot),
// and the register-streaming kernel for whatever is left (few output tiles: HBM-bound whatever the tiling).
struct HebPlan {
    bool tiled = false, swapped = false;
    int ra = 0;                                    // activation tiles per wave (TA = 8 ra)
    int te[2] = {0, 0}, n_mt[2] = {0, 0};          // up to two launches: error tiles per group, number of groups
    int n_nt = 1;
    int wave_tiles = 0;                            // streaming kernel: 64 x 64 wave tiles
    int ksplit = 1, rps = 0;
    int ksplit_cap = 1;                            // upper bound of ksplit that never decreases with `rows`: sizes the slabs
};

HebPlan plan_hebbian(const mcpc_engine* e, int ne, int na, int rows) {
    HebPlan h;
    const int et = ne / 16, at = na / 16;
    const bool wide = heb_wide(et, at);
    const bool narrow_in = heb_narrow_in(et, at);       // e.g. 256 x 32
    h.tiled = wide || narrow_in;
    h.swapped = narrow_in;
    if (wide) {
        h.ra = at >= 16 ? 2 : 1;
        h.n_nt = at / (8 * h.ra);
        if (et <= 8) { h.te[0] = 8; h.n_mt[0] = 1; }
        else if (et <= 16) { h.te[0] = 16; h.n_mt[0] = 1; }
        else {
            // et = 17 b + 16 a exactly when b = et mod 16 groups of 17 fit; otherwise groups of 17 with a ragged last one
            const int b17 = et % 16, a16 = (et - 17 * b17) / 16;
            if (et - 17 * b17 >= 0) { h.te[0] = 17; h.n_mt[0] = b17; h.te[1] = 16; h.n_mt[1] = a16; }
            else { h.te[0] = 17; h.n_mt[0] = (et + 16) / 17; }
            if (h.n_mt[0] == 0) { h.te[0] = h.te[1]; h.n_mt[0] = h.n_mt[1]; h.te[1] = 0; h.n_mt[1] = 0; }
        }
    } else if (narrow_in) {
        h.ra = 2; h.n_nt = 1; h.te[0] = at; h.n_mt[0] = 1;          // E slot = activations (at tiles), A slot = errors (16 tiles)
    }
    if (h.tiled) {
        // ~48 stages of 32 rows per workgroup (0.3 ms at cfg-M): short enough that the step kernel's next segment never
        // waits long for CUs, long enough that the slab traffic stays at a few percent of the spill's
        int want = e->knobs.dw_ksplit > 0 ? e->knobs.dw_ksplit : std::max(1, rows / (48 * kHebKB));
        // a small flush (the reference's batch of 256: 25 600 rows, 16 splits) would run 16-48 workgroups of 48 stages on an idle chip,
        // 0.10-0.17 ms per Linear: with at least 8 stages per workgroup, split until the launch has about a workgroup per CU
        if (e->knobs.dw_ksplit <= 0) {
            const int cols = std::max(1, (h.n_mt[0] + h.n_mt[1]) * h.n_nt);
            want = std::max(want, std::min(rows / (8 * kHebKB), (256 + cols - 1) / cols));
        }
        want = std::min(want, std::max(1, rows / kHebKB));
        h.ksplit_cap = want;
        h.rps = ((rows + want - 1) / want + kHebKB - 1) / kHebKB * kHebKB;
        h.ksplit = (rows + h.rps - 1) / h.rps;
    } else {
        h.wave_tiles = ((ne + 63) / 64) * ((na + 63) / 64);
        int ksplit = std::max(1, std::min(4096 / h.wave_tiles, rows / 64));
        h.ksplit_cap = ksplit;
        h.rps = ((rows + ksplit - 1) / ksplit + 15) / 16 * 16;
        h.ksplit = (rows + h.rps - 1) / h.rps;
    }
    return h;
}

template <int TE, int RA, bool SW = false>
int launch_heb(const HebArgs& a, hipStream_t stream) {
    constexpr int lds_bytes = 2 * kHebKB * (heb_lds_stride(16 * TE) + heb_lds_stride(16 * 8 * RA)) * (int)sizeof(float);
    static bool attr_set[16] = {false};      // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)mcpc_heb_kernel<TE, RA, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(MCPC_EHIP, "hipFuncSetAttribute failed for the Hebbian kernel");
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((mcpc_heb_kernel<TE, RA, SW>), dim3(a.n_mt * a.n_nt * a.ksplit), dim3(kHebThreads), lds_bytes, stream, a);
    return 0;
}

template <int TE, int RA, bool SW = false>
int launch_heb7(const HebArgs& a, hipStream_t stream) {
    // two plane buffers + the bias sums + every wave's transpose scratch for one activation tile (2 row tiles x 4 x 17 float4)
    constexpr int lds_bytes = 2 * 2 * 16 * TE * kHeb7LD * 2 + 16 * TE * (int)sizeof(float) + 8 * 2 * 272 * (int)sizeof(float);
    static bool attr_set[16] = {false};      // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)mcpc_heb7_kernel<TE, RA, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(MCPC_EHIP, "hipFuncSetAttribute failed for the Hebbian kernel");
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((mcpc_heb7_kernel<TE, RA, SW>), dim3(a.n_mt * a.n_nt * a.ksplit), dim3(kHebThreads), lds_bytes, stream, a);
    return 0;
}

// Allocated by the first run that accumulates Hebbian sums (inference-only engines never pay for it): the spill ring, the
// slabs of the split-K partial sums (sized for a flush of half the ring), the low-priority stream and events of the
// overlapped flush.
int ensure_spill(mcpc_engine* e, hipStream_t stream) {
    if (e->spill_ready) return 0;
    int rc;
    if (e->half_slots < e->slots && !e->aux) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo = least urgent
        // Streams become hardware queues in the order they are created, and the queues are dealt out over the pipes of the
        // command processor: which pipe the two flush queues share with whom is worth 0.5 % of the learning call (94.3
        // against 94.7 us per step) and can be neither asked nor requested.  One stream created in front of them -- never
        // used -- puts them where they measured best in a process whose only other compute queue is the caller's (any one
        // extra stream, of any priority, did; two did worse by 2 %: 96.6).
        if (!e->spacer) (void)hipStreamCreateWithFlags(&e->spacer, hipStreamNonBlocking);
        if (hipStreamCreateWithPriority(&e->aux, hipStreamNonBlocking, lo) != hipSuccess) return fail(MCPC_EHIP, "hipStreamCreateWithPriority failed");
        if (e->knobs.flush_streams >= 2 &&
            (hipStreamCreateWithPriority(&e->aux3, hipStreamNonBlocking, lo) != hipSuccess ||
             hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess))
            return fail(MCPC_EHIP, "second flush stream could not be created");
        for (int h = 0; h < kMaxRingParts; ++h)
            if (hipEventCreateWithFlags(&e->ev_steps[h], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_flush[h], hipEventDisableTiming) != hipSuccess)
                return fail(MCPC_EHIP, "hipEventCreate failed");
    }
    for (int l = 0; l < e->L; ++l) {
        const size_t n = (size_t)e->slots * e->Bpad * e->npad[l];
        if (!e->spill_a[l] && (rc = dmalloc(e->spill_a[l], n))) return rc;
        if (l >= 1 && !e->spill_e[l] && (rc = dmalloc(e->spill_e[l], n))) return rc;
    }
    if (e->has_head && !e->spill_eo && (rc = dmalloc(e->spill_eo, (size_t)e->slots * e->Bpad * e->out_pad))) return rc;
    // rows of a 16-chain unit that is all padding (never launched) read as zero in every slot: the Hebbian GEMMs sum over all Bpad rows.
    // On the RUN's stream: the step kernels and (behind ev_steps) the flushes of this and every later run are ordered after it, also when
    // the caller's stream is a non-blocking one that the null stream does not synchronise with.
    if (e->nwg_live * e->ct < e->Bpad && e->ws == 2) {
        const size_t live = (size_t)e->nwg_live * 16, dead = (size_t)e->Bpad - live;
        auto zero_tail = [&](float* base, int npad) {
            return hipMemset2DAsync(base + live * npad, (size_t)e->Bpad * npad * sizeof(float), 0, dead * npad * sizeof(float), (size_t)e->slots, stream) == hipSuccess;
        };
        bool ok = true;
        for (int l = 0; l < e->L; ++l) { ok = ok && zero_tail(e->spill_a[l], e->npad[l]); if (l >= 1) ok = ok && zero_tail(e->spill_e[l], e->npad[l]); }
        if (e->has_head) ok = ok && zero_tail(e->spill_eo, e->out_pad);
        if (!ok) return fail(MCPC_EHIP, "hipMemset2D of the spill ring's padding rows failed");
    }
    const int nlin = e->L + (e->has_head ? 1 : 0);
    const int max_rows = e->half_slots * e->Bpad;
    size_t total = 0;
    for (int j = 1; j < nlin; ++j) {
        Lin& ln = e->lin[j];
        const HebPlan h = plan_hebbian(e, ln.out_pad, ln.in_pad, max_rows);
        ln.slab_off = total;
        ln.slab_floats = (size_t)h.ksplit_cap * ((size_t)ln.out_pad * ln.in_pad + ln.out_pad);
        total += ln.slab_floats;
    }
    if (!e->slab && (rc = dmalloc(e->slab, total))) return rc;
    e->slab_floats = total;
    e->spill_ready = true;
    return 0;
}

// One Hebbian flush: fold `n_slots` spilled steps into the gradient sums of every Linear j >= 1.  Every Linear has its own
// slab region, so the GEMMs of a flush are independent launches and ONE reduction launch follows them.
static inline int spill_id_a_host(int l) { return l; }                  // (mcpc_kernels.h: spill_id_a)
int flush_spill(mcpc_engine* e, int n_slots, int slot0, hipStream_t stream, int part) {
    const unsigned* const smax = e->spillmax + (size_t)part * kSpillTensors;      // largest |value| per spilled tensor of this segment
    const int rows = n_slots * e->Bpad;
    const int nlin = e->L + (e->has_head ? 1 : 0);
    ReduceJobs jobs{};
    unsigned max_blocks = 1;
    // Overlapped flush: the launches alternate between two low-priority streams (kernels of one stream run one after the
    // other, each leaving the idle CUs half empty while its last workgroups finish); the reduction joins them.
    hipStream_t const main_stream = stream;
    const bool two = e->aux3 != nullptr && stream == e->aux;
    double load[2] = {0.0, 0.0};                            // work queued on each stream so far (output columns x rows)
    if (two) { HIP_TRY(hipEventRecord(e->ev_fork, main_stream)); HIP_TRY(hipStreamWaitEvent(e->aux3, e->ev_fork, 0)); }
    auto next_stream = [&](double cost) {                   // greedy: the launch goes to the stream with less work queued
        const int k = (two && load[1] < load[0]) ? 1 : 0;
        load[k] += cost;
        return k ? e->aux3 : main_stream;
    };
    for (int j = 1; j < nlin; ++j) {
        Lin& ln = e->lin[j];
        const int ne = ln.out_pad, na = ln.in_pad;
        const float* E = (j < e->L ? e->spill_e[j] : e->spill_eo) + (size_t)slot0 * e->Bpad * ne;
        const float* A = e->spill_a[j - 1] + (size_t)slot0 * e->Bpad * na;
        const HebPlan h = plan_hebbian(e, ne, na, rows);
        const int e_id = j < e->L ? kMaxLatent + j : kSpillIdEo;          // (mcpc_kernels.h: spill_id_e / kSpillIdEo)
        float* slab = e->slab + ln.slab_off;
        float* slab_b = slab + (size_t)h.ksplit * ne * na;
        if ((size_t)h.ksplit * ((size_t)ne * na + ne) > ln.slab_floats)
            return fail(MCPC_ESTATE, "internal: Hebbian slabs of Linear %d (%d splits) exceed their allocation", j, h.ksplit);
        if (h.swapped) {
            stream = next_stream((double)ne * na);
            // transposed product: the activations take the E slot, the errors the A slot; slab = [split][na][ne]
            HebArgs a{A, E, slab, slab_b, rows, na, ne, h.rps, 1, 1, h.ksplit, 0, smax + spill_id_a_host(j - 1), smax + e_id};
            int rc = 0;
            if (!e->knobs.heb_fp32) {
                if (h.te[0] == 1) rc = launch_heb7<1, 2, true>(a, stream);
                else if (h.te[0] == 2) rc = launch_heb7<2, 2, true>(a, stream);
                else rc = launch_heb7<4, 2, true>(a, stream);
            } else
            if (h.te[0] == 1) rc = launch_heb<1, 2, true>(a, stream);
            else if (h.te[0] == 2) rc = launch_heb<2, 2, true>(a, stream);
            else rc = launch_heb<4, 2, true>(a, stream);
            if (rc) return rc;
        } else if (h.tiled) {
            int col = 0;
            for (int part = 0; part < 2; ++part) {
                if (h.n_mt[part] == 0) continue;
                HebArgs a{E, A, slab, slab_b, rows, ne, na, h.rps, h.n_mt[part], h.n_nt, h.ksplit, col, smax + e_id, smax + spill_id_a_host(j - 1)};
                stream = next_stream((double)h.n_mt[part] * h.te[part] * 16 * na);
                int rc = 0;
                const int te = h.te[part];
                if (!e->knobs.heb_fp32) {
                    if (te == 17 && h.ra == 2 && !e->knobs.heb171) rc = launch_heb7<17, 2>(a, stream);
                    else if (te == 17 && h.ra == 2) {
                        // 136 accumulators + the rest do not fit 256 registers (19 spilled): 8 activation tiles per workgroup instead,
                        // i.e. twice the activation groups -- this group's 272 error columns are read twice (+6.5 MB per step at cfg-M)
                        HebArgs a1 = a;
                        a1.n_nt = 2 * a.n_nt;
                        rc = launch_heb7<17, 1>(a1, stream);
                    }
                    else if (te == 16 && h.ra == 2) rc = launch_heb7<16, 2>(a, stream);
                    else if (te == 8 && h.ra == 2) rc = launch_heb7<8, 2>(a, stream);
                    else if (te == 17) rc = launch_heb7<17, 1>(a, stream);
                    else if (te == 16) rc = launch_heb7<16, 1>(a, stream);
                    else rc = launch_heb7<8, 1>(a, stream);
                } else
                if (te == 17 && h.ra == 2) rc = launch_heb<17, 2>(a, stream);
                else if (te == 16 && h.ra == 2) rc = launch_heb<16, 2>(a, stream);
                else if (te == 8 && h.ra == 2) rc = launch_heb<8, 2>(a, stream);
                else if (te == 17) rc = launch_heb<17, 1>(a, stream);
                else if (te == 16) rc = launch_heb<16, 1>(a, stream);
                else rc = launch_heb<8, 1>(a, stream);
                if (rc) return rc;
                col += h.n_mt[part] * te * 16;
            }
        } else {
            stream = next_stream((double)ne * na);
            hipLaunchKernelGGL(mcpc_dw_kernel, dim3((h.wave_tiles + 3) / 4, h.ksplit), dim3(256), 0, stream, E, A, slab, slab_b,
                               rows, ne, na, h.rps);
        }
        const float sign = j < e->L ? -1.0f : 1.0f;
        jobs.job[jobs.n_jobs++] = ReduceJob{slab, ln.G, ne * na, h.ksplit, sign, h.swapped ? ne : 0, h.swapped ? na : 0};
        jobs.job[jobs.n_jobs++] = ReduceJob{slab_b, ln.Gb, ne, h.ksplit, sign, 0, 0};
        max_blocks = std::max(max_blocks, (unsigned)std::min<size_t>(((size_t)ne * na + 255) / 256, 2048));
    }
    stream = main_stream;
    if (two) { HIP_TRY(hipEventRecord(e->ev_join, e->aux3)); HIP_TRY(hipStreamWaitEvent(main_stream, e->ev_join, 0)); }
    if (jobs.n_jobs > 0)
        hipLaunchKernelGGL(mcpc_reduce_jobs_kernel, dim3(max_blocks, jobs.n_jobs), dim3(256), 0, stream, jobs);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int mcpc_run(mcpc_engine* e, const mcpc_run_desc* r, void* stream_) {
    if (!e || !r) return fail(MCPC_EINVAL, "null argument");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(e->d.device));
    const int nlin = e->L + (e->has_head ? 1 : 0);
    for (int j = 0; j < nlin; ++j)
        if (!e->lin[j].bound) return fail(MCPC_ESTATE, "Linear %d has no bound parameters (mcpc_bind_params + mcpc_params_changed first)", j);
    if (r->T < 1 || r->t_begin < 0 || r->n_steps < 1 || r->t_begin + r->n_steps > r->T)
        return fail(MCPC_EINVAL, "bad step range: T=%d t_begin=%d n_steps=%d", r->T, r->t_begin, r->n_steps);
    if (r->loss_kind < 0 || r->loss_kind > 2) return fail(MCPC_EINVAL, "loss_kind=%d", r->loss_kind);
    if (r->loss_kind != MCPC_LOSS_NONE) {
        if (!e->has_head) return fail(MCPC_EINVAL, "a loss needs a read-out Linear (n_out > 0)");
        if (!e->target_bound) return fail(MCPC_ESTATE, "loss requested but no target bound");
        if (r->loss_kind == MCPC_LOSS_GAUSSIAN && !(r->loss_var > 0.0)) return fail(MCPC_EINVAL, "loss_var must be positive");
        if (r->mask_start < 0 || r->mask_start >= e->d.n_out) return fail(MCPC_EINVAL, "mask_start=%d outside 0..%d", r->mask_start, e->d.n_out - 1);
    }
    if (r->xopt_kind != MCPC_XOPT_SGD && r->xopt_kind != MCPC_XOPT_ADAM) return fail(MCPC_EINVAL, "xopt_kind=%d", r->xopt_kind);
    if (!(r->lr > 0.0)) return fail(MCPC_EINVAL, "lr must be positive");
    if (r->noise_mode < 0 || r->noise_mode > 2) return fail(MCPC_EINVAL, "noise_mode=%d", r->noise_mode);
    if (r->noise_mode != MCPC_NOISE_NONE && r->xopt_kind != MCPC_XOPT_SGD)
        return fail(MCPC_EINVAL, "the fused Langevin kick is defined for SGD on x only (reference utils/model.py:35-44 steps the same optimizer)");
    if (r->noise_mode != MCPC_NOISE_NONE && !(r->noise_var >= 0.0)) return fail(MCPC_EINVAL, "noise_var must be >= 0");
    if (r->noise_mode == MCPC_NOISE_EXTERNAL)
        for (int l = 0; l < e->L; ++l)
            if (!r->ext_noise[l]) return fail(MCPC_EINVAL, "ext_noise[%d] is null", l);
    if (!r->update_x)
        for (int l = 0; l < e->L; ++l)
            if (!r->xgrad[l]) return fail(MCPC_EINVAL, "update_x=0 needs xgrad[%d]", l);
    if (r->energy_mode < 0 || r->energy_mode > 2) return fail(MCPC_EINVAL, "energy_mode=%d", r->energy_mode);
    if (r->energy_mode != MCPC_ENERGY_NONE && !r->energies_out) return fail(MCPC_EINVAL, "energies_out is null");
    if (r->rec_count < 0 || (r->rec_count > 0 && r->rec_stride < 1)) return fail(MCPC_EINVAL, "bad record schedule");
    const int acc_b = std::max(r->acc_begin, 0), acc_e = std::min(r->acc_end, r->T);
    if (!e->retired.empty()) free_completed_retired(e);

    // ---- per-run device tables ----------------------------------------------------------------
    if (r->xopt_kind == MCPC_XOPT_ADAM) {
        const size_t need = (size_t)r->n_steps * 2;
        if (need > e->adam_cap) {
            // earlier launches may still read the old table: retire it instead of synchronising
            if (e->adam_coef) retire(e, e->adam_coef, stream);
            e->adam_coef = nullptr;
            size_t cap = 1024;
            while (cap < need) cap *= 2;
            int rc = dmalloc(e->adam_coef, cap);
            if (rc) return rc;
            e->adam_cap = cap;
        }
        // pinned staging table, double-buffered; the event wait below is a no-op unless the upload issued two Adam runs ago
        // has still not executed
        const int hb = e->adam_next; e->adam_next ^= 1;
        if (!e->adam_ev[hb]) HIP_TRY(hipEventCreateWithFlags(&e->adam_ev[hb], hipEventDisableTiming));
        else HIP_TRY(hipEventSynchronize(e->adam_ev[hb]));
        if (need > e->adam_host_cap[hb]) {
            if (e->adam_host[hb]) HIP_TRY(hipHostFree(e->adam_host[hb]));
            e->adam_host[hb] = nullptr;
            void* q = nullptr;
            HIP_TRY(hipHostMalloc(&q, e->adam_cap * sizeof(float), hipHostMallocDefault));
            e->adam_host[hb] = (float*)q; e->adam_host_cap[hb] = e->adam_cap;
        }
        float* tab = e->adam_host[hb];
        for (int s = 0; s < r->n_steps; ++s) {
            const double step = (double)(r->adam_step0 + s + 1);
            const double bc1 = 1.0 - std::pow(r->beta1, step);
            const double bc2 = 1.0 - std::pow(r->beta2, step);
            tab[2 * s] = (float)(-(r->lr / bc1));         // addcdiv_'s value = -step_size, a python double rounded to fp32 once
            tab[2 * s + 1] = (float)std::sqrt(bc2);              // bias_correction2_sqrt, likewise
        }
        HIP_TRY(hipMemcpyAsync(e->adam_coef, tab, need * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(e->adam_ev[hb], stream));
        if (r->adam_step0 == 0)
            for (int l = 0; l < e->L; ++l) {
                HIP_TRY(hipMemsetAsync(e->m[l], 0, (size_t)e->Bpad * e->npad[l] * 4, stream));
                HIP_TRY(hipMemsetAsync(e->v[l], 0, (size_t)e->Bpad * e->npad[l] * 4, stream));
            }
    }
    const size_t erows = r->energy_mode == MCPC_ENERGY_ALL ? (size_t)r->T : 1;
    // energy partials per step: one slot per workgroup; the in-place kernel indexes them by 16-chain tile (its 32- and 16-chain
    // forms can then serve the same call)
    const size_t eslots = e->ws == 2 ? (size_t)e->Bpad / 16 : (size_t)e->nwg;
    if (r->energy_mode != MCPC_ENERGY_NONE && erows > e->epart_rows) {
        if (e->epart) retire(e, e->epart, stream);        // earlier launches may still write it
        e->epart = nullptr;
        // geometric growth: a caller whose T creeps up call by call re-allocates O(log T) times, not every call
        const size_t cap = std::max(erows, e->epart_rows + e->epart_rows / 2);
        int rc = dmalloc(e->epart, cap * eslots * (kMaxLatent + 1));
        if (rc) return rc;
        // (the slots of a 16-chain unit that is all padding are never written: they must read as zero)
        HIP_TRY(hipMemsetAsync(e->epart, 0, cap * eslots * (kMaxLatent + 1) * sizeof(double), stream));
        e->epart_rows = cap;
    }
    // mu_1 = inputs W0^T + b0 (constant during the run: weights only change between runs)
    {
        const Lin& l0 = e->lin[0];
        const size_t total = (size_t)e->Bpad * e->npad[0];
        hipLaunchKernelGGL(mcpc_mu1_kernel, dim3(grid_for(total)), dim3(256), 0, stream, e->inputs, l0.W, l0.bias, e->mu1,
                           e->d.batch, e->d.n_in, e->d.sizes[0], e->Bpad, e->npad[0]);
    }
    const bool run_accumulates = acc_b < acc_e && r->t_begin < acc_e && r->t_begin + r->n_steps > acc_b;
    if (run_accumulates) { const int rc = ensure_spill(e, stream); if (rc) return rc; }
    if (r->acc_reset && run_accumulates) {
        for (int j = 0; j < nlin; ++j) {
            HIP_TRY(hipMemsetAsync(e->lin[j].G, 0, (size_t)e->lin[j].out_pad * e->lin[j].g_ld * 4, stream));
            HIP_TRY(hipMemsetAsync(e->lin[j].Gb, 0, (size_t)e->lin[j].out_pad * 4, stream));
        }
        HIP_TRY(hipMemsetAsync(e->e0sum, 0, (size_t)e->Bpad * e->npad[0] * 4, stream));
    }

    // ---- kernel parameters ----------------------------------------------------------------------
    KParams P{};
    for (int l = 0; l < e->L; ++l) {
        KLayer& K = P.layer[l];
        K.x = e->x[l]; K.m = e->m[l]; K.v = e->v[l];
        K.xgrad = r->update_x ? nullptr : r->xgrad[l];
        K.rec = r->rec_count > 0 ? r->rec_x[l] : nullptr;
        K.spill_e = l == 0 ? e->e0sum : e->spill_e[l];
        K.spill_a = e->spill_a[l];
        K.spill_e_tm = (l >= 1 && e->lin[l].spill_tm) ? 1 : 0;               // E_l is the error operand of Linear l,
        K.spill_a_tm = (l + 1 < nlin && e->lin[l + 1].spill_tm) ? 1 : 0;      // f(x_l) the activation operand of Linear l + 1
        K.ext_noise = nullptr;
        if (l >= 1) {
            K.Wf = (const f32x4*)e->lin[l].Wf; K.Wb = (const f32x4*)e->lin[l].Wb; K.bias = e->lin[l].bias_pad;
        }
        K.n = e->d.sizes[l]; K.npad = e->npad[l]; K.ntiles = e->npad[l] / 16;
        K.act = e->d.acts[l]; K.ecoef = e->d.ecoef[l];
        K.lds_a = e->lds_a[l]; K.lds_e = e->lds_e[l]; K.ld = e->npad[l] + kLdPad;
        K.lds_x = e->lds_x[l]; K.lds_bias = e->lds_bias[l];
    }
    if (e->has_head) {
        KHead& H = P.head;
        const Lin& ln = e->lin[e->L];
        H.Wf = (const f32x4*)ln.Wf; H.Wb = (const f32x4*)ln.Wb; H.bias = ln.bias_pad;
        H.ybits = e->ybits; H.y_binary = e->knobs.no_ybits ? e->y_binary + 1 : e->y_binary; H.y_bounded = e->y_binary + 2; H.ywords = e->ywords;
        H.y = e->ypad; H.ytile = e->ytile; H.rec_out = r->rec_count > 0 ? r->rec_out : nullptr; H.spill_e = e->spill_eo;
        H.spill_tm = e->lin[e->L].spill_tm ? 1 : 0;
        H.n = e->d.n_out; H.npad = e->out_pad; H.ntiles = e->out_pad / 16;
        H.loss_kind = r->loss_kind;
        H.inv_var = r->loss_kind == MCPC_LOSS_GAUSSIAN ? (float)(1.0 / r->loss_var) : 1.0f;
        H.mask_start = r->loss_kind == MCPC_LOSS_NONE ? 0 : r->mask_start;
        H.lds_eo = e->lds_eo; H.ld = kChunkTiles * 16 + kLdPad;
        H.lds_bias = e->lds_hbias; H.lds_yw = e->lds_yw;
    }
    P.mu1 = e->mu1; P.epart = e->epart; P.epart_slots = (int)eslots;
    P.phases = e->phases; P.n_phases = e->n_phases; P.wexp = e->wexp; P.spillmax = nullptr; P.lds_spillmax = e->lds_spillmax; P.lds_rowexp = e->lds_rowexp; P.g_first = e->g_first;
    P.stagger_cycles = e->knobs.stagger;
    P.L = e->L; P.has_head = e->has_head; P.B = e->d.batch; P.Bpad = e->Bpad; P.T = r->T;
    P.xopt = r->xopt_kind; P.update_x = r->update_x ? 1 : 0;
    P.lr = (float)r->lr; P.beta2 = (float)r->beta2;
    P.omb1 = (float)(1.0 - r->beta1); P.omb2 = (float)(1.0 - r->beta2); P.eps = (float)r->eps;
    P.noise_mode = r->update_x ? r->noise_mode : MCPC_NOISE_NONE;
    P.noise_scale = (float)std::sqrt(r->noise_var * r->lr);
    P.seed = r->seed; P.step_base = r->step_base; P.chain_base = r->chain_base;
    P.acc_begin = acc_b; P.acc_end = acc_e;
    P.energy_mode = r->energy_mode;
    P.rec_begin = r->rec_begin; P.rec_stride = std::max(r->rec_stride, 1); P.rec_count = r->rec_count;
    P.lds_red = e->lds_red;
    P.lds_ws_sync = e->lds_ws_sync;
    P.ws_prio = e->knobs.ws_prio;
    {
        int widest = e->out_pad;
        for (int l = 0; l < e->L; ++l) widest = std::max(widest, e->npad[l]);
        P.lean_ok = e->Bpad < (1 << 24) && (uint64_t)e->Bpad * (uint64_t)widest * 4u < (1ull << 32) && !e->knobs.no_lean;
    }
    P.err = e->err; P.dummy = e->dummy; P.lds_floats = e->lds_bytes / 4; P.lds_zero = e->lds_zero;
    P.clk = e->profiling ? e->clk : nullptr;
    P.xl = e->xl ? 1 : 0;
    {
        // bytes of Hebbian spill per step: beyond 8 MB (a quarter of the eight 4 MB L2s) the stores go out at system scope
        size_t per_step = (size_t)e->Bpad * e->out_pad;
        for (int l = 0; l < e->L; ++l) per_step += (size_t)e->Bpad * e->npad[l] * (l >= 1 ? 2 : 1);
        P.spill_sys = per_step * sizeof(float) > ((size_t)8 << 20) ? 1 : 0;
    }
#ifdef MCPC_STAMPS
    if (!e->dbg) { int rc = dmalloc(e->dbg, (size_t)e->nwg * 2 * kMaxWaves * 16); if (rc) return rc; }
    P.dbg = e->dbg;
#endif
    // The unified-wave kernel (mcpc_steps_u.h) serves the lean runs of an engine that holds its plan: fused SGD update with or without
    // the Philox kick, Adam without noise.  Everything else -- gradients-only runs, injected noise -- keeps the main plan's kernel.
    bool use_u = e->u.on && (e->u.prefer || (e->has_head && r->loss_kind == MCPC_LOSS_NONE)) && e->ws == 2 && r->update_x && P.lean_ok &&
                 ((r->xopt_kind == MCPC_XOPT_SGD && r->noise_mode != MCPC_NOISE_EXTERNAL) ||
                  (r->xopt_kind == MCPC_XOPT_ADAM && r->noise_mode == MCPC_NOISE_NONE));
    if (use_u) {
        const UPlan& u = e->u;
        for (int l = 0; l < e->L; ++l) {
            KLayer& K = P.layer[l];
            K.lds_a = u.lds_a[l]; K.lds_e = u.lds_e[l]; K.lds_x = u.lds_x[l]; K.lds_bias = u.lds_bias[l];
        }
        if (e->has_head) { P.head.lds_eo = u.lds_eo; P.head.ld = e->out_pad + kLdPad; P.head.lds_bias = u.lds_hbias; P.head.lds_yw = u.lds_yw; }
        P.phases = u.phases; P.n_phases = u.n_phases; P.g_first = 0;
        P.lds_spillmax = u.lds_spillmax; P.lds_rowexp = u.lds_rowexp; P.lds_red = u.lds_red; P.lds_ws_sync = u.lds_ws_sync;
        P.lds_floats = u.lds_bytes / 4; P.lds_zero = u.lds_zero; P.xl = 1;
    }

    // ---- step segments: non-accumulating stretches run as one persistent launch; accumulating
    //      stretches are cut at the spill ring's capacity and followed by a Hebbian flush ----------
    int t = r->t_begin;
    const int end = r->t_begin + r->n_steps;
    const bool overlap = e->aux != nullptr;
    int half = 0;                                         // part of the ring the next accumulating segment spills into
    const int n_parts = std::max(1, e->slots / std::max(1, e->half_slots));
    // HIP events around every launch of the step kernel (at most kMaxProfBrackets since profiling was switched on: a caller that
    // leaves it on forever stops collecting, it does not accumulate HIP events without bound)
    constexpr size_t kMaxProfBrackets = 1 << 16;
    bool bracket_open = false;
    auto prof_begin = [&]() -> int {
        bracket_open = false;
        if (!e->profiling) return 0;
        if (e->events_used >= kMaxProfBrackets) return 0;
        bracket_open = true;
        if (e->events_used == e->events.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return fail(MCPC_EHIP, "hipEventCreate failed");
            e->events.emplace_back(a, b);
        }
        return hipEventRecord(e->events[e->events_used].first, stream) == hipSuccess ? 0 : fail(MCPC_EHIP, "hipEventRecord failed");
    };
    auto prof_end = [&](double steps) -> int {
        if (!e->profiling || !bracket_open) return 0;
        if (hipEventRecord(e->events[e->events_used].second, stream) != hipSuccess) return fail(MCPC_EHIP, "hipEventRecord failed");
        ++e->events_used; e->prof_steps += steps;
        return 0;
    };
    // one cycle of the round schedule (setup_rounds): rr_k launches of q steps, every unit in rr_m of them; afterwards every
    // unit is at t0 + rr_m q.  `base` carries the spill pointers of the ring part in a Hebbian segment.
    bool rr_ok = e->rr && r->update_x;
#ifdef MCPC_STAMPS
    rr_ok = false;
#endif
    auto run_round_cycle = [&](const KParams& base, int t0, int q) -> int {
        KParams Q = base;
        Q.t0 = t0; Q.spill_t0 = t0; Q.n_steps = q; Q.rr_q = q;
        const int s0 = t0 - r->t_begin;
        Q.adam_coef = r->xopt_kind == MCPC_XOPT_ADAM ? e->adam_coef + 2 * (size_t)s0 : nullptr;
        if (r->noise_mode == MCPC_NOISE_EXTERNAL)
            for (int l = 0; l < e->L; ++l) Q.layer[l].ext_noise = r->ext_noise[l] + (size_t)s0 * e->d.batch * e->d.sizes[l];
        for (int i = 0; i < e->rr_k; ++i) {
            Q.wg_list = e->rr_tab + e->rr_off[i]; Q.wg_rel = Q.wg_list + e->rr_count[i];
            { const int rc = prof_begin(); if (rc) return rc; }
            if (use_u) hipLaunchKernelGGL((mcpc_steps_u_kernel<true>), dim3(e->rr_count[i]), dim3(kUThreads), e->u.lds_bytes, stream, Q);
            else hipLaunchKernelGGL((mcpc_steps_ws2_kernel<1, true>), dim3(e->rr_count[i]), dim3(kWs2Threads), e->lds_bytes, stream, Q);
            { const int rc = prof_end((double)q * e->rr_count[i] / e->nwg_live); if (rc) return rc; }
        }
        HIP_TRY(hipGetLastError());
        return 0;
    };
    while (t < end) {
        const bool in_acc = t >= acc_b && t < acc_e;
        int n;
        if (in_acc) {
            const int rem = std::min(end, acc_e) - t;
            n = std::min(rem, e->half_slots);
            // the flush of a stretch's LAST segment has no step kernel to hide behind: keep that segment short
            const int tail = e->knobs.flush_tail;
            if (overlap && tail > 0 && rem <= e->half_slots && rem >= 2 * tail && std::min(end, acc_e) == acc_e) n = rem - tail;
        }
        else n = (t < acc_b ? std::min(end, acc_b) : end) - t;
        // a launch's row-exponent words carry their generation -- the step of the launch, or step x entries + entry for a ring slot -- in 24
        // bits (mcpc_kernels.h: rowexp_track): longer stretches are cut into several launches (a wrapped generation would never
        // supersede the stale word)
        n = std::min(n, ((1 << 24) - 2) / std::max(std::max(e->n_phases, e->u.n_phases), 1));
        int rr_q = 0;
        if (rr_ok && in_acc) {
            rr_q = n / e->rr_m;                               // a Hebbian segment is one cycle (fewer steps than rr_m left: plain launch)
            if (rr_q >= 1) n = rr_q * e->rr_m;
        } else if (rr_ok) {
            // whole cycles, longest launches first; fewer than rr_m steps left run as one plain launch (hardware rounds)
            while (n >= e->rr_m) {
                const int q = std::min(std::max(1, e->knobs.rr_qmax), n / e->rr_m);
                const int rc = run_round_cycle(P, t, q);
                if (rc) return rc;
                t += q * e->rr_m; n -= q * e->rr_m;
            }
            if (n == 0) continue;
        }
        const int slot0 = in_acc && overlap ? half * e->half_slots : 0;
        P.t0 = t; P.n_steps = n; P.spill_t0 = t;
        const int s0 = t - r->t_begin;
        P.adam_coef = r->xopt_kind == MCPC_XOPT_ADAM ? e->adam_coef + 2 * (size_t)s0 : nullptr;
        if (r->noise_mode == MCPC_NOISE_EXTERNAL)
            for (int l = 0; l < e->L; ++l) P.layer[l].ext_noise = r->ext_noise[l] + (size_t)s0 * e->d.batch * e->d.sizes[l];
        if (in_acc) {
            for (int l = 0; l < e->L; ++l) {
                const size_t off = (size_t)slot0 * e->Bpad * e->npad[l];
                P.layer[l].spill_a = e->spill_a[l] + off;
                if (l >= 1) P.layer[l].spill_e = e->spill_e[l] + off;
            }
            if (e->has_head) P.head.spill_e = e->spill_eo + (size_t)slot0 * e->Bpad * e->out_pad;
            // this half of the ring may still be read by the flush that was started two segments ago
            if (overlap && e->flush_pending[half]) { HIP_TRY(hipStreamWaitEvent(stream, e->ev_flush[half], 0)); e->flush_pending[half] = false; }
            // the segment's largest |value| per spilled tensor: starts at zero, raised by the step kernel's workgroups, read by the flush
            P.spillmax = e->spillmax + (size_t)(overlap ? half : 0) * kSpillTensors;
            HIP_TRY(hipMemsetAsync(P.spillmax, 0, kSpillTensors * sizeof(unsigned), stream));
        } else {
            P.spillmax = nullptr;
        }
        if (rr_q >= 1) {
            const int rc = run_round_cycle(P, t, rr_q);
            if (rc) return rc;
        } else {
        { const int rc = prof_begin(); if (rc) return rc; }
        if (use_u) hipLaunchKernelGGL((mcpc_steps_u_kernel<false>), dim3(e->nwg_live), dim3(kUThreads), e->u.lds_bytes, stream, P);
        else if (e->ws == 2) hipLaunchKernelGGL((mcpc_steps_ws2_kernel<1>), dim3(e->nwg_live), dim3(kWs2Threads), e->lds_bytes, stream, P);
        else hipLaunchKernelGGL((mcpc_steps_kernel<1, 4>), dim3(e->nwg), dim3(256), e->lds_bytes, stream, P);
        { const int rc = prof_end((double)n); if (rc) return rc; }
        }
        HIP_TRY(hipGetLastError());
#ifdef MCPC_STAMPS
        {
            static const char* names_ws2[16] = {"G top", "G wait deps", "G HEADB gemm", "G gemm", "G prefetch next", "G store+publish",
                                                "(launch, s_memtime)", "(launch, 100 MHz)", "E other", "E loads", "E wait block", "E epilogue FWD", "E epilogue HEADF", "E epilogue BWD", "-", "-"};
            static const char* names[16] = {"FWD prologue", "FWD gemm", "FWD epilogue", "HEADF prologue", "HEADF gemm", "HEADF epilogue",
                                            "HEADB prologue", "HEADB gemm", "HEADB (acc->b)", "BWD prologue", "BWD gemm", "BWD epilogue",
                                            "energy", "barrier", "-", "-"};
            HIP_TRY(hipStreamSynchronize(stream));
            std::vector<unsigned long long> h((size_t)e->nwg * e->nw * 16);
            HIP_TRY(hipMemcpy(h.data(), e->dbg, h.size() * 8, hipMemcpyDeviceToHost));
#ifdef MCPC_STAMPS_ENTRY
            {
                // per table entry: mean / max over the G (q = 1, 2) or E (q = 3, 4) waves, cycles per step
                static const char* what[5] = {"", "G wait (deps)", "G total", "E wait for block", "E total"};
                static const char* tname[5] = {"FWD", "HEADF", "HEADB", "BWD", "ENERGY"};
                std::vector<KPhase> tab(e->n_phases);
                HIP_TRY(hipMemcpy(tab.data(), e->phases, tab.size() * sizeof(KPhase), hipMemcpyDeviceToHost));
                const bool want_g = MCPC_STAMPS_ENTRY <= 2;
                fprintf(stderr, "[stamps] launch t0=%d n=%d: %s per table entry, cycles per step (mean / max over the %s waves)\n", t, n,
                        what[MCPC_STAMPS_ENTRY], want_g ? "G" : "E");
                double tot = 0;
                for (int i = 0; i < 16 && i < e->n_phases; ++i) {
                    double sum = 0, mx = 0; size_t cnt = 0;
                    for (size_t w = 0; w < (size_t)e->nwg * e->nw; ++w) {
                        const bool is_g = (int)(w % e->nw) < e->nw / 2;
                        if (is_g != want_g) continue;
                        const double v = (double)h[w * 16 + i];
                        sum += v; mx = std::max(mx, v); ++cnt;
                    }
                    tot += sum / cnt / n;
                    fprintf(stderr, "[stamps]   entry %2d %-6s layer %d tiles %2d nkb %2d  mean %8.0f  max %8.0f\n", i, tname[tab[i].type], tab[i].layer,
                            tab[i].ntiles, tab[i].nkb, sum / cnt / n, mx / n);
                }
                fprintf(stderr, "[stamps]   sum over entries: %.0f cycles per step\n", tot);
            }
#else
            double tot = 0, sum[16] = {0}, mx[16] = {0};
            for (size_t w = 0; w < (size_t)e->nwg * e->nw; ++w)
                for (int i = 0; i < 16; ++i) { sum[i] += (double)h[w * 16 + i]; mx[i] = std::max(mx[i], (double)h[w * 16 + i]); }
            static const char* names_u[16] = {"top of entry", "barrier", "gemm", "prefetch next", "epilogue", "-", "-", "presplit", "gemm fwd nt=1", "gemm fwd nt=2", "gemm fwd nt=3", "gemm fwd nt=4", "gemm bwd nt=1", "gemm bwd nt=2", "gemm bwd nt=3", "gemm bwd nt=4"};
            if (e->ws == 2 && !use_u) {
                fprintf(stderr, "[stamps] shader clock during the launch = %.3f GHz (s_memtime ticks per 100 MHz wall-clock tick)\n", sum[6] / sum[7] * 0.1);
                sum[6] = sum[7] = 0;
            }
            for (int i = 0; i < 16; ++i) tot += sum[i];
            fprintf(stderr, "[stamps] launch t0=%d n=%d: mean cycles/step/wave = %.0f\n", t, n, tot / (e->nwg * e->nw) / n);
            for (int i = 0; i < 16; ++i)
                fprintf(stderr, "[stamps]   %-18s %5.1f%%  mean %8.0f  max %8.0f cycles/step\n", (use_u ? names_u : e->ws == 2 ? names_ws2 : names)[i], 100.0 * sum[i] / tot,
                        sum[i] / (e->nwg * e->nw) / n, mx[i] / n);
#endif
        }
#endif
        if (in_acc) {
            if (overlap) {
                // the Hebbian GEMMs of this half run on the low-priority stream while the next segment steps
                HIP_TRY(hipEventRecord(e->ev_steps[half], stream));
                HIP_TRY(hipStreamWaitEvent(e->aux, e->ev_steps[half], 0));
                int rc = flush_spill(e, n, slot0, e->aux, half);
                if (rc) return rc;
                HIP_TRY(hipEventRecord(e->ev_flush[half], e->aux));
                e->flush_pending[half] = true;
                half = (half + 1) % n_parts;
            } else {
                int rc = flush_spill(e, n, 0, stream, 0);
                if (rc) return rc;
            }
        }
        t += n;
    }
    // everything that follows on the caller's stream (dw0, gradient read-out, the next run) sees finished sums
    for (int h = 0; h < kMaxRingParts; ++h)
        if (overlap && e->flush_pending[h]) { HIP_TRY(hipStreamWaitEvent(stream, e->ev_flush[h], 0)); e->flush_pending[h] = false; }
    if (run_accumulates) {
        // Linear 0 sees a constant input: fold sum_t e_1 now, then clear the running sum
        Lin& l0 = e->lin[0];
        // one block per (unit, input column | bias): a fixed-order tree over the chains
        hipLaunchKernelGGL(mcpc_dw0_kernel, dim3(l0.n_out, e->inputs != nullptr ? l0.n_in + 1 : 1), dim3(256), 0, stream, e->e0sum,
                           e->inputs, l0.G, l0.Gb, e->d.batch, l0.n_out, e->npad[0], l0.n_in, l0.g_ld);
        HIP_TRY(hipMemsetAsync(e->e0sum, 0, (size_t)e->Bpad * e->npad[0] * 4, stream));
    }
    if (r->energy_mode == MCPC_ENERGY_ALL) {
        hipLaunchKernelGGL(mcpc_energy_reduce_kernel, dim3((r->n_steps + 3) / 4), dim3(256), 0, stream,
                           e->epart + (size_t)r->t_begin * eslots * (kMaxLatent + 1),
                           r->energies_out + (size_t)r->t_begin * kEnergyCols, r->n_steps, (int)eslots, e->L);
    } else if (r->energy_mode == MCPC_ENERGY_LAST && end == r->T) {
        hipLaunchKernelGGL(mcpc_energy_reduce_kernel, dim3(1), dim3(256), 0, stream, e->epart, r->energies_out, 1, (int)eslots, e->L);
    }
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

int mcpc_read_param_grads(mcpc_engine* e, int j, float* dW, float* db, float scale, int accumulate, void* stream_) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    const int nlin = e->L + (e->has_head ? 1 : 0);
    if (j < 0 || j >= nlin) return fail(MCPC_EINVAL, "Linear index %d out of range", j);
    if (!dW) return fail(MCPC_EINVAL, "null dW");
    HIP_TRY(hipSetDevice(e->d.device));
    const Lin& ln = e->lin[j];
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(mcpc_export_grad_kernel, dim3(grid_for((size_t)ln.n_out * ln.n_in)), dim3(256), 0, stream, ln.G, dW,
                       ln.n_out, ln.n_in, ln.g_ld, scale, accumulate);
    if (db)
        hipLaunchKernelGGL(mcpc_export_grad_kernel, dim3(grid_for((size_t)ln.n_out)), dim3(256), 0, stream, ln.Gb, db,
                           ln.n_out, 1, 1, scale, accumulate);
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

// ---- multi-GPU: the one collective of a learning call ---------------------------------------------
int mcpc_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == MCPC_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id_out) return fail(MCPC_EINVAL, "null id buffer");
    if (const int rc = rccl_load()) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(MCPC_EHIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    std::memcpy(id_out, &id, sizeof(id));
    return MCPC_OK;
}

int mcpc_comm_init(mcpc_engine* e, int n_ranks, int rank, const void* id_in) {
    if (!e || !id_in) return fail(MCPC_EINVAL, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MCPC_EINVAL, "rank %d of %d", rank, n_ranks);
    if (e->comm) return fail(MCPC_ESTATE, "the engine already has a communicator (mcpc_comm_destroy first)");
    if (const int rc = rccl_load()) return rc;
    HIP_TRY(hipSetDevice(e->d.device));
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&e->comm, n_ranks, id, rank);
    if (r != ncclSuccess) { e->comm = nullptr; return fail(MCPC_EHIP, "ncclCommInitRank(%d of %d): %s", rank, n_ranks, g_rccl.GetErrorString(r)); }
    e->comm_ranks = n_ranks;
    return MCPC_OK;
}

int mcpc_allreduce_grads(mcpc_engine* e, float* flat, int64_t n_floats, void* stream_) {
    if (!e || !flat) return fail(MCPC_EINVAL, "null argument");
    if (!e->comm) return fail(MCPC_ESTATE, "mcpc_allreduce_grads before mcpc_comm_init");
    if (n_floats != mcpc_param_count(e)) return fail(MCPC_EINVAL, "bucket of %lld floats, the engine's parameters have %lld", (long long)n_floats, (long long)mcpc_param_count(e));
    HIP_TRY(hipSetDevice(e->d.device));
    const ncclResult_t r = g_rccl.AllReduce(flat, flat, (size_t)n_floats, ncclFloat, ncclSum, e->comm, (hipStream_t)stream_);
    if (r != ncclSuccess) return fail(MCPC_EHIP, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    return MCPC_OK;
}

int mcpc_comm_destroy(mcpc_engine* e) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    if (e->comm) {
        (void)hipSetDevice(e->d.device);
        const ncclResult_t r = g_rccl.CommDestroy(e->comm);
        e->comm = nullptr; e->comm_ranks = 0;
        if (r != ncclSuccess) return fail(MCPC_EHIP, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    }
    return MCPC_OK;
}

int64_t mcpc_param_count(const mcpc_engine* e) {
    if (!e) return 0;
    const int nlin = e->L + (e->has_head ? 1 : 0);
    int64_t n = 0;
    for (int j = 0; j < nlin; ++j) n += (int64_t)e->lin[j].n_out * e->lin[j].n_in + (e->lin[j].bias ? e->lin[j].n_out : 0);
    return n;
}

int mcpc_read_param_grads_flat(mcpc_engine* e, float* flat, int64_t n_floats, float scale, void* stream_) {
    if (!e || !flat) return fail(MCPC_EINVAL, "null argument");
    if (n_floats != mcpc_param_count(e)) return fail(MCPC_EINVAL, "flat buffer holds %lld floats, parameters need %lld", (long long)n_floats, (long long)mcpc_param_count(e));
    const int nlin = e->L + (e->has_head ? 1 : 0);
    int64_t off = 0;
    for (int j = 0; j < nlin; ++j) {
        const Lin& ln = e->lin[j];
        float* dW = flat + off; off += (int64_t)ln.n_out * ln.n_in;
        float* db = nullptr;
        if (ln.bias) { db = flat + off; off += ln.n_out; }
        int rc = mcpc_read_param_grads(e, j, dW, db, scale, 0, stream_);
        if (rc) return rc;
    }
    return MCPC_OK;
}

int mcpc_philox_normals(int device, uint64_t seed, uint64_t step, int layer, uint64_t chain_base, int batch, int n_units,
                        float* out, int raw, void* stream_) {
    if (!out || batch < 1 || n_units < 1 || layer < 0 || layer > 255) return fail(MCPC_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const size_t total = (size_t)batch * ((n_units + 3) / 4);
    hipLaunchKernelGGL(mcpc_philox_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream_, seed, step, layer, chain_base,
                       batch, n_units, out, raw);
    HIP_TRY(hipGetLastError());
    return MCPC_OK;
}

int mcpc_query(const mcpc_engine* e, int32_t* lds_bytes, int32_t* chains_per_wg, int32_t* n_workgroups, int32_t* spill_slots) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    if (lds_bytes) *lds_bytes = e->lds_bytes;
    if (chains_per_wg) *chains_per_wg = e->ct;
    if (n_workgroups) *n_workgroups = e->nwg_live;
    if (spill_slots) *spill_slots = e->slots;
    return MCPC_OK;
}

const char* mcpc_step_kernel_name(const mcpc_engine* e) {
    if (!e) return "";
    // (an engine that holds the unified-wave kernel's plan runs its fused calls -- what a benchmark times -- on that kernel)
    if (e->u.prefer) return e->rr ? e->u_rr_name.c_str() : "mcpc::mcpc_steps_u_kernel<false>";
    if (e->rr) return e->rr_name.c_str();
    if (e->ws == 2) return "mcpc::mcpc_steps_ws2_kernel<1, false>";
    return "mcpc::mcpc_steps_kernel<1, 4>";
}

int mcpc_sync_check(mcpc_engine* e, void* stream_) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->d.device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, e->err, sizeof(int), hipMemcpyDeviceToHost));
    if (flag != 0) {
        (void)hipMemset(e->err, 0, sizeof(int));
        return fail(MCPC_ESTATE, "the step kernel reported error word 0x%x (a workgroup-internal progress wait ran out): results of the last run are invalid", flag);
    }
    return MCPC_OK;
}

int mcpc_last_shader_clock_ghz(mcpc_engine* e, float* ghz) {
    if (!e || !ghz) return fail(MCPC_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, e->clk, sizeof h, hipMemcpyDeviceToHost));              // (waits for the device)
    *ghz = h[1] ? (float)((double)h[0] / (double)h[1] * 0.1) : 0.f;
    return MCPC_OK;
}

// ---- diagnostic: poison the LDS of every CU (include/mcpc.h: mcpc_debug_poison_lds) -------------------------------------------
}  // extern "C"
namespace {
// One workgroup per CU at a time (it takes the whole 160 KiB of LDS): fills it with `word`, then holds its CU for ~20 us so that
// the dispatcher has to place the other workgroups of the launch on the other CUs.  `seen[xcc * 64 + cu]` counts the visits.
__global__ __launch_bounds__(256) void mcpc_poison_lds_kernel(uint32_t word, int* seen) {
    extern __shared__ uint32_t poison_lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) poison_lds[i] = word;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // HW_ID: cu_id bits 11:8, sh_id bit 12, se_id bits 15:13 (gfx9)
        const int cu = (int)((hw >> 8) & 0xff);
        atomicAdd(&seen[(xcc & 7) * 256 + cu], 1);
        const unsigned long long t_end = wall_clock64() + 2000;          // 100 MHz ticks: 20 us
        while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    // keep the stores observable: a read the compiler cannot drop
    if (poison_lds[(threadIdx.x * 97) % (160 * 1024 / 4)] != word) atomicAdd(&seen[8 * 256], 1);
}
}  // namespace
extern "C" {

int mcpc_debug_poison_lds(int device, uint32_t word, void* stream_) {
    HIP_TRY(hipSetDevice(device));
    int n_cu = 256;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n_cu <= 0) n_cu = 256;
    static bool attr_set[16] = {false};
    if (device >= 0 && device < 16 && !attr_set[device]) {
        HIP_TRY(hipFuncSetAttribute((const void*)mcpc_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[device] = true;
    }
    int* seen = nullptr;
    HIP_TRY(hipMalloc((void**)&seen, (8 * 256 + 1) * sizeof(int)));
    hipStream_t stream = (hipStream_t)stream_;
    int rc = MCPC_OK;
    // two passes of four workgroups per CU: a CU that was still draining another kernel during the first is caught by the second
    for (int pass = 0; pass < 2 && rc == MCPC_OK; ++pass) {
        if (hipMemsetAsync(seen, 0, (8 * 256 + 1) * sizeof(int), stream) != hipSuccess) { rc = fail(MCPC_EHIP, "hipMemsetAsync failed"); break; }
        hipLaunchKernelGGL(mcpc_poison_lds_kernel, dim3(4 * n_cu), dim3(256), 160 * 1024, stream, word, seen);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) rc = fail(MCPC_EHIP, "the LDS poison kernel failed");
    }
    if (rc == MCPC_OK) {
        std::vector<int> h(8 * 256 + 1);
        if (hipMemcpy(h.data(), seen, h.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MCPC_EHIP, "hipMemcpy failed");
        else {
            int visited = 0;
            for (int i = 0; i < 8 * 256; ++i) visited += h[i] > 0;
            if (visited < n_cu) rc = fail(MCPC_ESTATE, "LDS poison reached %d of %d compute units", visited, n_cu);
            else if (h[8 * 256] != 0) rc = fail(MCPC_ESTATE, "LDS poison read-back mismatch");
        }
    }
    (void)hipFree(seen);
    return rc;
}

int mcpc_set_profiling(mcpc_engine* e, int enable) {
    if (!e) return fail(MCPC_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->d.device));
    e->profiling = enable != 0;
    e->events_used = 0;
    e->prof_steps = 0;
    if (enable) HIP_TRY(hipMemset(e->clk, 0, 2 * sizeof(unsigned long long)));      // (synchronises with the device: a diagnostic call)
    return MCPC_OK;
}

namespace {
int sum_events(const std::vector<std::pair<hipEvent_t, hipEvent_t>>& ev, size_t used, float* total) {
    *total = 0.f;
    for (size_t i = 0; i < used; ++i) {
        HIP_TRY(hipEventSynchronize(ev[i].second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ev[i].first, ev[i].second));
        *total += t;
    }
    return 0;
}
}  // namespace

int mcpc_last_step_kernel_ms(mcpc_engine* e, float* ms, int32_t* n_launches, int64_t* n_steps) {
    if (!e || !ms) return fail(MCPC_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    const int rc = sum_events(e->events, e->events_used, ms);
    if (rc) return rc;
    if (n_launches) *n_launches = (int32_t)e->events_used;
    if (n_steps) *n_steps = (int64_t)std::llround(e->prof_steps);
    return MCPC_OK;
}

}  // extern "C"
