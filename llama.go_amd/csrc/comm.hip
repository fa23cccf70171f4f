// csrc/comm.hip — multi-GPU side of the C-ABI: RCCL point-to-point behind lh_comm_*, and the pods pipeline scheduler
// (lh_pipeline_*) that SURVEY §8f row 3 places "in the C layer".  Reference: the reference's only parallel dimension is
// request-level pods (pkg/server/server.go:84-106 Engine, :151 one llama.Context per Do); layers shard in contiguous blocks
// (pkg/llama/llama.go:246-370 touches only layer il's weights and KV slice), so the one exchange per stage boundary is the
// fp32 residual stream [n x embd] (llama.go:369), plus the 4-byte token id from the last rank back to rank 0.
//
// librccl.so.1 is dlopen'ed on first use: a process that already carries an RCCL (e.g. PyTorch's bundled one, same SONAME)
// gets that very instance, a bare C/Go host gets /opt/rocm/lib's through this library's RUNPATH.  No all-reduce exists on
// this path, so nothing here is ring-bandwidth bound: every hop uses exactly one xGMI link.
#include "plan.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>

namespace lh {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    std::string err;
};

static Rccl* rccl(lh_ctx* ctx) {
    static std::mutex mu;
    static Rccl r;
    std::lock_guard<std::mutex> lk(mu);
    if (r.handle) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) { set_error(ctx, "RCCL not available: %s", dlerror()); return nullptr; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p) ok = false; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");   // optional
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    if (!ok) { set_error(ctx, "RCCL library lacks a required symbol"); dlclose(r.handle); r.handle = nullptr; return nullptr; }
    return &r;
}

#define LH_NCCL(ctx, R, expr)                                                                                   \
    do {                                                                                                        \
        ncclResult_t e__ = (expr);                                                                              \
        if (e__ != ncclSuccess) {                                                                               \
            lh::set_error(ctx, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(e__), __FILE__, __LINE__);   \
            return LH_EHIP;                                                                                     \
        }                                                                                                       \
    } while (0)

// ---- the schedule (pure host arithmetic) ------------------------------------------------------------------
// Q = max(pods, world) ticks per unit round.  Rank r is active in tick t iff k = t - r satisfies 0 <= k, k / Q < units and
// k % Q < pods; it then evaluates stream k % Q at unit k / Q.  The token a stream needs for unit u + 1 is produced by the
// last rank in tick u*Q + p + world - 1 and needed by rank 0 in tick (u+1)*Q + p: Q >= world makes that strictly later.
static inline bool tick_of(uint32_t t, uint32_t r, uint32_t Q, uint32_t pods, uint32_t units, int32_t* stream, int32_t* unit) {
    *stream = -1; *unit = -1;
    if (t < r) return false;
    const uint32_t k = t - r;
    if (k / Q >= units || k % Q >= pods) return false;
    *stream = (int32_t)(k % Q);
    *unit = (int32_t)(k / Q);
    return true;
}
static inline uint32_t tick_count(uint32_t world, uint32_t pods, uint32_t units) {
    if (!units || !pods || !world) return 0;
    const uint32_t Q = std::max(pods, world);
    return Q * (units - 1) + pods + world - 1;
}

template <typename StageFn, typename ExchangeFn>
static int run_ticks(uint32_t rank, uint32_t world, uint32_t pods, uint32_t units, StageFn&& stage, ExchangeFn&& exchange) {
    const uint32_t Q = std::max(pods, world), total = tick_count(world, pods, units), prev = (rank + world - 1) % world;
    for (uint32_t t = 0; t < total; ++t) {
        int32_t s, u, rs, ru;
        const bool active = tick_of(t, rank, Q, pods, units, &s, &u);
        const bool recv = tick_of(t, prev, Q, pods, units, &rs, &ru);
        int rc;
        if (active && (rc = stage((uint32_t)s, (uint32_t)u))) return rc;
        if ((active || recv) && (rc = exchange(s, u, rs, ru))) return rc;
    }
    return 0;
}

}  // namespace lh

using namespace lh;

struct lh_comm {
    lh_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    Rccl* r = nullptr;
    lh_comm_hooks hooks = {};
    bool use_hooks = false;
    bool aborted = false;
    // host staging of the hooks transport (pinned)
    char *send_host = nullptr, *recv_host = nullptr;
    uint64_t send_cap = 0, recv_cap = 0;
};

// One scheduling unit of the pipeline: the pods of a rank that take a tick together (lh_batch: one weight pass for all of them).
struct Group {
    std::vector<uint32_t> pods;     // stream indices, ascending
    lh_batch* batch = nullptr;
    float *x_in = nullptr, *x_out = nullptr;   // residual rows received / produced, rows_cap x embd
    uint32_t rows_cap = 0;
    uint32_t* hist = nullptr;       // first / last rank: ids per unit [units][B] (rank 0: received from the last rank; last rank: produced)
    uint32_t n_hist = 0, hist_cap = 0;   // units recorded / capacity in units
};

struct lh_pipeline {
    lh_ctx* ctx = nullptr;
    lh_comm* comm = nullptr;
    int rank = 0, world = 1;
    std::vector<lh_llama*> pods;
    std::vector<uint32_t> past;     // per stream: position of the next unit (host mirror; every rank keeps the same values)
    std::vector<uint32_t> group_of, row_of;
    std::vector<Group> groups;
    uint32_t d = 0, ctx_size = 0, vocab = 0;
    bool started = false;           // a prompt has been run: there is a token to continue from (the same on every rank)
    bool sampling = false;
    // lh_pipeline_profile: HIP events around every stage and every exchange of a run (where a shortfall of the N > 1 curve comes from:
    // the rank's own compute or the hops)
    // context swap across ranks (server.go:160-172): KeepCount (the same on every rank) and, on rank 0, every stream's prompt - together with the
    // ids it received (Group::hist) the contents of the reference's lastNTokens ring, oldest first
    uint32_t keep = 0;
    std::vector<std::vector<uint32_t>> prompt_host;
    bool profiling = false;
    std::vector<hipEvent_t> ev;     // 3 per tick: before the stage, behind it, behind the exchange
    uint32_t ev_used = 0;
    lh_pipeline_stats stats = {};
};

extern "C" {

int lh_comm_unique_id(lh_ctx* ctx, uint8_t id[LH_COMM_ID_BYTES]) {
    if (!id) LH_FAIL(ctx, LH_EINVAL, "lh_comm_unique_id: NULL id");
    Rccl* r = rccl(ctx);
    if (!r) return LH_EUNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == LH_COMM_ID_BYTES, "unique id size");
    ncclUniqueId u;
    LH_NCCL(ctx, r, r->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return LH_OK;
}

int lh_comm_init(lh_ctx* ctx, int rank, int world, const uint8_t id[LH_COMM_ID_BYTES], lh_comm** out) {
    if (!ctx || !id || !out) LH_FAIL(ctx, LH_EINVAL, "lh_comm_init: NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) LH_FAIL(ctx, LH_EINVAL, "lh_comm_init: rank %d outside world %d", rank, world);
    Rccl* r = rccl(ctx);
    if (!r) return LH_EUNSUPPORTED;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    LH_NCCL(ctx, r, r->CommInitRank(&c, world, u, rank));
    lh_comm* cm = new lh_comm();
    cm->ctx = ctx; cm->rank = rank; cm->world = world; cm->nccl = c; cm->r = r;
    *out = cm;
    return LH_OK;
}

int lh_comm_init_hooks(lh_ctx* ctx, int rank, int world, const lh_comm_hooks* hooks, lh_comm** out) {
    if (!ctx || !hooks || !hooks->exchange || !out) LH_FAIL(ctx, LH_EINVAL, "lh_comm_init_hooks: NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) LH_FAIL(ctx, LH_EINVAL, "lh_comm_init_hooks: rank %d outside world %d", rank, world);
    lh_comm* cm = new lh_comm();
    cm->ctx = ctx; cm->rank = rank; cm->world = world; cm->hooks = *hooks; cm->use_hooks = true;
    *out = cm;
    return LH_OK;
}

void lh_comm_destroy(lh_comm* cm) {
    if (!cm) return;
    hipSetDevice(cm->ctx->device);
    hipStreamSynchronize(cm->ctx->stream);
    if (cm->nccl) cm->r->CommDestroy(cm->nccl);
    if (cm->send_host) hipHostFree(cm->send_host);
    if (cm->recv_host) hipHostFree(cm->recv_host);
    delete cm;
}

// Tear the communicator down without waiting for peers (ncclCommAbort): operations the peers have pending against this rank fail
// instead of blocking.  The handle stays valid for lh_comm_destroy only.
int lh_comm_abort(lh_comm* cm) {
    if (!cm) return LH_EINVAL;
    if (cm->nccl && cm->r->CommAbort) { cm->r->CommAbort(cm->nccl); cm->nccl = nullptr; }
    if (cm->use_hooks && cm->hooks.abort && !cm->aborted) cm->hooks.abort(cm->hooks.user);   // the transport's own way of failing the peers' pending receives
    cm->aborted = true;
    return LH_OK;
}

int lh_comm_rank(const lh_comm* cm) { return cm ? cm->rank : 0; }
int lh_comm_world(const lh_comm* cm) { return cm ? cm->world : 1; }

int lh_comm_exchange(lh_comm* cm, const void* send_dev, uint64_t send_bytes, int send_peer, void* recv_dev, uint64_t recv_bytes, int recv_peer) {
    if (!cm) return LH_EINVAL;
    lh_ctx* ctx = cm->ctx;
    const bool snd = send_dev && send_bytes, rcv = recv_dev && recv_bytes;
    if (!snd && !rcv) return LH_OK;
    if ((snd && (send_peer < 0 || send_peer >= cm->world)) || (rcv && (recv_peer < 0 || recv_peer >= cm->world)))
        LH_FAIL(ctx, LH_EINVAL, "lh_comm_exchange: peer outside the world of %d", cm->world);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (cm->aborted) LH_FAIL(ctx, LH_EHIP, "lh_comm_exchange: the communicator was aborted");
    if (!cm->use_hooks) {
        Rccl* r = cm->r;
        LH_NCCL(ctx, r, r->GroupStart());
        ncclResult_t e1 = snd ? r->Send(send_dev, send_bytes, ncclUint8, send_peer, cm->nccl, ctx->stream) : ncclSuccess;
        ncclResult_t e2 = rcv ? r->Recv(recv_dev, recv_bytes, ncclUint8, recv_peer, cm->nccl, ctx->stream) : ncclSuccess;
        ncclResult_t e3 = r->GroupEnd();
        if (e1 != ncclSuccess || e2 != ncclSuccess || e3 != ncclSuccess)
            LH_FAIL(ctx, LH_EHIP, "RCCL send/recv failed: %s", r->GetErrorString(e1 != ncclSuccess ? e1 : e2 != ncclSuccess ? e2 : e3));
        return LH_OK;
    }
    // hooks transport: stage through pinned host memory, hand both directions to one call
    auto grow = [&](char** p, uint64_t* cap, uint64_t need) -> int {
        if (need <= *cap) return 0;
        if (*p) LH_HIP(ctx, hipHostFree(*p));
        *p = nullptr; *cap = 0;
        LH_HIP(ctx, hipHostMalloc((void**)p, need * 2, hipHostMallocDefault));
        *cap = need * 2;
        return 0;
    };
    int rc;
    if (snd) {
        if ((rc = grow(&cm->send_host, &cm->send_cap, send_bytes))) return rc;
        LH_HIP(ctx, hipMemcpyAsync(cm->send_host, send_dev, send_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (rcv && (rc = grow(&cm->recv_host, &cm->recv_cap, recv_bytes))) return rc;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cm->hooks.exchange(cm->hooks.user, snd ? cm->send_host : nullptr, snd ? send_bytes : 0, send_peer, rcv ? cm->recv_host : nullptr, rcv ? recv_bytes : 0, recv_peer))
        LH_FAIL(ctx, LH_EHIP, "lh_comm_exchange: the transport hook reported an error");
    if (rcv) {
        LH_HIP(ctx, hipMemcpyAsync(recv_dev, cm->recv_host, recv_bytes, hipMemcpyHostToDevice, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next tick
    }
    return LH_OK;
}

// ---- schedule ---------------------------------------------------------------------------------------------------------
int lh_pipeline_schedule(uint32_t rank, uint32_t world, uint32_t pods, uint32_t units, lh_tick* out, uint32_t cap) {
    if (!world || rank >= world || !pods) return LH_EINVAL;
    const uint32_t total = tick_count(world, pods, units), Q = std::max(pods, world), prev = (rank + world - 1) % world;
    for (uint32_t t = 0; t < total && t < cap && out; ++t) {
        lh_tick& k = out[t];
        k.t = t;
        tick_of(t, rank, Q, pods, units, &k.stream, &k.unit);
        tick_of(t, prev, Q, pods, units, &k.recv_stream, &k.recv_unit);
    }
    return (int)total;
}

int lh_pipeline_run_hooks(uint32_t rank, uint32_t world, uint32_t pods, uint32_t units, const lh_pipeline_hooks* h) {
    if (!h || !h->stage || !h->exchange || !world || rank >= world || !pods) return LH_EINVAL;
    return run_ticks(rank, world, pods, units, [&](uint32_t s, uint32_t u) { return h->stage(h->user, s, u); },
                     [&](int32_t s, int32_t u, int32_t rs, int32_t ru) { return h->exchange(h->user, s, u, rs, ru); });
}

// ---- the product pipeline -------------------------------------------------------------------------------------------------
// Grouping: G = min(pods, world) groups keep every rank busy (a group is in flight on one rank at a time); more when a group would
// exceed max_rows rows.  Stream p belongs to group p * G / pods (contiguous blocks).
static uint32_t group_count(uint32_t pods, uint32_t world, uint32_t max_rows) {
    uint32_t G = std::min(pods, world);
    while ((pods + G - 1) / G > max_rows) ++G;
    return G;
}

uint32_t lh_pipeline_group_count(uint32_t pods, uint32_t world, uint32_t max_rows_per_tick) {
    if (!pods || !world) return 0;
    return group_count(pods, world, max_rows_per_tick ? max_rows_per_tick : 64u);
}

int lh_pipeline_create_grouped(lh_ctx* ctx, lh_comm* comm, lh_llama* const* pods, uint32_t n_pods, uint32_t max_rows, lh_pipeline** out) {
    if (!ctx || !pods || !n_pods || !out) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: NULL argument");
    *out = nullptr;
    if (comm && comm->ctx != ctx) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: the communicator belongs to another context (one stream must order compute and p2p)");
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    lh_pipeline* pl = new lh_pipeline();
    pl->ctx = ctx; pl->comm = comm; pl->rank = rank; pl->world = world;
    for (uint32_t i = 0; i < n_pods; ++i) {
        lh_llama* m = pods[i];
        if (!m || m->ctx != ctx) { delete pl; LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: pod %u lives on another context", i); }
        const ModelDesc& md = m->plan->md;
        if (md.first_stage() != (rank == 0) || md.last_stage() != (rank == world - 1)) {
            delete pl;
            LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: pod %u holds layers [%u,%u) of %u, which is not rank %d of %d in a contiguous layer shard", i, md.layer0, md.layer1, md.L, rank, world);
        }
        if (i == 0) { pl->d = md.d; pl->ctx_size = md.ctx; pl->vocab = md.V; }
        else if (md.d != pl->d || md.ctx != pl->ctx_size) { delete pl; LH_FAIL(ctx, LH_ESHAPE, "lh_pipeline_create: pods differ in shape"); }
        pl->pods.push_back(m);
    }
    pl->past.assign(n_pods, 0);
    // rows per tick: as many as the P-row kernels take (64) unless the caller asks for fewer; 1 = every stream on its own
    const uint32_t cap = 64u;   // (fp32 and block-int8 alike: plan.hip BATCH_ROWS_MAX)
    const uint32_t mr = max_rows ? std::min(max_rows, cap) : cap;
    const uint32_t G = group_count(n_pods, (uint32_t)world, mr);
    pl->groups.resize(G);
    pl->group_of.resize(n_pods); pl->row_of.resize(n_pods);
    for (uint32_t p = 0; p < n_pods; ++p) {
        const uint32_t g = (uint32_t)(((uint64_t)p * G) / n_pods);
        pl->group_of[p] = g;
        pl->row_of[p] = (uint32_t)pl->groups[g].pods.size();
        pl->groups[g].pods.push_back(p);
    }
    for (Group& gr : pl->groups) {
        std::vector<lh_llama*> members;
        for (uint32_t p : gr.pods) members.push_back(pl->pods[p]);
        int rc = lh_batch_create(ctx, members.data(), (uint32_t)members.size(), &gr.batch);
        if (rc) { lh_pipeline_destroy(pl); return rc; }
        if (rank == 0 || rank == world - 1) {
            gr.hist_cap = pl->ctx_size + 1;
            const size_t bytes = (size_t)gr.hist_cap * gr.pods.size() * 4;
            if (hipMalloc((void**)&gr.hist, bytes) != hipSuccess || hipMemsetAsync(gr.hist, 0, bytes, ctx->stream) != hipSuccess) {
                (void)hipGetLastError();
                lh_pipeline_destroy(pl);
                LH_FAIL(ctx, LH_ENOMEM, "lh_pipeline_create: device allocation failed");
            }
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { lh_pipeline_destroy(pl); LH_FAIL(ctx, LH_EHIP, "lh_pipeline_create: stream synchronisation failed"); }
    *out = pl;
    return LH_OK;
}

int lh_pipeline_create(lh_ctx* ctx, lh_comm* comm, lh_llama* const* pods, uint32_t n_pods, lh_pipeline** out) {
    return lh_pipeline_create_grouped(ctx, comm, pods, n_pods, 0, out);
}

void lh_pipeline_destroy(lh_pipeline* pl) {
    if (!pl) return;
    hipSetDevice(pl->ctx->device);
    hipStreamSynchronize(pl->ctx->stream);
    for (hipEvent_t e : pl->ev) hipEventDestroy(e);
    for (Group& gr : pl->groups) {
        if (gr.batch) lh_batch_destroy(gr.batch);
        if (gr.x_in) hipFree(gr.x_in);
        if (gr.x_out) hipFree(gr.x_out);
        if (gr.hist) hipFree(gr.hist);
    }
    delete pl;
}

uint32_t lh_pipeline_groups(const lh_pipeline* pl) { return pl ? (uint32_t)pl->groups.size() : 0; }

static int group_ensure_rows(lh_pipeline* pl, Group& gr, uint32_t rows) {
    if (rows <= gr.rows_cap) return 0;
    lh_ctx* ctx = pl->ctx;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (gr.x_in) LH_HIP(ctx, hipFree(gr.x_in));
    if (gr.x_out) LH_HIP(ctx, hipFree(gr.x_out));
    gr.x_in = gr.x_out = nullptr; gr.rows_cap = 0;
    if (pl->rank != 0) LH_HIP(ctx, hipMalloc((void**)&gr.x_in, (size_t)rows * pl->d * 4));
    if (pl->rank != pl->world - 1) LH_HIP(ctx, hipMalloc((void**)&gr.x_out, (size_t)rows * pl->d * 4));
    gr.rows_cap = rows;   // (the groups' captured ticks notice the new addresses: lh_batch re-captures)
    return 0;
}

static int pipeline_run(lh_pipeline* pl, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps, const lh_sample_params* smp, uint32_t ring_size) {
    if (!pl) return LH_EINVAL;
    lh_ctx* ctx = pl->ctx;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t P = (uint32_t)pl->pods.size(), G = (uint32_t)pl->groups.size(), R = (uint32_t)pl->world, r = (uint32_t)pl->rank;
    const bool first = r == 0, last = r == R - 1, prefill = n_prompt != nullptr;
    const uint32_t units = steps + (prefill ? 1 : 0);
    if (!units) return LH_OK;
    // ---- validation: rank-independent conditions only (every rank takes the same decision, so no rank is left waiting in a
    // receive for a peer that returned), all of it before any state changes.  prompts themselves exist on rank 0 only: their ids
    // are checked there by lh_batch_prompt before rank 0 launches anything of the run.
    if (!prefill && !pl->started) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: no token to continue from (run a prompt first)");
    if (!prefill && (smp != nullptr) != pl->sampling)
        LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: streams whose prompt ran %s the sampler continue with %s", pl->sampling ? "with" : "without", pl->sampling ? "lh_pipeline_run_sample" : "lh_pipeline_run");
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t np = prefill ? n_prompt[p] : 0;
        if (prefill && !np) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u has an empty prompt", p);
        if (np > pl->ctx_size) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u: a prompt of %u tokens exceeds the context window of %u", p, np, pl->ctx_size);
        // past the window every stream swaps context like server.Do (server.go:160-172): an unsharded pipeline inside its lh_batch ticks, a sharded
        // one in the unit where a stream stands at the window's end - the re-fed run travels through the stages in front of that unit's tick
        if (pl->keep >= pl->ctx_size && (uint64_t)(prefill ? 0 : pl->past[p]) + np + steps > pl->ctx_size)
            LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u would leave the context window of %u and KeepCount %u leaves no room to swap", p, pl->ctx_size, pl->keep);
    }
    // what only ONE rank can see (the prompts live on rank 0, and on the last rank of a sampled run): that rank tears the communicator
    // down before it returns, so the others' first receive fails instead of waiting for it
    auto local_fail = [&](const char* what, uint32_t p, uint32_t v) -> int {
        if (pl->comm && R > 1) lh_comm_abort(pl->comm);
        LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u: %s (%u)", p, what, v);
    };
    if (prefill && (first || (smp && last))) {
        if (!prompts) return local_fail("this rank needs the prompts", 0, r);
        for (uint32_t p = 0; p < P; ++p) {
            if (!prompts[p]) return local_fail("no prompt", p, 0);
            for (uint32_t j = 0; j < n_prompt[p]; ++j)
                if (prompts[p][j] >= pl->vocab) return local_fail("token id outside the vocabulary", p, prompts[p][j]);
        }
    }
    int rc;
    for (Group& gr : pl->groups) {
        uint32_t rows = (uint32_t)gr.pods.size();
        if (prefill) { uint32_t t = 0; for (uint32_t p : gr.pods) t += n_prompt[p]; rows = std::max(rows, t); }
        if ((rc = group_ensure_rows(pl, gr, rows))) return rc;
        if (prefill) {
            gr.n_hist = 0;
            // sampler state of the group's rows: ring = the prompt ids (server.go:193-197); NULL = greedy
            std::vector<const uint32_t*> init;
            std::vector<uint32_t> ninit;
            for (uint32_t p : gr.pods) { init.push_back(first && smp ? prompts[p] : nullptr); ninit.push_back(first && smp ? n_prompt[p] : 0); }
            if (smp && last && !first) {
                // the last rank samples: the repeat penalty runs over the lastNTokens ring, which starts as the prompt ids - a sharded sampled run
                // needs the prompts on the last rank too (checked above)
                init.clear(); ninit.clear();
                for (uint32_t p : gr.pods) { init.push_back(prompts[p]); ninit.push_back(n_prompt[p]); }
            }
            if ((rc = lh_batch_set_sampler(gr.batch, smp, ring_size, smp ? init.data() : nullptr, smp ? ninit.data() : nullptr))) return rc;
        }
    }
    if (prefill) { pl->sampling = smp != nullptr; std::fill(pl->past.begin(), pl->past.end(), 0u); }
    if (prefill && first && R > 1) {
        pl->prompt_host.assign(P, std::vector<uint32_t>());
        for (uint32_t p = 0; p < P; ++p) pl->prompt_host[p].assign(prompts[p], prompts[p] + n_prompt[p]);
    }
    // A decode unit of a SHARDED pipeline in which streams of the group stand at the window's end is a swap unit: every such stream's re-fed run
    // ((ctx - keep) / 2 tokens, the pending one last) is evaluated as one Eval at position keep on every rank in turn - its residual rows travel
    // in FRONT of the tick's rows in the same exchange - and the tick then takes the pending token behind it.  Every rank derives the same swap
    // set from its mirror of the positions; only rank 0 needs the tokens.
    const uint32_t swap_n = pl->keep < pl->ctx_size ? (pl->ctx_size - pl->keep) / 2 : 0;
    auto swap_rows = [&](uint32_t g) -> uint32_t {   // rows of the re-fed runs of group g's next decode unit, from the positions as they are NOW
        uint32_t t = 0;
        if (R > 1) for (uint32_t p : pl->groups[g].pods) if (pl->past[p] >= pl->ctx_size) t += swap_n;
        return t;
    };
    std::vector<uint32_t> unit_rows(G, 0);           // rows the group's CURRENT unit sends / receives (set by its stage, or by recv_rows on a rank that only receives)
    auto rows_of = [&](uint32_t g, uint32_t u) -> uint32_t {
        const Group& gr = pl->groups[g];
        if (!(prefill && u == 0)) return (uint32_t)gr.pods.size();
        uint32_t t = 0;
        for (uint32_t p : gr.pods) t += n_prompt[p];
        return t;
    };
    // ids of a finished unit: a copy into the group's history (what lh_pipeline_tokens reads)
    auto record = [&](Group& gr, const uint32_t* ids_dev) -> int {
        if (gr.n_hist >= gr.hist_cap) {   // streams that swap context outlive the window: the history grows
            const uint32_t cap2 = gr.hist_cap * 2;
            uint32_t* h2 = nullptr;
            LH_HIP(ctx, hipMalloc((void**)&h2, (size_t)cap2 * gr.pods.size() * 4));
            LH_HIP(ctx, hipMemcpyAsync(h2, gr.hist, (size_t)gr.n_hist * gr.pods.size() * 4, hipMemcpyDeviceToDevice, ctx->stream));
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            LH_HIP(ctx, hipFree(gr.hist));
            gr.hist = h2; gr.hist_cap = cap2;
        }
        LH_HIP(ctx, hipMemcpyAsync(gr.hist + (size_t)gr.n_hist * gr.pods.size(), ids_dev, gr.pods.size() * 4, hipMemcpyDeviceToDevice, ctx->stream));
        gr.n_hist++;
        return 0;
    };
    auto stage = [&](uint32_t g, uint32_t u) -> int {
        Group& gr = pl->groups[g];
        int rc2;
        if (prefill && u == 0) {
            std::vector<const uint32_t*> pr;
            std::vector<uint32_t> np;
            for (uint32_t p : gr.pods) { pr.push_back(first ? prompts[p] : nullptr); np.push_back(n_prompt[p]); pl->past[p] = n_prompt[p]; }
            unit_rows[g] = rows_of(g, u);
            rc2 = lh_batch_prompt(gr.batch, first ? pr.data() : nullptr, np.data(), first ? nullptr : gr.x_in, last ? nullptr : gr.x_out);
        } else {
            const uint32_t B = (uint32_t)gr.pods.size(), extra = swap_rows(g);
            unit_rows[g] = B + extra;
            if (extra) {
                if ((rc2 = group_ensure_rows(pl, gr, B + extra))) return rc2;
                // rank 0: the ring of every swapping stream = its prompt + the ids received so far (Group::hist: [unit][row]); the pending id is the newest
                std::vector<uint32_t> ids;
                if (first) {
                    ids.resize((size_t)gr.n_hist * B);
                    LH_HIP(ctx, hipMemcpyAsync(ids.data(), gr.hist, ids.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
                    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
                }
                std::vector<uint32_t> newpos(B), refeed;
                uint32_t off = 0;
                for (uint32_t i = 0; i < B; ++i) {
                    const uint32_t p = gr.pods[i];
                    newpos[i] = pl->past[p];
                    if (pl->past[p] < pl->ctx_size) continue;
                    if (first) {
                        std::vector<uint32_t> ring = pl->prompt_host[p];
                        for (uint32_t k = 0; k < gr.n_hist; ++k) ring.push_back(ids[(size_t)k * B + i]);
                        if (ring.size() < swap_n) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u: the window's tokens are not known on rank 0", p);
                        refeed.assign(ring.end() - swap_n, ring.end());
                    }
                    if (swap_n && (rc2 = lh_llama_stage(pl->pods[p], first ? refeed.data() : nullptr, nullptr, first ? nullptr : gr.x_in + (size_t)off * pl->d,
                                                        last ? nullptr : gr.x_out + (size_t)off * pl->d, swap_n, pl->keep, nullptr, nullptr)))
                        return rc2;
                    off += swap_n;
                    newpos[i] = pl->keep + swap_n;
                    pl->past[p] = newpos[i];
                }
                if ((rc2 = lh_batch_set(gr.batch, nullptr, newpos.data()))) return rc2;   // token ids stay what the last exchange delivered
            }
            rc2 = lh_batch_stage(gr.batch, first ? nullptr : gr.x_in + (size_t)extra * pl->d, last ? nullptr : gr.x_out + (size_t)extra * pl->d, nullptr, nullptr);
            for (uint32_t p : gr.pods) pl->past[p] += 1;
        }
        if (rc2) return rc2;
        if (last) return record(gr, lh_batch_ids_dev(gr.batch));
        return 0;
    };
    auto exchange = [&](int32_t s, int32_t u, int32_t rs, int32_t ru) -> int {
        if (R == 1 && !pl->comm) return 0;   // unsharded: a whole-model batch feeds the produced ids back by itself
        const void* sb = nullptr; uint64_t sbytes = 0;
        void* rb = nullptr; uint64_t rbytes = 0;
        if (s >= 0) {
            Group& gr = pl->groups[s];
            if (last) { sb = lh_batch_ids_dev(gr.batch); sbytes = gr.pods.size() * 4; }
            else { sb = gr.x_out; sbytes = (uint64_t)unit_rows[s] * pl->d * 4; }
        }
        if (rs >= 0) {
            Group& gr = pl->groups[rs];
            if (first) { rb = lh_batch_tokens_dev(gr.batch); rbytes = gr.pods.size() * 4; }
            else {
                // what the predecessor sends for unit ru of group rs: this rank has not run that unit yet, so its position mirror says what the unit is
                const uint32_t rows = (prefill && ru == 0) ? rows_of((uint32_t)rs, 0) : (uint32_t)gr.pods.size() + swap_rows((uint32_t)rs);
                int rc3 = group_ensure_rows(pl, gr, rows);
                if (rc3) return rc3;
                rb = gr.x_in; rbytes = (uint64_t)rows * pl->d * 4;
            }
        }
        int rc2 = lh_comm_exchange(pl->comm, sb, sbytes, (int)((r + 1) % R), rb, rbytes, (int)((r + R - 1) % R));
        if (rc2) return rc2;
        if (rs >= 0 && first && !last) return record(pl->groups[rs], lh_batch_tokens_dev(pl->groups[rs].batch));
        return 0;
    };
    // profiling: the same ticks with three events each on the compute stream (read back behind the run's final synchronisation)
    pl->ev_used = 0;
    auto mark = [&]() -> int {
        if (pl->ev_used == pl->ev.size()) { hipEvent_t e; LH_HIP(ctx, hipEventCreate(&e)); pl->ev.push_back(e); }
        LH_HIP(ctx, hipEventRecord(pl->ev[pl->ev_used++], ctx->stream));
        return 0;
    };
    auto stage_p = [&](uint32_t g, uint32_t u) -> int {
        int rc2;
        if (pl->profiling && (rc2 = mark())) return rc2;
        if ((rc2 = stage(g, u))) return rc2;
        return pl->profiling ? mark() : 0;
    };
    auto exchange_p = [&](int32_t s, int32_t u, int32_t rs, int32_t ru) -> int {
        int rc2;
        if (pl->profiling && s < 0) { if ((rc2 = mark()) || (rc2 = mark())) return rc2; }   // a receive-only tick: an empty stage interval
        if ((rc2 = exchange(s, u, rs, ru))) return rc2;
        return pl->profiling ? mark() : 0;
    };
    rc = run_ticks(r, R, G, units, stage_p, exchange_p);
    if (rc) {
        // a rank that fails mid-run must not leave its peers waiting in a receive: tear the communicator down (RCCL: abort; the peers'
        // pending operations then fail instead of blocking).  The pipeline is unusable afterwards.
        if (pl->comm) lh_comm_abort(pl->comm);
        return rc;
    }
    if (prefill) pl->started = true;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (pl->profiling) {
        for (uint32_t i = 0; i + 2 < pl->ev_used; i += 3) {
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, pl->ev[i], pl->ev[i + 1]) == hipSuccess && hipEventElapsedTime(&b, pl->ev[i + 1], pl->ev[i + 2]) == hipSuccess) {
                pl->stats.ticks += 1; pl->stats.stage_ms += a; pl->stats.exchange_ms += b;
            }
        }
    }
    return LH_OK;
}

int lh_pipeline_set_keep(lh_pipeline* pl, uint32_t keep) {
    if (!pl) return LH_EINVAL;
    pl->keep = keep;
    for (lh_llama* m : pl->pods) lh_llama_set_keep(m, keep);
    return LH_OK;
}
int lh_pipeline_profile(lh_pipeline* pl, int on) {
    if (!pl) return LH_EINVAL;
    pl->profiling = on != 0;
    if (on) pl->stats = lh_pipeline_stats{};
    return LH_OK;
}
int lh_pipeline_stats_read(lh_pipeline* pl, lh_pipeline_stats* out) {
    if (!pl || !out) return LH_EINVAL;
    *out = pl->stats;
    return LH_OK;
}
// `iters` ring shifts of `bytes` (every rank sends to its successor and receives from its predecessor in one grouped call, as a tick's
// exchange does), timed with HIP events on the compute stream: microseconds per shift.  World of one: a self send/recv.
int lh_pipeline_hop_probe(lh_pipeline* pl, uint32_t bytes, uint32_t iters, float* us_per_hop) {
    if (!pl || !us_per_hop || !bytes || !iters) return LH_EINVAL;
    lh_ctx* ctx = pl->ctx;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!pl->comm) { *us_per_hop = 0.f; return LH_OK; }
    void *sb = nullptr, *rb = nullptr;
    LH_HIP(ctx, hipMalloc(&sb, bytes));
    LH_HIP(ctx, hipMalloc(&rb, bytes));
    LH_HIP(ctx, hipMemsetAsync(sb, 0, bytes, ctx->stream));
    const int R = pl->world, r = pl->rank;
    int rc = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t i = 0; i < 8 && !rc; ++i) rc = lh_comm_exchange(pl->comm, sb, bytes, (r + 1) % R, rb, bytes, (r + R - 1) % R);   // warm-up: connections, buffers
    if (!rc) {
        hipEventRecord(e0, ctx->stream);
        for (uint32_t i = 0; i < iters && !rc; ++i) rc = lh_comm_exchange(pl->comm, sb, bytes, (r + 1) % R, rb, bytes, (r + R - 1) % R);
        hipEventRecord(e1, ctx->stream);
    }
    hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    if (!rc) hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(sb); hipFree(rb);
    *us_per_hop = ms * 1e3f / (float)iters;
    return rc;
}

int lh_pipeline_run(lh_pipeline* pl, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps) {
    return pipeline_run(pl, prompts, n_prompt, steps, nullptr, 0);
}

int lh_pipeline_run_sample(lh_pipeline* pl, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps, const lh_sample_params* sp, uint32_t ring_size) {
    if (!pl) return LH_EINVAL;
    if (!sp) LH_FAIL(pl->ctx, LH_EINVAL, "lh_pipeline_run_sample: null sampler parameters");
    return pipeline_run(pl, prompts, n_prompt, steps, sp, ring_size);
}

int lh_pipeline_tokens(lh_pipeline* pl, uint32_t pod, uint32_t* out, uint32_t cap) {
    if (!pl || pod >= pl->pods.size()) return LH_EINVAL;
    lh_ctx* ctx = pl->ctx;
    Group& gr = pl->groups[pl->group_of[pod]];
    if (!gr.hist) return 0;  // a middle rank sees no ids
    const uint32_t n = gr.n_hist, m = std::min(n, cap), B = (uint32_t)gr.pods.size();
    if (m && out) {
        LH_HIP(ctx, hipSetDevice(ctx->device));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        LH_HIP(ctx, hipMemcpy2D(out, 4, gr.hist + pl->row_of[pod], (size_t)B * 4, 4, m, hipMemcpyDeviceToHost));
    }
    return (int)n;
}

}  // extern "C"
