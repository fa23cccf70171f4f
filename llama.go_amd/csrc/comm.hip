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
    // host staging of the hooks transport (pinned)
    char *send_host = nullptr, *recv_host = nullptr;
    uint64_t send_cap = 0, recv_cap = 0;
};

struct PodState {
    lh_llama* m = nullptr;
    float *x_in = nullptr, *x_out = nullptr;  // residual stream received / produced, rows_cap x embd
    uint32_t rows_cap = 0;
    uint32_t* recv_ids = nullptr;  // rank 0: ids received from the last rank, in order
    uint32_t* prod_ids = nullptr;  // last rank: ids produced, in order
    uint32_t n_recv = 0, n_prod = 0, ids_cap = 0;
    uint32_t past = 0;             // position of the next unit
    uint32_t pending_rows = 0;     // rows of the unit in flight (bookkeeping of the current run)
};

struct lh_pipeline {
    lh_ctx* ctx = nullptr;
    lh_comm* comm = nullptr;
    int rank = 0, world = 1;
    std::vector<PodState> pods;
    uint32_t d = 0, ctx_size = 0;
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
int lh_pipeline_create(lh_ctx* ctx, lh_comm* comm, lh_llama* const* pods, uint32_t n_pods, lh_pipeline** out) {
    if (!ctx || !pods || !n_pods || !out) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: NULL argument");
    *out = nullptr;
    if (comm && comm->ctx != ctx) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: the communicator belongs to another context (one stream must order compute and p2p)");
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    auto pl = std::make_unique<lh_pipeline>();
    pl->ctx = ctx; pl->comm = comm; pl->rank = rank; pl->world = world;
    for (uint32_t i = 0; i < n_pods; ++i) {
        lh_llama* m = pods[i];
        if (!m || m->ctx != ctx) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: pod %u lives on another context", i);
        const ModelDesc& md = m->plan->md;
        if (md.first_stage() != (rank == 0) || md.last_stage() != (rank == world - 1))
            LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_create: pod %u holds layers [%u,%u) of %u, which is not rank %d of %d in a contiguous layer shard", i, md.layer0, md.layer1, md.L, rank, world);
        if (i == 0) { pl->d = md.d; pl->ctx_size = md.ctx; }
        else if (md.d != pl->d || md.ctx != pl->ctx_size) LH_FAIL(ctx, LH_ESHAPE, "lh_pipeline_create: pods differ in shape");
        PodState ps;
        ps.m = m;
        pl->pods.push_back(ps);
    }
    for (PodState& ps : pl->pods) {
        ps.ids_cap = pl->ctx_size + 1;
        if (rank == 0) { LH_HIP(ctx, hipMalloc((void**)&ps.recv_ids, (size_t)ps.ids_cap * 4)); LH_HIP(ctx, hipMemsetAsync(ps.recv_ids, 0, (size_t)ps.ids_cap * 4, ctx->stream)); }
        if (rank == world - 1) { LH_HIP(ctx, hipMalloc((void**)&ps.prod_ids, (size_t)ps.ids_cap * 4)); LH_HIP(ctx, hipMemsetAsync(ps.prod_ids, 0, (size_t)ps.ids_cap * 4, ctx->stream)); }
    }
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = pl.release();
    return LH_OK;
}

void lh_pipeline_destroy(lh_pipeline* pl) {
    if (!pl) return;
    hipSetDevice(pl->ctx->device);
    hipStreamSynchronize(pl->ctx->stream);
    for (PodState& ps : pl->pods) {
        if (ps.x_in) hipFree(ps.x_in);
        if (ps.x_out) hipFree(ps.x_out);
        if (ps.recv_ids) hipFree(ps.recv_ids);
        if (ps.prod_ids) hipFree(ps.prod_ids);
    }
    delete pl;
}

static int pod_ensure_rows(lh_pipeline* pl, PodState& ps, uint32_t rows) {
    if (rows <= ps.rows_cap) return 0;
    lh_ctx* ctx = pl->ctx;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ps.x_in) LH_HIP(ctx, hipFree(ps.x_in));
    if (ps.x_out) LH_HIP(ctx, hipFree(ps.x_out));
    ps.x_in = ps.x_out = nullptr; ps.rows_cap = 0;
    if (pl->rank != 0) LH_HIP(ctx, hipMalloc((void**)&ps.x_in, (size_t)rows * pl->d * 4));
    if (pl->rank != pl->world - 1) LH_HIP(ctx, hipMalloc((void**)&ps.x_out, (size_t)rows * pl->d * 4));
    ps.rows_cap = rows;
    return 0;
}

int lh_pipeline_run(lh_pipeline* pl, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps) {
    if (!pl) return LH_EINVAL;
    lh_ctx* ctx = pl->ctx;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t P = (uint32_t)pl->pods.size(), R = (uint32_t)pl->world, r = (uint32_t)pl->rank;
    const bool first = r == 0, last = r == R - 1, prefill = n_prompt != nullptr;
    const uint32_t units = steps + (prefill ? 1 : 0);
    if (!units) return LH_OK;
    if (prefill && first && !prompts) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: rank 0 needs the prompts");
    int rc;
    for (uint32_t p = 0; p < P; ++p) {
        PodState& ps = pl->pods[p];
        const uint32_t np = prefill ? n_prompt[p] : 0;
        if (prefill) {
            if (!np) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u has an empty prompt", p);
            if (first && !prompts[p]) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u has no prompt", p);
            ps.past = 0; ps.n_recv = 0; ps.n_prod = 0;
        } else if (first && ps.n_recv == 0) {
            LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u has no token to continue from (run a prompt first)", p);
        }
        if ((uint64_t)ps.past + np + steps > pl->ctx_size) LH_FAIL(ctx, LH_EINVAL, "lh_pipeline_run: stream %u would leave the context window of %u", p, pl->ctx_size);
        if ((rc = pod_ensure_rows(pl, ps, std::max(np, 1u)))) return rc;
    }
    auto rows_of = [&](uint32_t p, uint32_t u) -> uint32_t { return (prefill && u == 0) ? n_prompt[p] : 1u; };
    auto stage = [&](uint32_t p, uint32_t u) -> int {
        PodState& ps = pl->pods[p];
        const uint32_t n = rows_of(p, u);
        const uint32_t* tok_host = nullptr;
        const uint32_t* tok_dev = nullptr;
        if (first) {
            if (n > 1 || (prefill && u == 0)) tok_host = prompts[p];
            else tok_dev = ps.recv_ids + (ps.n_recv - 1);  // the id the last rank produced for the previous unit
        }
        uint32_t* amax = last ? ps.prod_ids + ps.n_prod : nullptr;
        int rc2 = lh_llama_stage(ps.m, tok_host, tok_dev, first ? nullptr : ps.x_in, last ? nullptr : ps.x_out, n, ps.past, nullptr, amax);
        if (rc2) return rc2;
        if (last) ps.n_prod++;
        ps.past += n;
        ps.pending_rows = n;
        return 0;
    };
    auto exchange = [&](int32_t s, int32_t u, int32_t rs, int32_t ru) -> int {
        const void* sb = nullptr; uint64_t sbytes = 0;
        void* rb = nullptr; uint64_t rbytes = 0;
        if (s >= 0) {
            PodState& ps = pl->pods[s];
            if (last) { sb = ps.prod_ids + (ps.n_prod - 1); sbytes = 4; }
            else { sb = ps.x_out; sbytes = (uint64_t)rows_of((uint32_t)s, (uint32_t)u) * pl->d * 4; }
        }
        if (rs >= 0) {
            PodState& ps = pl->pods[rs];
            if (first) { rb = ps.recv_ids + ps.n_recv; rbytes = 4; ps.n_recv++; }
            else { rb = ps.x_in; rbytes = (uint64_t)rows_of((uint32_t)rs, (uint32_t)ru) * pl->d * 4; }
        }
        if (R == 1 && !pl->comm) {  // unsharded: the produced id is the received id
            if (sb && rb) LH_HIP(ctx, hipMemcpyAsync(rb, sb, 4, hipMemcpyDeviceToDevice, ctx->stream));
            return 0;
        }
        return lh_comm_exchange(pl->comm, sb, sbytes, (int)((r + 1) % R), rb, rbytes, (int)((r + R - 1) % R));
    };
    if ((rc = run_ticks(r, R, P, units, stage, exchange))) return rc;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_pipeline_tokens(lh_pipeline* pl, uint32_t pod, uint32_t* out, uint32_t cap) {
    if (!pl || pod >= pl->pods.size()) return LH_EINVAL;
    lh_ctx* ctx = pl->ctx;
    PodState& ps = pl->pods[pod];
    const uint32_t* src = pl->rank == 0 ? ps.recv_ids : ps.prod_ids;
    const uint32_t n = pl->rank == 0 ? ps.n_recv : ps.n_prod;
    if (!src) return 0;  // a middle rank sees no ids
    const uint32_t m = std::min(n, cap);
    if (m && out) {
        LH_HIP(ctx, hipSetDevice(ctx->device));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        LH_HIP(ctx, hipMemcpy(out, src, (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    return (int)n;
}

}  // extern "C"
