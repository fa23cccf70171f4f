// csrc/abi.hip — contexts, persistent buffers, synthetic-weight generator, RoPE table.
// C-ABI entry points declared in include/llamahip.h.
#include "common.h"
#include "kernels_q8_pack.h"
#include <stdarg.h>
#include <math.h>
#include <algorithm>

namespace lh {

static thread_local std::string g_thread_err;

void set_error(lh_ctx* ctx, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_thread_err = buf;
    if (ctx) ctx->err = buf;
}

static std::mutex g_dev_mu;
static std::unordered_map<int, DeviceState*> g_devs;

DeviceState* device_state(int device) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_devs.find(device);
    if (it != g_devs.end()) return it->second;
    DeviceState* ds = new DeviceState();
    ds->device = device;
    if (hipGetDeviceProperties(&ds->prop, device) != hipSuccess) { delete ds; return nullptr; }
    ds->num_cu = ds->prop.multiProcessorCount;
    g_devs[device] = ds;
    return ds;
}

Buffer* find_buffer(DeviceState* ds, lh_buf id) {
    std::lock_guard<std::mutex> lk(ds->mu);
    auto it = ds->bufs.find(id);
    return it == ds->bufs.end() ? nullptr : it->second.get();
}

Buffer* find_buffer_fast(lh_ctx* ctx, lh_buf id) {
    DeviceState* ds = ctx->ds;
    const uint64_t g = ds->bufs_gen.load(std::memory_order_acquire);
    if (ctx->buf_snap_gen != g) {
        std::lock_guard<std::mutex> lk(ds->mu);
        ctx->buf_snap.assign((size_t)ds->next_id, nullptr);
        for (auto& kv : ds->bufs) if (kv.first < ds->next_id) ctx->buf_snap[(size_t)kv.first] = kv.second.get();
        ctx->buf_snap_gen = ds->bufs_gen.load(std::memory_order_relaxed);
    }
    return id < ctx->buf_snap.size() ? ctx->buf_snap[(size_t)id] : nullptr;
}

int ensure_arena(lh_ctx* ctx, uint64_t bytes) {
    if (bytes <= ctx->arena_bytes) return 0;
    uint64_t want = bytes + (bytes >> 2) + (1u << 20);
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) LH_HIP(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
    LH_HIP(ctx, hipMalloc((void**)&ctx->arena, want));
    ctx->arena_bytes = want;
    return 0;
}

int ensure_staging(lh_ctx* ctx, uint64_t bytes) {
    if (bytes <= ctx->staging_bytes) return 0;
    uint64_t want = bytes * 2 + 4096;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->staging) LH_HIP(ctx, hipHostFree(ctx->staging));
    ctx->staging = nullptr;
    ctx->staging_bytes = 0;
    LH_HIP(ctx, hipHostMalloc((void**)&ctx->staging, want, hipHostMallocDefault));
    ctx->staging_bytes = want;
    return 0;
}

// RoPE table, computed on the host in f64 with the reference's expressions (ml.go:2307-2310):
//   theta = pow(10000, -i0/dims);  cos(p*theta), sin(p*theta)     for i0 = 0, 2, .., dims-2.
int ensure_rope_table(lh_ctx* ctx, uint32_t positions, uint32_t dims, const double2** table) {
    DeviceState* ds = ctx->ds;
    if (!dims || dims % 2) LH_FAIL(ctx, LH_ESHAPE, "rope: rotation width %u must be even and non-zero", dims);
    std::lock_guard<std::mutex> lk(ds->mu);
    DeviceState::RopeTable& rt = ds->rope[dims];
    if (rt.dev && rt.positions >= positions) { *table = rt.dev; return 0; }
    uint32_t npos = positions < 256 ? 256 : positions;
    if (rt.positions * 2 > npos) npos = rt.positions * 2;
    const uint32_t half = dims / 2;
    std::vector<double2> h((size_t)npos * half);
    for (uint32_t i = 0; i < half; ++i) {
        const int i0 = (int)(2 * i);
        const double theta = pow(10000.0, (double)(-i0) / (double)dims);
        for (uint32_t p = 0; p < npos; ++p) {
            h[(size_t)p * half + i].x = cos((double)p * theta);
            h[(size_t)p * half + i].y = sin((double)p * theta);
        }
    }
    double2* dev = nullptr;
    LH_HIP(ctx, hipMalloc((void**)&dev, h.size() * sizeof(double2)));
    LH_HIP(ctx, hipMemcpy(dev, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice));
    rt.dev = dev;  // the previous (smaller) table stays allocated: see DeviceState::rope
    rt.positions = npos;
    *table = dev;
    return 0;
}

// ---- synthetic weights (DESIGN.md "Synthetic model") -------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_fill_synth(float* __restrict__ dst, uint64_t n, uint64_t key, float scale, float offset) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const uint64_t h = mix64(key + i);
        const int k = (int)(h >> 40);
        const float v = (float)(2 * k - (1 << 24)) * (1.0f / 16777216.0f);  // exact
        const float sv = __fmul_rn(scale, v);                                // one rounding, never fused
        dst[i] = __fadd_rn(offset, sv);
    }
}

}  // namespace lh

using namespace lh;

// Read-only HBM stream (lh_hbm_read_probe): every workgroup walks its own contiguous share with 16-byte non-temporal loads, eight
// independent loads per lane in flight - the access pattern of the decode weight stream without any arithmetic behind it: what this
// box's memory system delivers to a kernel that does nothing else (the yardstick next to the nominal 8 TB/s; SURVEY 8d "vs measured").
typedef float probe_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_read_probe(const probe_f4* __restrict__ src, uint64_t n16, float* __restrict__ sink) {
    const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x, b0 = (uint64_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    float acc = 0.f;
    constexpr int U = 8;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += (uint64_t)1024 * U) {
        probe_f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t j = i + (uint64_t)u * 1024;
            v[u] = __builtin_nontemporal_load(src + (j < b1 ? j : b0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 12345.678f) sink[blockIdx.x] = acc;   // keeps the loads alive; practically never true
}

namespace lh {
std::atomic<int> g_route_log_on{0};
static std::mutex g_route_mu;
static std::vector<std::string> g_route_names;
// "(k_stream_mm2<MAXT, NCT, KC2>)" / "lh::k_embed" / "k_gemm_b9" -> the family name
void route_note(const char* kernel_expr) {
    std::string n(kernel_expr);
    size_t b = 0;
    while (b < n.size() && (n[b] == '(' || n[b] == ' ')) ++b;
    size_t e = b;
    while (e < n.size() && (isalnum((unsigned char)n[e]) || n[e] == '_' || n[e] == ':')) ++e;
    n = n.substr(b, e - b);
    const size_t c = n.rfind("::");
    if (c != std::string::npos) n = n.substr(c + 2);
    std::lock_guard<std::mutex> g(g_route_mu);
    for (const auto& s : g_route_names) if (s == n) return;
    g_route_names.push_back(n);
}
}  // namespace lh

extern "C" {

int lh_hbm_read_probe(lh_ctx* ctx, uint64_t bytes, uint32_t repeats, float* gbps) {
    if (!ctx || !gbps || !repeats || bytes < (1u << 20)) return LH_EINVAL;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    float* buf = nullptr;
    float* sink = nullptr;
    LH_HIP(ctx, hipMalloc((void**)&buf, bytes));
    if (hipMalloc((void**)&sink, 4096 * 4) != hipSuccess) { hipFree(buf); LH_FAIL(ctx, LH_ENOMEM, "lh_hbm_read_probe: allocation failed"); }
    hipMemsetAsync(buf, 0, bytes, ctx->stream);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t grid = (uint32_t)ctx->ds->num_cu;
    LH_LAUNCH(k_read_probe, dim3(grid), dim3(1024), 0, ctx->stream, (const probe_f4*)buf, bytes / 16, sink);
    hipEventRecord(e0, ctx->stream);
    for (uint32_t r = 0; r < repeats; ++r) LH_LAUNCH(k_read_probe, dim3(grid), dim3(1024), 0, ctx->stream, (const probe_f4*)buf, bytes / 16, sink);
    hipEventRecord(e1, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(buf);
    hipFree(sink);
    if (e != hipSuccess || ms <= 0.f) LH_FAIL(ctx, LH_EHIP, "lh_hbm_read_probe: %s", hipGetErrorString(e));
    *gbps = (float)((double)(bytes / 16 * 16) * repeats / (ms * 1e-3) / 1e9);
    return LH_OK;
}

int lh_abi_version(void) { return LH_ABI_VERSION; }

int lh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* lh_last_error(lh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_thread_err.c_str(); }

int lh_ctx_create(int device, void* stream, lh_ctx** out) {
    if (!out) LH_FAIL(nullptr, LH_EINVAL, "lh_ctx_create: out is NULL");
    *out = nullptr;
    int n = lh_device_count();
    if (n <= 0) LH_FAIL(nullptr, LH_ENODEVICE, "lh_ctx_create: no HIP device visible (the HIP path has no CPU fallback)");
    if (device < 0 || device >= n) LH_FAIL(nullptr, LH_EINVAL, "lh_ctx_create: device %d out of range (0..%d)", device, n - 1);
    LH_HIP(nullptr, hipSetDevice(device));
    DeviceState* ds = device_state(device);
    if (!ds) LH_FAIL(nullptr, LH_EHIP, "lh_ctx_create: cannot query device %d", device);
    lh_ctx* c = new lh_ctx();
    c->device = device;
    c->ds = ds;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; LH_FAIL(nullptr, LH_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    *out = c;
    return LH_OK;
}

int lh_ctx_sync(lh_ctx* ctx) {
    if (!ctx) return LH_EINVAL;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

void* lh_ctx_stream(lh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int lh_tensor_register(lh_ctx* ctx, uint64_t key, int dtype, const uint32_t ne[4], int persistent, const void* host, lh_buf* out) {
    if (!ctx || !ne || !out) LH_FAIL(ctx, LH_EINVAL, "lh_tensor_register: NULL argument");
    (void)persistent;
    if (dtype != 0 && dtype != 7) LH_FAIL(ctx, LH_EUNSUPPORTED, "lh_tensor_register: dtype %d not supported (f32 and block-int8 only; the reference loader accepts f32/f16, llama.go:956-959)", dtype);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (dtype == 7) {
        if (ne[0] % 32 || ne[2] != 1 || ne[3] != 1 || !host) LH_FAIL(ctx, LH_ESHAPE, "lh_tensor_register: block-int8 needs a 2-D matrix with columns %% 32 == 0 and host blocks");
        const uint64_t nblocks = (uint64_t)ne[0] * ne[1] / 32;
        auto b = std::make_unique<Buffer>();
        b->nfloats = (uint64_t)ne[0] * ne[1]; b->dtype = 7; b->rows = ne[1]; b->cols = ne[0]; b->key = key; b->device = ctx->device;
        const uint64_t qbytes = (b->nfloats + 255) & ~(uint64_t)255;
        b->bytes = qbytes + nblocks * 4;
        unsigned int* raw = nullptr;
        LH_HIP(ctx, hipMalloc((void**)&b->dev, b->bytes));
        b->scales = (float*)((char*)b->dev + qbytes);
        LH_HIP(ctx, hipMalloc((void**)&raw, nblocks * 36));
        LH_HIP(ctx, hipMemcpy(raw, host, nblocks * 36, hipMemcpyHostToDevice));
        LH_LAUNCH(k_q8_deinterleave, dim3((unsigned)std::min<uint64_t>((nblocks + 255) / 256, 65535)), dim3(256), 0, ctx->stream, (const unsigned int*)raw,
                           (signed char*)b->dev, b->scales, nblocks);
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        LH_HIP(ctx, hipFree(raw));
        std::lock_guard<std::mutex> lk(ctx->ds->mu);
        lh_buf id = ctx->ds->next_id++;
        if (key) ctx->ds->by_key[key] = id;
        ctx->ds->bufs[id] = std::move(b);
        ctx->ds->bufs_gen.fetch_add(1, std::memory_order_release);
        *out = id;
        return LH_OK;
    }
    DeviceState* ds = ctx->ds;
    if (key) {
        std::lock_guard<std::mutex> lk(ds->mu);
        auto it = ds->by_key.find(key);
        if (it != ds->by_key.end()) { *out = it->second; return LH_OK; }
    }
    uint64_t n = (uint64_t)ne[0] * ne[1] * ne[2] * ne[3];
    if (n == 0) LH_FAIL(ctx, LH_EINVAL, "lh_tensor_register: empty tensor");
    auto b = std::make_unique<Buffer>();
    b->nfloats = n;
    b->bytes = n * 4;
    b->dtype = dtype;
    b->key = key;
    b->device = ctx->device;
    hipError_t e = hipMalloc((void**)&b->dev, b->bytes);
    if (e != hipSuccess) LH_FAIL(ctx, LH_ENOMEM, "lh_tensor_register: hipMalloc(%llu bytes): %s", (unsigned long long)b->bytes, hipGetErrorString(e));
    if (host) LH_HIP(ctx, hipMemcpyAsync(b->dev, host, b->bytes, hipMemcpyHostToDevice, ctx->stream));
    else LH_HIP(ctx, hipMemsetAsync(b->dev, 0, b->bytes, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ds->mu);
    lh_buf id = ds->next_id++;
    if (key) ds->by_key[key] = id;
    ds->bufs[id] = std::move(b);
    ds->bufs_gen.fetch_add(1, std::memory_order_release);
    *out = id;
    return LH_OK;
}

static int get_buf(lh_ctx* ctx, lh_buf buf, uint64_t off, uint64_t n, Buffer** out, const char* who) {
    if (!ctx) return LH_EINVAL;
    Buffer* b = find_buffer(ctx->ds, buf);
    if (!b) LH_FAIL(ctx, LH_EINVAL, "%s: unknown buffer %llu", who, (unsigned long long)buf);
    if (off > b->nfloats || n > b->nfloats - off) LH_FAIL(ctx, LH_EINVAL, "%s: range [%llu,+%llu) outside buffer of %llu floats", who, (unsigned long long)off, (unsigned long long)n, (unsigned long long)b->nfloats);
    *out = b;
    return LH_OK;
}

int lh_buf_upload(lh_ctx* ctx, lh_buf buf, uint64_t off, const float* host, uint64_t n) {
    Buffer* b;
    int rc = get_buf(ctx, buf, off, n, &b, "lh_buf_upload");
    if (rc) return rc;
    if (!host) LH_FAIL(ctx, LH_EINVAL, "lh_buf_upload: host is NULL");
    if (b->dtype != 0) LH_FAIL(ctx, LH_EINVAL, "lh_buf_upload: not an f32 buffer");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LH_HIP(ctx, hipMemcpyAsync(b->dev + off, host, n * 4, hipMemcpyHostToDevice, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_buf_read(lh_ctx* ctx, lh_buf buf, uint64_t off, float* dst, uint64_t n) {
    Buffer* b;
    int rc = get_buf(ctx, buf, off, n, &b, "lh_buf_read");
    if (rc) return rc;
    if (!dst) LH_FAIL(ctx, LH_EINVAL, "lh_buf_read: dst is NULL");
    if (b->dtype == 7) return lh_buf_read_q8(ctx, buf, off, dst, n);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LH_HIP(ctx, hipMemcpyAsync(dst, b->dev + off, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_buf_fill_synth(lh_ctx* ctx, lh_buf buf, uint64_t off, uint64_t n, uint64_t seed, uint32_t tensor_id, float scale, float offset) {
    Buffer* b;
    int rc = get_buf(ctx, buf, off, n, &b, "lh_buf_fill_synth");
    if (rc) return rc;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t key = mix64(seed ^ ((uint64_t)tensor_id * 0xD6E8FEB86659FD93ull));
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    LH_LAUNCH(k_fill_synth, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, b->dev + off, n, key, scale, offset);
    LH_HIP(ctx, hipGetLastError());
    return LH_OK;
}

int lh_buf_quantize_q8(lh_ctx* ctx, lh_buf src, uint32_t rows, uint32_t cols, lh_buf* out) {
    Buffer* sb;
    if (!out) return LH_EINVAL;
    int rc = get_buf(ctx, src, 0, (uint64_t)rows * cols, &sb, "lh_buf_quantize_q8");
    if (rc) return rc;
    if (sb->dtype != 0 || cols % 32) LH_FAIL(ctx, LH_ESHAPE, "lh_buf_quantize_q8: needs an f32 matrix with columns %% 32 == 0");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = (uint64_t)rows * cols, nblocks = n / 32;
    auto b = std::make_unique<Buffer>();
    b->nfloats = n; b->dtype = 7; b->rows = rows; b->cols = cols; b->device = ctx->device;
    const uint64_t qbytes = (n + 255) & ~(uint64_t)255;
    b->bytes = qbytes + nblocks * 4;
    LH_HIP(ctx, hipMalloc((void**)&b->dev, b->bytes));
    b->scales = (float*)((char*)b->dev + qbytes);
    LH_LAUNCH(k_quantize_q8, dim3((unsigned)std::min<uint64_t>((nblocks + 255) / 256, 65535)), dim3(256), 0, ctx->stream, (const float*)sb->dev,
                       (signed char*)b->dev, b->scales, nblocks);
    LH_HIP(ctx, hipGetLastError());
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->ds->mu);
    lh_buf id = ctx->ds->next_id++;
    ctx->ds->bufs[id] = std::move(b);
    ctx->ds->bufs_gen.fetch_add(1, std::memory_order_release);
    *out = id;
    return LH_OK;
}

int lh_buf_read_q8(lh_ctx* ctx, lh_buf buf, uint64_t off, float* dst, uint64_t n) {
    Buffer* b;
    int rc = get_buf(ctx, buf, off, n, &b, "lh_buf_read_q8");
    if (rc) return rc;
    if (b->dtype != 7 || !dst) LH_FAIL(ctx, LH_EINVAL, "lh_buf_read_q8: not a block-int8 buffer");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<signed char> q(n);
    const uint64_t b0 = off / 32, b1 = (off + n + 31) / 32;
    std::vector<float> sc(b1 - b0);
    LH_HIP(ctx, hipMemcpy(q.data(), (signed char*)b->dev + off, n, hipMemcpyDeviceToHost));
    LH_HIP(ctx, hipMemcpy(sc.data(), b->scales + b0, (b1 - b0) * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) dst[i] = sc[(off + i) / 32 - b0] * (float)q[i];
    return LH_OK;
}

int lh_buf_free(lh_ctx* ctx, lh_buf buf) {
    if (!ctx) return LH_EINVAL;
    DeviceState* ds = ctx->ds;
    std::unique_ptr<Buffer> b;
    {
        std::lock_guard<std::mutex> lk(ds->mu);
        auto it = ds->bufs.find(buf);
        if (it == ds->bufs.end()) return LH_EINVAL;
        b = std::move(it->second);
        ds->bufs.erase(it);
        ds->bufs_gen.fetch_add(1, std::memory_order_release);
        if (b->key) ds->by_key.erase(b->key);
    }
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(b->dev);
    return LH_OK;
}

void* lh_buf_devptr(lh_ctx* ctx, lh_buf buf) {
    if (!ctx) return nullptr;
    Buffer* b = find_buffer(ctx->ds, buf);
    return b ? (void*)b->dev : nullptr;
}

uint64_t lh_buf_nfloats(lh_ctx* ctx, lh_buf buf) {
    if (!ctx) return 0;
    Buffer* b = find_buffer(ctx->ds, buf);
    return b ? b->nfloats : 0;
}

int lh_route_log(int on) {
    if (on) { std::lock_guard<std::mutex> g(lh::g_route_mu); lh::g_route_names.clear(); }
    lh::g_route_log_on.store(on ? 1 : 0);
    return LH_OK;
}
int64_t lh_route_names(char* buf, uint64_t cap) {
    std::lock_guard<std::mutex> g(lh::g_route_mu);
    std::string all;
    for (const auto& s : lh::g_route_names) { all += s; all += '\n'; }
    if (buf && cap) { const size_t n = std::min<size_t>(all.size(), (size_t)cap - 1); memcpy(buf, all.data(), n); buf[n] = 0; }
    return (int64_t)all.size() + 1;
}

}  // extern "C"
