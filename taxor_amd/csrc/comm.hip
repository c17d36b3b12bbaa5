// comm.hip -- the multi-GPU exchange steps of the path for a C++ host that drives several devices from ONE process
// (taxor search --gpus N): RCCL over xGMI, behind the C ABI (include/taxor_gpu.h, "Several GPUs of one node").
//
// Reads are independent (src/main/taxor_search.cpp:214), so the path shards by reads with the index replicated; that
// leaves exactly two exchange steps:
//   1. the index reaches every GPU: ONE upload over PCIe into device 0, overlapped chunk by chunk with an ncclBroadcast
//      of what has already arrived (instead of N uploads of the same 113 GB from host memory);
//   2. per round of batches, the per-read results of every device are gathered on device 0 with grouped
//      ncclSend/ncclRecv -- point-to-point, every peer on its own xGMI link, not a ring -- and leave through one D2H copy.
// The sizes that an MPI-style job would first all-gather are known to the one host process, so no size exchange exists.
//
// RCCL is bound at run time (dlopen): libtaxor_gpu.so carries no link-time dependency on it, a process that never
// creates a communicator never loads it, and inside a Python process that already holds torch's RCCL that copy is used.
// TAXOR_COMM_HOST is the same interface with every transfer staged through host memory (each device's own PCIe link);
// it is selectable for machines without a working RCCL and is what the parity tests compare the RCCL transport with.
// A failure is an error return with a message -- a communicator never silently changes transport.
#include "../../include/taxor_gpu_tools.h"
#include "tuning.h"
using taxor::tune_env;

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_create_empty(const taxor_hixf_view *v, int device, taxor_gpu_index **out);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_slab(taxor_gpu_index *idx, uint8_t **slab, uint64_t *slab_bytes,
                                                                      const uint64_t **ixf_off, uint64_t *n_ixf, int *device);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_upload(taxor_gpu_index *idx, const taxor_hixf_view *v,
                                                                        void (*progress)(void *, uint64_t), void *ctx);
extern "C" __attribute__((visibility("hidden"))) int taxor_searcher_device_results(taxor_gpu_searcher *s, const uint64_t **d_read_off,
                                                                                   const int64_t **d_user_bin, const uint32_t **d_count,
                                                                                   const uint32_t **d_n_hashes, uint64_t *n_reads,
                                                                                   uint64_t *n_tuples, int *device);

// The handful of RCCL declarations this file needs, stated here instead of including <rccl/rccl.h>: the library is bound with
// dlopen, so a ROCm installation without the RCCL development header still builds libtaxor_gpu.so (TAXOR_COMM_HOST serves
// there), and nothing is compiled against one copy's header and then run against another's.  These are the stable parts of
// the NCCL 2 API (opaque communicator handle, result code with 0 = success, ncclUint8 = 1); the copy found at run time is
// asked for its version and anything but major version 2 is refused (rccl() below).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;          // an enum in the header; 0 = ncclSuccess
typedef int ncclDataType_t;        // an enum in the header
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint8 = 1;

namespace {

int cfail(int code, const char *fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    taxor_set_last_error(buf);
    return code;
}

#define C_HIP(expr)                                                                                                   \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess) return cfail(TAXOR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// the RCCL entry points this file uses, resolved once
struct Rccl {
    void *handle = nullptr;
    std::string origin, error;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    int version = 0;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        // a copy that is already in the process (torch's, inside Python) first; then the ROCm installation's
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if ((x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) { x.origin = std::string(n) + " (already loaded)"; break; }
        if (!x.handle)
            for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) { x.origin = n; break; }
        if (!x.handle) {
            const char *e = dlerror();
            x.error = std::string("RCCL is not loadable (librccl.so.1): ") + (e ? e : "not found");
            return x;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(x.handle, name);
            if (!p && x.error.empty()) x.error = std::string("RCCL lacks the symbol ") + name;
            return p;
        };
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(sym("ncclGetVersion"));
        if (x.error.empty()) {
            // NCCL_VERSION(X,Y,Z): X*1000 + Y*100 + Z up to 2.8, X*10000 + Y*100 + Z from 2.9 on
            if (x.GetVersion(&x.version) != ncclSuccess) x.error = "ncclGetVersion failed";
            else {
                const int major = x.version >= 10000 ? x.version / 10000 : x.version / 1000;
                if (major != 2)
                    x.error = "the RCCL found (" + x.origin + ") reports version code " + std::to_string(x.version) +
                              ": this library speaks the NCCL 2 API only";
            }
        }
        x.ok = x.error.empty();
        return x;
    }();
    return r;
}

#define C_NCCL(expr)                                                                                                       \
    do {                                                                                                                   \
        ncclResult_t r_ = (expr);                                                                                          \
        if (r_ != ncclSuccess)                                                                                             \
            return cfail(TAXOR_E_HIP, "RCCL: %s failed: %s (%s:%d)", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?", \
                         __FILE__, __LINE__);                                                                              \
    } while (0)

// rank r's read offsets (local, n+1 of them) -> the gathered CSR: out[i] = in[i] + tuple_base for i < n
__global__ void k_rebase_offsets(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n, uint64_t tuple_base)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = in[i] + tuple_base;
}

template <typename T> struct GBuf {           // growable buffer on one device
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 1024;
        C_HIP(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace

struct taxor_gpu_comm {
    int transport = TAXOR_COMM_RCCL;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;        // RCCL transport: one per device, all in this process (ncclCommInitAll)
    std::vector<hipStream_t> streams;     // one per device, for the collectives and the result transfers
    // gather target on devices[0]
    GBuf<uint64_t> g_read_off, g_off_tmp;
    GBuf<int64_t> g_ub;
    GBuf<uint32_t> g_cnt, g_nh;
    // host side of the gathered results (valid until the next gather on this communicator)
    std::vector<uint64_t> h_read_off;
    std::vector<int64_t> h_ub;
    std::vector<uint32_t> h_cnt, h_nh;
    taxor_gpu_comm_stats stats{};
    bool self_exchange = false;           // test hook (taxor_gpu_comm_set_self_exchange): rank 0's own part of a gather travels
                                          // through ncclSend/ncclRecv to itself instead of a device-to-device copy
};

extern "C" __attribute__((visibility("hidden"))) void taxor_runtime_env_once();

// Known bytes through both exchange primitives of a fresh RCCL communicator, verified on the host: an in-place ncclBroadcast
// from rank 0 must arrive on every device, and a grouped ncclSend / ncclRecv from EVERY rank (rank 0 included: to itself) must
// land in rank 0's buffer at the sender's slot.  256 KiB per rank, a few milliseconds; run once per communicator so that a
// multi-GPU job whose xGMI / IPC path is broken on this machine ends here with a message instead of producing a report
// from bytes nobody checked (the transport is never changed silently: the error names --gather host).
static int comm_selftest(taxor_gpu_comm *c)
{
    Rccl &R = rccl();
    const size_t n = c->devices.size();
    constexpr size_t W = 1u << 16;                                       // 32-bit words per rank
    auto pat = [](uint32_t tag, size_t i) { return (tag + 1u) * 0x9E3779B1u ^ (uint32_t)i * 0x85EBCA77u; };
    std::vector<uint32_t *> bc(n, nullptr), tx(n, nullptr);
    uint32_t *rx = nullptr;
    std::vector<uint32_t> h(W);
    int rc = TAXOR_OK;
    std::string msg;
    auto bad = [&](const std::string &m) { if (rc == TAXOR_OK) { rc = TAXOR_E_HIP; msg = m; } };
    auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) bad(std::string("RCCL self-test: ") + what + ": " + hipGetErrorString(e)); return e == hipSuccess; };
    for (size_t i = 0; i < n && rc == TAXOR_OK; ++i) {
        if (!hip_ok(hipSetDevice(c->devices[i]), "hipSetDevice")) break;
        if (!hip_ok(hipMalloc((void **)&bc[i], W * 4), "hipMalloc") || !hip_ok(hipMalloc((void **)&tx[i], W * 4), "hipMalloc")) break;
        for (size_t k = 0; k < W; ++k) h[k] = pat((uint32_t)i, k);
        if (!hip_ok(hipMemcpy(tx[i], h.data(), W * 4, hipMemcpyHostToDevice), "hipMemcpy")) break;
        if (i == 0) {
            for (size_t k = 0; k < W; ++k) h[k] = pat(1000u, k);
            if (!hip_ok(hipMemcpy(bc[0], h.data(), W * 4, hipMemcpyHostToDevice), "hipMemcpy")) break;
            if (!hip_ok(hipMalloc((void **)&rx, n * W * 4), "hipMalloc") || !hip_ok(hipMemset(rx, 0, n * W * 4), "hipMemset")) break;
        } else if (!hip_ok(hipMemset(bc[i], 0, W * 4), "hipMemset")) break;
    }
    if (rc == TAXOR_OK) {
        ncclResult_t r = R.GroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
            (void)hipSetDevice(c->devices[i]);
            r = R.Broadcast(bc[i], bc[i], W * 4, ncclUint8, 0, c->comms[i], c->streams[i]);
        }
        const ncclResult_t r2 = R.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) bad(std::string("RCCL self-test: ncclBroadcast failed: ") + R.GetErrorString(r));
    }
    if (rc == TAXOR_OK) {
        ncclResult_t r = R.GroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
            (void)hipSetDevice(c->devices[i]);
            r = R.Send(tx[i], W * 4, ncclUint8, 0, c->comms[i], c->streams[i]);
            (void)hipSetDevice(c->devices[0]);
            if (r == ncclSuccess) r = R.Recv(rx + i * W, W * 4, ncclUint8, (int)i, c->comms[0], c->streams[0]);
        }
        const ncclResult_t r2 = R.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) bad(std::string("RCCL self-test: grouped ncclSend/ncclRecv failed: ") + R.GetErrorString(r));
    }
    for (size_t i = 0; i < n && rc == TAXOR_OK; ++i)
        if (hip_ok(hipSetDevice(c->devices[i]), "hipSetDevice")) hip_ok(hipStreamSynchronize(c->streams[i]), "stream synchronisation after the collectives");
    for (size_t i = 0; i < n && rc == TAXOR_OK; ++i) {
        if (!hip_ok(hipSetDevice(c->devices[i]), "hipSetDevice") || !hip_ok(hipMemcpy(h.data(), bc[i], W * 4, hipMemcpyDeviceToHost), "hipMemcpy")) break;
        for (size_t k = 0; k < W; ++k)
            if (h[k] != pat(1000u, k)) { bad("RCCL self-test: the broadcast from device " + std::to_string(c->devices[0]) + " delivered wrong bytes to device " + std::to_string(c->devices[i])); break; }
    }
    for (size_t i = 0; i < n && rc == TAXOR_OK; ++i) {
        if (!hip_ok(hipSetDevice(c->devices[0]), "hipSetDevice") || !hip_ok(hipMemcpy(h.data(), rx + i * W, W * 4, hipMemcpyDeviceToHost), "hipMemcpy")) break;
        for (size_t k = 0; k < W; ++k)
            if (h[k] != pat((uint32_t)i, k)) { bad("RCCL self-test: ncclSend from device " + std::to_string(c->devices[i]) + " arrived wrong on device " + std::to_string(c->devices[0])); break; }
    }
    for (size_t i = 0; i < n; ++i) {
        (void)hipSetDevice(c->devices[i]);
        if (bc[i]) (void)hipFree(bc[i]);
        if (tx[i]) (void)hipFree(tx[i]);
    }
    (void)hipSetDevice(c->devices[0]);
    if (rx) (void)hipFree(rx);
    if (rc != TAXOR_OK) return cfail(rc, "%s (the host transport, --gather host, stages the same transfers through host memory)", msg.c_str());
    c->stats.selftest_bytes = (uint64_t)n * W * 4 * 2;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_comm_create(const int *devices, uint32_t n_devices, int transport, taxor_gpu_comm **out)
{
    taxor_runtime_env_once();
    if (!devices || !n_devices || !out) return cfail(TAXOR_E_ARG, "comm_create: no devices");
    if (transport != TAXOR_COMM_RCCL && transport != TAXOR_COMM_HOST) return cfail(TAXOR_E_ARG, "comm_create: unknown transport %d", transport);
    int have = 0;
    C_HIP(hipGetDeviceCount(&have));
    for (uint32_t i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= have) return cfail(TAXOR_E_ARG, "comm_create: device %d does not exist (%d visible)", devices[i], have);
    auto c = new taxor_gpu_comm();
    c->transport = transport;
    c->devices.assign(devices, devices + n_devices);
    c->streams.assign(n_devices, nullptr);
    for (uint32_t i = 0; i < n_devices; ++i) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            taxor_gpu_comm_destroy(c);
            return cfail(TAXOR_E_HIP, "comm_create: stream on device %d: %s", devices[i], hipGetErrorString(e));
        }
    }
    if (transport == TAXOR_COMM_RCCL) {
        for (uint32_t i = 0; i < n_devices; ++i)
            for (uint32_t j = i + 1; j < n_devices; ++j)
                if (devices[i] == devices[j]) {
                    taxor_gpu_comm_destroy(c);
                    return cfail(TAXOR_E_ARG, "comm_create: device %d is listed twice; RCCL takes one rank per device (use the host transport)", devices[i]);
                }
        Rccl &R = rccl();
        if (!R.ok) {
            taxor_gpu_comm_destroy(c);
            return cfail(TAXOR_E_HIP, "comm_create: %s", R.error.c_str());
        }
        c->comms.assign(n_devices, nullptr);
        const ncclResult_t r = R.CommInitAll(c->comms.data(), (int)n_devices, c->devices.data());
        if (r != ncclSuccess) {
            c->comms.clear();
            taxor_gpu_comm_destroy(c);
            return cfail(TAXOR_E_HIP, "comm_create: ncclCommInitAll over %u devices failed: %s", n_devices, R.GetErrorString(r));
        }
        if (int rc = comm_selftest(c)) {
            const std::string msg = taxor_gpu_last_error();
            taxor_gpu_comm_destroy(c);
            return cfail(rc, "comm_create: %s", msg.c_str());
        }
    }
    *out = c;
    return TAXOR_OK;
}

extern "C" void taxor_gpu_comm_destroy(taxor_gpu_comm *c)
{
    if (!c) return;
    for (size_t i = 0; i < c->devices.size(); ++i) {
        (void)hipSetDevice(c->devices[i]);
        if (i < c->streams.size() && c->streams[i]) (void)hipStreamSynchronize(c->streams[i]);
        if (i < c->comms.size() && c->comms[i]) (void)rccl().CommDestroy(c->comms[i]);
        if (i < c->streams.size() && c->streams[i]) (void)hipStreamDestroy(c->streams[i]);
    }
    if (!c->devices.empty()) {
        (void)hipSetDevice(c->devices[0]);
        c->g_read_off.release(); c->g_off_tmp.release(); c->g_ub.release(); c->g_cnt.release(); c->g_nh.release();
    }
    delete c;
}

extern "C" int taxor_gpu_comm_set_self_exchange(taxor_gpu_comm *c, int on)
{
    if (!c) return cfail(TAXOR_E_ARG, "comm_set_self_exchange: null communicator");
    if (on && c->transport != TAXOR_COMM_RCCL) return cfail(TAXOR_E_ARG, "comm_set_self_exchange: only the RCCL transport has a send/recv path");
    c->self_exchange = on != 0;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_comm_info(const taxor_gpu_comm *c, taxor_gpu_comm_stats *out)
{
    if (!c || !out) return cfail(TAXOR_E_ARG, "comm_info: null argument");
    *out = c->stats;
    out->rccl_version = c->transport == TAXOR_COMM_RCCL ? rccl().version : 0;
    out->transport = c->transport;
    out->n_devices = (uint32_t)c->devices.size();
    return TAXOR_OK;
}

// =====================================================================================================================
// exchange step 1: the index
// =====================================================================================================================
extern "C" int taxor_gpu_index_create_replicated(taxor_gpu_comm *c, const taxor_hixf_view *view, taxor_gpu_index **out)
{
    if (!c || !view || !out) return cfail(TAXOR_E_ARG, "index_create_replicated: null argument");
    const size_t n = c->devices.size();
    for (size_t i = 0; i < n; ++i) out[i] = nullptr;
    auto destroy_all = [&] {
        for (size_t i = 0; i < n; ++i) { if (out[i]) taxor_gpu_index_destroy(out[i]); out[i] = nullptr; }
    };
    const auto t0 = std::chrono::steady_clock::now();
    // (An RCCL communicator of ONE rank takes the RCCL path below too: create-empty, the upload thread with its watermark, the
    // broadcast loop behind it -- an in-place ncclBroadcast on one rank moves nothing, but every line a larger run executes
    // is executed, which is what a one-GPU test box can verify.)
    if (c->transport == TAXOR_COMM_HOST) {
        // every replica through its own PCIe link, concurrently (a device listed twice gets two replicas)
        std::vector<std::thread> up;
        std::vector<std::string> errs(n);
        std::vector<int> rcs(n, 0);
        for (size_t i = 0; i < n; ++i)
            up.emplace_back([&, i] {
                rcs[i] = taxor_gpu_index_create(view, c->devices[i], &out[i]);
                if (rcs[i]) errs[i] = taxor_gpu_last_error();
            });
        for (auto &t : up) t.join();
        for (size_t i = 0; i < n; ++i)
            if (rcs[i]) {
                const int rc = rcs[i];
                const std::string msg = errs[i];
                destroy_all();
                return cfail(rc, "%s", msg.c_str());
            }
        c->stats.index_bytes = taxor_gpu_index_data_bytes(out[0]);
        c->stats.index_upload_bytes = c->stats.index_bytes * n;
        c->stats.index_broadcast_bytes = 0;
        c->stats.index_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return TAXOR_OK;
    }

    // RCCL: tables everywhere, fingerprints over PCIe into device 0 only, broadcast behind the upload
    for (size_t i = 0; i < n; ++i)
        if (int rc = taxor_index_create_empty(view, c->devices[i], &out[i])) {
            const std::string msg = taxor_gpu_last_error();
            destroy_all();
            return cfail(rc, "%s", msg.c_str());
        }
    std::vector<uint8_t *> slab(n, nullptr);
    uint64_t slab_bytes = 0, n_ixf = 0;
    for (size_t i = 0; i < n; ++i) {
        int dev;
        uint64_t sb, ni;
        const uint64_t *off;
        if (taxor_index_slab(out[i], &slab[i], &sb, &off, &ni, &dev)) { destroy_all(); return cfail(TAXOR_E_INTERNAL, "index_create_replicated: slab"); }
        if (i == 0) { slab_bytes = sb; n_ixf = ni; }
        else if (sb != slab_bytes) { destroy_all(); return cfail(TAXOR_E_INTERNAL, "index_create_replicated: replicas differ in size"); }
    }
    // Upload thread: the library's own upload into device 0 (pieces, possibly several threads; api.hip), which reports the
    // slab offset below which device 0 holds final bytes -- the watermark the broadcast follows.
    static const uint64_t piece = [] { const char *e = tune_env("TAXOR_COMM_PIECE_MB"); const long v = e ? atol(e) : 0; return (uint64_t)(v > 0 ? v : 1024) << 20; }();
    std::atomic<uint64_t> watermark{0};
    std::atomic<int> up_rc{0};
    std::string up_err;
    std::thread uploader([&] {
        const int rc = taxor_index_upload(out[0], view, [](void *ctx, uint64_t b) { static_cast<std::atomic<uint64_t> *>(ctx)->store(b); }, &watermark);
        if (rc) { up_err = taxor_gpu_last_error(); up_rc = rc; }
        watermark = slab_bytes;           // also on failure: the broadcast loop must end
    });
    // Broadcast behind it: whatever lies below the watermark and has not been sent, once it is worth a collective
    // (>= one piece) or the upload is complete.  One ncclBroadcast per device inside a group, root = rank 0.
    Rccl &R = rccl();
    uint64_t sent = 0;
    int rc = TAXOR_OK;
    std::string bc_err;
    while (sent < slab_bytes) {
        const uint64_t wm = watermark.load();
        if (wm - sent < piece && wm < slab_bytes) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
        const uint64_t len = wm - sent;
        ncclResult_t r = R.GroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
            (void)hipSetDevice(c->devices[i]);           // one thread drives every rank of the group: make the rank's device current
            r = R.Broadcast(slab[i] + sent, slab[i] + sent, (size_t)len, ncclUint8, 0, c->comms[i], c->streams[i]);   // in place; the send buffer counts on the root only
        }
        const ncclResult_t r2 = R.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { rc = TAXOR_E_HIP; bc_err = std::string("ncclBroadcast of the index: ") + R.GetErrorString(r); break; }
        c->stats.index_broadcast_calls++;
        sent = wm;
    }
    uploader.join();
    if (rc == TAXOR_OK)
        for (size_t i = 0; i < n; ++i) {
            hipError_t e = hipSetDevice(c->devices[i]);
            if (e == hipSuccess) e = hipStreamSynchronize(c->streams[i]);
            if (e != hipSuccess) { rc = TAXOR_E_HIP; bc_err = std::string("index broadcast: ") + hipGetErrorString(e); break; }
        }
    if (up_rc.load()) { rc = up_rc.load(); bc_err = "index upload to device " + std::to_string(c->devices[0]) + ": " + up_err; }
    uint64_t uploaded = 0;
    for (uint64_t i = 0; i < n_ixf; ++i)
        if (view->source || view->ixf[i].data) uploaded += 3 * view->ixf[i].seg_len * view->ixf[i].stride;
    if (rc) {
        destroy_all();
        return cfail(rc, "%s", bc_err.c_str());
    }
    c->stats.index_bytes = taxor_gpu_index_data_bytes(out[0]);
    c->stats.index_upload_bytes = uploaded;
    c->stats.index_broadcast_bytes = slab_bytes * (n - 1);
    c->stats.index_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return TAXOR_OK;
}

// =====================================================================================================================
// exchange step 2: the per-read results of one round (searcher i ran its own batch on devices[i])
// =====================================================================================================================
extern "C" int taxor_gpu_gather_results(taxor_gpu_comm *c, taxor_gpu_searcher *const *searchers, taxor_gpu_results *out)
{
    if (!c || !searchers || !out) return cfail(TAXOR_E_ARG, "gather_results: null argument");
    const size_t n = c->devices.size();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> nr(n), nt(n), rbase(n + 1, 0), tbase(n + 1, 0);
    std::vector<const uint64_t *> d_ro(n);
    std::vector<const int64_t *> d_ub(n);
    std::vector<const uint32_t *> d_ct(n), d_nh(n);
    for (size_t i = 0; i < n; ++i) {
        if (!searchers[i]) return cfail(TAXOR_E_ARG, "gather_results: searcher %zu is null", i);
        int dev = -1;
        if (int rc = taxor_searcher_device_results(searchers[i], &d_ro[i], &d_ub[i], &d_ct[i], &d_nh[i], &nr[i], &nt[i], &dev)) return rc;
        if (dev != c->devices[i]) return cfail(TAXOR_E_ARG, "gather_results: searcher %zu lives on device %d, the communicator's rank %zu is device %d", i, dev, i, c->devices[i]);
        rbase[i + 1] = rbase[i] + nr[i];
        tbase[i + 1] = tbase[i] + nt[i];
    }
    const uint64_t NR = rbase[n], NT = tbase[n];
    c->h_read_off.resize(NR + 1);
    c->h_ub.resize(NT);
    c->h_cnt.resize(NT);
    c->h_nh.resize(NR);

    if (c->transport == TAXOR_COMM_HOST) {
        // every device's results over its own PCIe link into the host arrays, offsets rebased on the host
        for (size_t i = 0; i < n; ++i) {
            C_HIP(hipSetDevice(c->devices[i]));
            hipStream_t st = c->streams[i];
            if (nr[i]) {
                C_HIP(hipMemcpyAsync(c->h_read_off.data() + rbase[i], d_ro[i], nr[i] * 8, hipMemcpyDeviceToHost, st));
                C_HIP(hipMemcpyAsync(c->h_nh.data() + rbase[i], d_nh[i], nr[i] * 4, hipMemcpyDeviceToHost, st));
            }
            if (nt[i]) {
                C_HIP(hipMemcpyAsync(c->h_ub.data() + tbase[i], d_ub[i], nt[i] * 8, hipMemcpyDeviceToHost, st));
                C_HIP(hipMemcpyAsync(c->h_cnt.data() + tbase[i], d_ct[i], nt[i] * 4, hipMemcpyDeviceToHost, st));
            }
        }
        for (size_t i = 0; i < n; ++i) {
            C_HIP(hipSetDevice(c->devices[i]));
            C_HIP(hipStreamSynchronize(c->streams[i]));
            if (tbase[i])
                for (uint64_t r = 0; r < nr[i]; ++r) c->h_read_off[rbase[i] + r] += tbase[i];
        }
        c->h_read_off[NR] = NT;
    } else {
        Rccl &R = rccl();
        C_HIP(hipSetDevice(c->devices[0]));
        if (c->g_read_off.reserve(NR + 1) || c->g_off_tmp.reserve(NR + n) || c->g_ub.reserve(NT + 1) || c->g_cnt.reserve(NT + 1) ||
            c->g_nh.reserve(NR + 1))
            return TAXOR_E_HIP;
        hipStream_t s0 = c->streams[0];
        // rank 0's own part: device-to-device on device 0
        const bool self = c->self_exchange;     // test hook: rank 0's part goes through the grouped send/recv below like a peer's
        if (nr[0] && !self) {
            C_HIP(hipMemcpyAsync(c->g_off_tmp.p, d_ro[0], nr[0] * 8, hipMemcpyDeviceToDevice, s0));
            C_HIP(hipMemcpyAsync(c->g_nh.p, d_nh[0], nr[0] * 4, hipMemcpyDeviceToDevice, s0));
        }
        if (nt[0] && !self) {
            C_HIP(hipMemcpyAsync(c->g_ub.p, d_ub[0], nt[0] * 8, hipMemcpyDeviceToDevice, s0));
            C_HIP(hipMemcpyAsync(c->g_cnt.p, d_ct[0], nt[0] * 4, hipMemcpyDeviceToDevice, s0));
        }
        // every peer -> rank 0, all transfers in one group: each pair (peer, 0) has its own xGMI link, so the n-1
        // transfers proceed side by side; byte counts, so one datatype serves all four arrays
        if (n > 1 || self) {
            C_NCCL(R.GroupStart());
            ncclResult_t r = ncclSuccess;
            for (size_t i = self ? 0 : 1; i < n && r == ncclSuccess; ++i) {
                (void)hipSetDevice(c->devices[i]);       // the sends belong to rank i's device ...
                if (nr[i] && r == ncclSuccess) r = R.Send(d_ro[i], nr[i] * 8, ncclUint8, 0, c->comms[i], c->streams[i]);
                if (nr[i] && r == ncclSuccess) r = R.Send(d_nh[i], nr[i] * 4, ncclUint8, 0, c->comms[i], c->streams[i]);
                if (nt[i] && r == ncclSuccess) r = R.Send(d_ub[i], nt[i] * 8, ncclUint8, 0, c->comms[i], c->streams[i]);
                if (nt[i] && r == ncclSuccess) r = R.Send(d_ct[i], nt[i] * 4, ncclUint8, 0, c->comms[i], c->streams[i]);
                (void)hipSetDevice(c->devices[0]);       // ... the matching receives to rank 0's
                if (nr[i] && r == ncclSuccess) r = R.Recv(c->g_off_tmp.p + rbase[i], nr[i] * 8, ncclUint8, (int)i, c->comms[0], s0);
                if (nr[i] && r == ncclSuccess) r = R.Recv(c->g_nh.p + rbase[i], nr[i] * 4, ncclUint8, (int)i, c->comms[0], s0);
                if (nt[i] && r == ncclSuccess) r = R.Recv(c->g_ub.p + tbase[i], nt[i] * 8, ncclUint8, (int)i, c->comms[0], s0);
                if (nt[i] && r == ncclSuccess) r = R.Recv(c->g_cnt.p + tbase[i], nt[i] * 4, ncclUint8, (int)i, c->comms[0], s0);
            }
            const ncclResult_t r2 = R.GroupEnd();
            if (r == ncclSuccess) r = r2;
            if (r != ncclSuccess) return cfail(TAXOR_E_HIP, "gather_results: grouped ncclSend/ncclRecv failed: %s", R.GetErrorString(r));
        }
        // rebase the offsets on device 0 (stream order: behind the receives), close the CSR, one D2H per array
        C_HIP(hipSetDevice(c->devices[0]));
        for (size_t i = 0; i < n; ++i)
            if (nr[i]) {
                const uint32_t grid = (uint32_t)std::min<uint64_t>((nr[i] + 255) / 256, 1024);
                hipLaunchKernelGGL(k_rebase_offsets, dim3(grid), dim3(256), 0, s0, c->g_off_tmp.p + rbase[i], c->g_read_off.p + rbase[i], nr[i], tbase[i]);
            }
        C_HIP(hipGetLastError());
        C_HIP(hipMemcpyAsync(c->g_read_off.p + NR, &NT, 8, hipMemcpyHostToDevice, s0));
        C_HIP(hipMemcpyAsync(c->h_read_off.data(), c->g_read_off.p, (NR + 1) * 8, hipMemcpyDeviceToHost, s0));
        if (NR) C_HIP(hipMemcpyAsync(c->h_nh.data(), c->g_nh.p, NR * 4, hipMemcpyDeviceToHost, s0));
        if (NT) {
            C_HIP(hipMemcpyAsync(c->h_ub.data(), c->g_ub.p, NT * 8, hipMemcpyDeviceToHost, s0));
            C_HIP(hipMemcpyAsync(c->h_cnt.data(), c->g_cnt.p, NT * 4, hipMemcpyDeviceToHost, s0));
        }
        // the sends complete on their own devices' streams; the receives and copies on device 0's
        for (size_t i = 1; i < n; ++i) {
            C_HIP(hipSetDevice(c->devices[i]));
            C_HIP(hipStreamSynchronize(c->streams[i]));
        }
        C_HIP(hipSetDevice(c->devices[0]));
        C_HIP(hipStreamSynchronize(s0));
        if (self) c->stats.self_exchange_bytes += nr[0] * 12 + nt[0] * 12;
    }
    out->n_reads = NR;
    out->n_tuples = NT;
    out->read_off = c->h_read_off.data();
    out->user_bin = c->h_ub.data();
    out->count = c->h_cnt.data();
    out->n_hashes = c->h_nh.data();
    c->stats.gathers++;
    c->stats.gather_bytes += (NR - nr[0]) * 12 + (NT - nt[0]) * 12;
    c->stats.gather_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return TAXOR_OK;
}
