// Single-process multi-GPU entry points: one rn_handle per device, image batches sharded
// contiguously, ONE collective on the data path -- an RCCL all-gather of every device's packed results
// (probs [cap, C] float32 followed by ids [cap] int64: 32 bytes per image at C = 6) over xGMI.
//
// The reference is one tf.Session on one device (network.py:89); this is what a non-Python host uses to drive
// several MI355X from one process.  (The Python benchmark keeps one process per GPU with torch.distributed.)
//
// librccl is resolved with dlopen at the first rn_group_create: a process that never asks for a group does not
// load it (ROOMNET_RCCL_LIB names another file to load instead; read once, at that first call, never on the launch
// path).  One host thread drives all devices: per device one stream (the handle's own), the forward pass is
// enqueued device by device, the all-gather is one ncclGroupStart/End bracket over all communicators, so the
// collective is stream-ordered behind each device's head kernel without a host synchronisation in between.
// The host-buffer entry rn_group_forward_u8 hands the shards to one PERSISTENT upload thread per device (started by
// rn_group_create, parked on a condition variable between calls, joined by rn_group_destroy): a copy out of pageable
// memory blocks the thread that issues it, and one thread issuing eight of them would run them one after the other.
// From pinned memory (rn_host_alloc) hipMemcpyAsync returns at once; the workers then cost two wake-ups per call.
// Every entry point leaves the caller's current HIP device as it found it.
//
// Groups of more than one device have NOT run on hardware yet (the development pool has one MI355X per box): the
// one-device group is what tests/test_group.py validates.
#include "rn_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace {

struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void* dl = nullptr;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

bool load_rccl() {
    std::call_once(g_rccl_once, [] {
        std::string last_error = "?";
        const char* forced = std::getenv("ROOMNET_RCCL_LIB");
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (forced && *forced) name = forced;
            g_rccl.dl = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.dl) break;
            const char* e = dlerror();           // (one call: dlerror() clears the error state it returns)
            if (e) last_error = e;
            if (forced && *forced) break;
        }
        if (!g_rccl.dl) {
            g_rccl_error = "dlopen(librccl) failed: " + last_error;
            return;
        }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(g_rccl.dl, n);
            if (!p && g_rccl_error.empty()) g_rccl_error = std::string("librccl lacks ") + n;
            return p;
        };
        g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
        g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
        g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
        g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return g_rccl_error.empty();
}

#define RN_NCCL(expr)                                                                                      \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            rn_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);   \
            return RN_E_HIP;                                                                               \
        }                                                                                                  \
    } while (0)

// the caller's current HIP device, put back when an entry point returns (the per-handle entry points do the same)
struct CurrentDeviceRestore {
    int prev = -1;
    CurrentDeviceRestore() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~CurrentDeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

// One upload worker per device: bound to its device once, then: wait for a job (source, bytes) -> hipMemcpyAsync on the
// device's stream -> report.  post() / wait() are called by the thread that owns the group (calls on a group are serialised).
struct UploadWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int device = 0;
    hipStream_t stream = nullptr;
    void* dst = nullptr;
    const void* src = nullptr;      // job: non-null while one is pending
    size_t bytes = 0;
    bool pending = false, done = false, quit = false;
    hipError_t rc = hipSuccess;

    void run() {
        hipError_t bind = hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return pending || quit; });
            if (quit) return;
            const void* s = src;
            const size_t b = bytes;
            lk.unlock();
            hipError_t e = bind != hipSuccess ? bind : hipMemcpyAsync(dst, s, b, hipMemcpyHostToDevice, stream);
            lk.lock();
            rc = e;
            pending = false;
            done = true;
            cv.notify_all();
        }
    }
    void post(const void* s, size_t b) {
        std::lock_guard<std::mutex> lk(mu);
        src = s;
        bytes = b;
        done = false;
        pending = true;
        cv.notify_all();
    }
    hipError_t wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        done = false;
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct rn_group {
    int ndev = 0;
    int cap = 0;                       // images per device and call (max_batch_per_device)
    int num_classes = 0;
    int im_side = 0;
    size_t slot_bytes = 0;             // packed result of one device: cap * (C * 4 + 8)
    std::vector<int> devices;
    std::vector<rn_handle*> handles;
    std::vector<ncclComm_t> comms;
    std::vector<uint8_t*> d_in;        // [cap, S, S, 3] staging for the host-buffer entry
    std::vector<uint8_t*> d_send;      // packed results of this device
    std::vector<uint8_t*> d_recv;      // [ndev][slot_bytes]: every device's results, on every device
    std::vector<uint8_t> h_recv;       // host copy of device 0's gathered buffer
    std::vector<int> counts;           // images per device of the last call
    std::vector<std::unique_ptr<UploadWorker>> workers;   // host-buffer entry: one persistent upload thread per device
};

extern "C" void rn_group_destroy(rn_group* g) {
    if (!g) return;
    CurrentDeviceRestore restore;
    for (auto& w : g->workers)
        if (w) w->stop();
    for (int d = 0; d < static_cast<int>(g->handles.size()); ++d) {
        if (hipSetDevice(g->devices[d]) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        if (d < static_cast<int>(g->comms.size()) && g->comms[d] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comms[d]);
        if (d < static_cast<int>(g->d_in.size()) && g->d_in[d]) (void)hipFree(g->d_in[d]);
        if (d < static_cast<int>(g->d_send.size()) && g->d_send[d]) (void)hipFree(g->d_send[d]);
        if (d < static_cast<int>(g->d_recv.size()) && g->d_recv[d]) (void)hipFree(g->d_recv[d]);
        rn_destroy(g->handles[d]);
    }
    delete g;
}

extern "C" int rn_group_create(const rn_weights* w, int ndev, const int* devices, int dtype, int max_batch_per_device,
                               unsigned flags, rn_group** out) {
    if (!out) {
        rn_set_error("rn_group_create: null out pointer");
        return RN_E_INVALID;
    }
    *out = nullptr;
    if (!w || ndev < 1 || ndev > 64 || max_batch_per_device < 1) {
        rn_set_error("rn_group_create: bad argument (ndev %d, max_batch_per_device %d)", ndev, max_batch_per_device);
        return RN_E_INVALID;
    }
    for (int a = 0; devices && a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devices[a] == devices[b]) {
                rn_set_error("rn_group_create: device %d listed twice", devices[a]);
                return RN_E_INVALID;
            }
    if (!load_rccl()) {
        rn_set_error("rn_group_create: %s", g_rccl_error.c_str());
        return RN_E_STATE;
    }
    CurrentDeviceRestore restore;
    rn_group* g = new (std::nothrow) rn_group();
    if (!g) {
        rn_set_error("rn_group_create: out of host memory");
        return RN_E_NOMEM;
    }
    g->ndev = ndev;
    g->cap = max_batch_per_device;
    g->num_classes = w->num_classes;
    g->im_side = w->im_side;
    g->slot_bytes = static_cast<size_t>(g->cap) * (static_cast<size_t>(w->num_classes) * 4 + 8);
    g->devices.resize(ndev);
    for (int d = 0; d < ndev; ++d) g->devices[d] = devices ? devices[d] : d;
    g->comms.assign(ndev, nullptr);
    g->d_in.assign(ndev, nullptr);
    g->d_send.assign(ndev, nullptr);
    g->d_recv.assign(ndev, nullptr);
    g->counts.assign(ndev, 0);
    auto fail = [&](int code) {
        rn_group_destroy(g);
        return code;
    };
    for (int d = 0; d < ndev; ++d) {
        rn_handle* h = nullptr;
        int rc = rn_create(w, g->devices[d], dtype, g->cap, flags, &h);
        if (rc != RN_OK) return fail(rc);
        g->handles.push_back(h);
    }
    const size_t in_bytes = static_cast<size_t>(g->cap) * w->im_side * w->im_side * 3;
    for (int d = 0; d < ndev; ++d) {
        if (hipSetDevice(g->devices[d]) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&g->d_in[d]), in_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&g->d_send[d]), g->slot_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&g->d_recv[d]), g->slot_bytes * ndev) != hipSuccess) {
            rn_set_error("rn_group_create: device %d: buffer allocation failed", g->devices[d]);
            return fail(RN_E_NOMEM);
        }
        // (a device that never gets an image still sends its slot in every all-gather: zeros, not uninitialised memory)
        if (hipMemset(g->d_send[d], 0, g->slot_bytes) != hipSuccess) {
            rn_set_error("rn_group_create: device %d: hipMemset failed", g->devices[d]);
            return fail(RN_E_HIP);
        }
    }
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), ndev, g->devices.data());
    if (r != ncclSuccess) {
        rn_set_error("rn_group_create: ncclCommInitAll over %d devices failed: %s", ndev, g_rccl.GetErrorString(r));
        return fail(RN_E_HIP);
    }
    g->h_recv.resize(g->slot_bytes * ndev);
    // (no C++ exception may cross the C ABI: a thread that cannot be started fails the call with a status)
    try {
        for (int d = 0; d < ndev; ++d) {
            g->workers.emplace_back(new UploadWorker());
            UploadWorker& w = *g->workers.back();
            w.device = g->devices[d];
            w.stream = g->handles[d]->stream;
            w.dst = g->d_in[d];
            w.th = std::thread([&w] { w.run(); });
        }
    } catch (const std::exception& ex) {
        rn_set_error("rn_group_create: cannot start the upload threads: %s", ex.what());
        return fail(RN_E_STATE);
    }
    *out = g;
    return RN_OK;
}

extern "C" int rn_group_size(const rn_group* g) { return g ? g->ndev : 0; }

extern "C" rn_handle* rn_group_handle(rn_group* g, int index) {
    if (!g || index < 0 || index >= g->ndev) {
        rn_set_error("rn_group_handle: bad argument");
        return nullptr;
    }
    return g->handles[index];
}

// contiguous split of n images: the first n % ndev devices take one more
static void shard(int n, int ndev, int d, int* lo, int* cnt) {
    const int base = n / ndev, rem = n % ndev;
    *lo = d * base + (d < rem ? d : rem);
    *cnt = base + (d < rem ? 1 : 0);
}

// The shard / slot plan of rn_group_forward_u8 as a function of (n, ndev) alone: no group, no device.
extern "C" int rn_group_plan(int n, int ndev, int max_batch_per_device, int num_classes, int* counts, int* offsets, size_t* slot_bytes) {
    if (ndev < 1 || n < 0 || max_batch_per_device < 1 || num_classes < 1 || !counts || !offsets) {
        rn_set_error("rn_group_plan: bad argument (n %d, ndev %d, max_batch_per_device %d, num_classes %d)", n, ndev, max_batch_per_device, num_classes);
        return RN_E_INVALID;
    }
    if (static_cast<long long>(n) > static_cast<long long>(ndev) * max_batch_per_device) {
        rn_set_error("rn_group_plan: n = %d out of range (0..%lld)", n, static_cast<long long>(ndev) * max_batch_per_device);
        return RN_E_RANGE;
    }
    for (int d = 0; d < ndev; ++d) shard(n, ndev, d, &offsets[d], &counts[d]);
    if (slot_bytes) *slot_bytes = static_cast<size_t>(max_batch_per_device) * (static_cast<size_t>(num_classes) * 4 + 8);
    return RN_OK;
}

static int group_run(rn_group* g, const uint8_t* const* d_shards, const int* counts) {
    // forward pass per device into its packed send buffer, then ONE all-gather bracket.  A device with counts[d] == 0 still
    // takes part in the collective (every rank must): it contributes its slot as it stands -- the results of its last
    // non-empty call, or zeros before the first -- and only the first counts[d] entries of a slot are results of THIS call.
    for (int d = 0; d < g->ndev; ++d) {
        g->counts[d] = counts[d];
        if (counts[d] == 0) continue;
        float* d_probs = reinterpret_cast<float*>(g->d_send[d]);
        int64_t* d_ids = reinterpret_cast<int64_t*>(g->d_send[d] + static_cast<size_t>(g->cap) * g->num_classes * 4);
        int rc = rn_forward_u8_device(g->handles[d], d_shards[d], counts[d], d_probs, d_ids);
        if (rc != RN_OK) return rc;
    }
    RN_NCCL(g_rccl.GroupStart());
    // (an error inside the bracket still closes it: RCCL keeps a group call open per thread until GroupEnd)
    for (int d = 0; d < g->ndev; ++d) {
        if (hipSetDevice(g->devices[d]) != hipSuccess) {
            (void)g_rccl.GroupEnd();
            rn_set_error("rn_group: hipSetDevice(%d) failed", g->devices[d]);
            return RN_E_HIP;
        }
        const ncclResult_t r = g_rccl.AllGather(g->d_send[d], g->d_recv[d], g->slot_bytes, ncclUint8, g->comms[d], g->handles[d]->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            rn_set_error("ncclAllGather on device %d failed: %s", g->devices[d], g_rccl.GetErrorString(r));
            return RN_E_HIP;
        }
    }
    RN_NCCL(g_rccl.GroupEnd());
    return RN_OK;
}

extern "C" int rn_group_forward_u8_device(rn_group* g, const uint8_t* const* d_shards, const int* counts) {
    if (!g || !d_shards || !counts) {
        rn_set_error("rn_group_forward_u8_device: null argument");
        return RN_E_INVALID;
    }
    for (int d = 0; d < g->ndev; ++d)
        if (counts[d] < 0 || counts[d] > g->cap || (counts[d] > 0 && !d_shards[d])) {
            rn_set_error("rn_group_forward_u8_device: device %d: %d images out of range (max_batch_per_device %d)", d, counts[d], g->cap);
            return RN_E_RANGE;
        }
    CurrentDeviceRestore restore;
    return group_run(g, d_shards, counts);
}

extern "C" int rn_group_result_buffer(rn_group* g, int index, void** d_gathered, size_t* slot_bytes) {
    if (!g || index < 0 || index >= g->ndev || !d_gathered) {
        rn_set_error("rn_group_result_buffer: bad argument");
        return RN_E_INVALID;
    }
    *d_gathered = g->d_recv[index];
    if (slot_bytes) *slot_bytes = g->slot_bytes;
    return RN_OK;
}

extern "C" int rn_group_sync(rn_group* g) {
    if (!g) {
        rn_set_error("null group");
        return RN_E_INVALID;
    }
    CurrentDeviceRestore restore;
    for (int d = 0; d < g->ndev; ++d) {
        RN_HIP(hipSetDevice(g->devices[d]));
        RN_HIP(hipStreamSynchronize(g->handles[d]->stream));
    }
    return RN_OK;
}

extern "C" int rn_group_forward_u8(rn_group* g, const uint8_t* bgr_nhwc, int n, float* probs, int64_t* ids) {
    if (!g || !bgr_nhwc || !probs || !ids) {
        rn_set_error("rn_group_forward_u8: null argument");
        return RN_E_INVALID;
    }
    if (n < 1 || n > g->cap * g->ndev) {
        rn_set_error("rn_group_forward_u8: n = %d out of range (1..%d)", n, g->cap * g->ndev);
        return RN_E_RANGE;
    }
    CurrentDeviceRestore restore;
    const size_t img_bytes = static_cast<size_t>(g->im_side) * g->im_side * 3;
    std::vector<const uint8_t*> shards(g->ndev, nullptr);
    std::vector<int> counts(g->ndev, 0);
    // the devices' persistent upload threads: hipMemcpyAsync out of pageable memory stages through the runtime's pinned
    // buffers on the CALLING thread and returns when the source has been read -- issued from one thread the devices' uploads
    // (38.5 MB each at 256 images of 224 x 224) would follow one another (~1.5 ms each) in front of 1.6 ms of compute
    for (int d = 0; d < g->ndev; ++d) {
        int lo, cnt;
        shard(n, g->ndev, d, &lo, &cnt);
        counts[d] = cnt;
        shards[d] = g->d_in[d];
        if (cnt == 0) continue;
        g->workers[d]->post(bgr_nhwc + static_cast<size_t>(lo) * img_bytes, static_cast<size_t>(cnt) * img_bytes);
    }
    hipError_t up_fail = hipSuccess;
    int up_dev = -1;
    for (int d = 0; d < g->ndev; ++d) {
        if (counts[d] == 0) continue;
        const hipError_t e = g->workers[d]->wait();           // (every posted job is waited for, also after a failure)
        if (e != hipSuccess && up_fail == hipSuccess) {
            up_fail = e;
            up_dev = d;
        }
    }
    if (up_fail != hipSuccess) {
        rn_set_error("rn_group_forward_u8: upload to device %d failed: %s", g->devices[up_dev], hipGetErrorString(up_fail));
        return RN_E_HIP;
    }
    int rc = group_run(g, shards.data(), counts.data());
    if (rc != RN_OK) return rc;
    // every device holds every device's results; read them back from the first one
    RN_HIP(hipSetDevice(g->devices[0]));
    RN_HIP(hipMemcpyAsync(g->h_recv.data(), g->d_recv[0], g->slot_bytes * g->ndev, hipMemcpyDeviceToHost, g->handles[0]->stream));
    rc = rn_group_sync(g);
    if (rc != RN_OK) return rc;
    const size_t C = static_cast<size_t>(g->num_classes);
    for (int d = 0; d < g->ndev; ++d) {
        int lo, cnt;
        shard(n, g->ndev, d, &lo, &cnt);
        const uint8_t* slot = g->h_recv.data() + g->slot_bytes * d;
        std::memcpy(probs + static_cast<size_t>(lo) * C, slot, static_cast<size_t>(cnt) * C * 4);
        std::memcpy(ids + lo, slot + static_cast<size_t>(g->cap) * C * 4, static_cast<size_t>(cnt) * 8);
    }
    return RN_OK;
}
