// rccl_stub.hip — TEST stand-in for the six RCCL entry points that rtmi_reduce_accum binds
// (ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclReduce, ncclGetErrorString), so that
// the several-contexts-in-one-process sum can run on a box with ONE GPU.  It is not part of the product:
// librtmi.so opens it only when the environment variable RTMI_RCCL_LIBRARY names it.
//
// A grouped reduce becomes: every rank's stream records an event; the root's stream waits for them, then
// one kernel adds the ranks' buffers in rank order (((b0 + b1) + b2) + ...) into the root's buffer; the
// other streams wait for that kernel before they go on (their buffers are read by it).  The fixed order
// makes the result comparable bit for bit with a host-side float sum in the same order.  Buffers of
// ranks on other GPUs are read through peer access.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;

struct World {
    int n = 0;
    std::vector<int> devices;
    int live = 0;
};

struct Pending {
    const float *send = nullptr;
    float *recv = nullptr;
    size_t count = 0;
    int root = -1;
    hipStream_t stream = nullptr;
    bool set = false;
};

struct Comm {
    World *world;
    int rank;
};

struct Ptrs {
    const float *p[kMaxRanks];
};

std::mutex g_lock;
int g_group_depth = 0;
std::vector<std::pair<Comm *, Pending>> g_ops;

__global__ void k_sum_in_rank_order(Ptrs src, int n, float *dst, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float acc = src.p[0][i];
    for (int r = 1; r < n; r++) acc = acc + src.p[r][i];
    dst[i] = acc;
}

ncclResult_t Flush()
{
    if (g_ops.empty()) return ncclSuccess;
    World *w = g_ops[0].first->world;
    if ((int)g_ops.size() != w->n || w->n > kMaxRanks) {
        g_ops.clear();
        return ncclInvalidUsage;
    }
    std::vector<Pending> by_rank((size_t)w->n);
    for (auto &op : g_ops) {
        if (op.first->world != w || by_rank[(size_t)op.first->rank].set) {
            g_ops.clear();
            return ncclInvalidUsage;
        }
        by_rank[(size_t)op.first->rank] = op.second;
    }
    g_ops.clear();
    const int root = by_rank[0].root;
    const size_t count = by_rank[0].count;
    for (const Pending &p : by_rank)
        if (p.root != root || p.count != count || root < 0 || root >= w->n) return ncclInvalidArgument;
    int saved = 0;
    (void)hipGetDevice(&saved);
    ncclResult_t out = ncclSuccess;
    std::vector<hipEvent_t> events;
    auto ok = [&](hipError_t e) {
        if (e != hipSuccess) out = ncclUnhandledCudaError;
        return e == hipSuccess;
    };
    // every other rank's stream -> event -> the root's stream
    for (int r = 0; r < w->n && out == ncclSuccess; r++) {
        if (r == root) continue;
        hipEvent_t ev;
        if (!ok(hipSetDevice(w->devices[(size_t)r])) || !ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming))) break;
        events.push_back(ev);
        if (!ok(hipEventRecord(ev, by_rank[(size_t)r].stream))) break;
        if (!ok(hipSetDevice(w->devices[(size_t)root])) || !ok(hipStreamWaitEvent(by_rank[(size_t)root].stream, ev, 0))) break;
    }
    if (out == ncclSuccess && ok(hipSetDevice(w->devices[(size_t)root]))) {
        for (int r = 0; r < w->n; r++)
            if (w->devices[(size_t)r] != w->devices[(size_t)root]) (void)hipDeviceEnablePeerAccess(w->devices[(size_t)r], 0);
        (void)hipGetLastError();   // "already enabled" is fine
        Ptrs src;
        for (int r = 0; r < kMaxRanks; r++) src.p[r] = r < w->n ? by_rank[(size_t)r].send : nullptr;
        const unsigned threads = 256;
        const unsigned blocks = (unsigned)((count + threads - 1) / threads);
        if (count) hipLaunchKernelGGL(k_sum_in_rank_order, dim3(blocks), dim3(threads), 0, by_rank[(size_t)root].stream, src, w->n,
                                      by_rank[(size_t)root].recv, count);
        ok(hipGetLastError());
        // the root's kernel -> event -> every other rank's stream
        hipEvent_t done;
        if (out == ncclSuccess && ok(hipEventCreateWithFlags(&done, hipEventDisableTiming))) {
            events.push_back(done);
            if (ok(hipEventRecord(done, by_rank[(size_t)root].stream))) {
                for (int r = 0; r < w->n && out == ncclSuccess; r++) {
                    if (r == root) continue;
                    if (!ok(hipSetDevice(w->devices[(size_t)r]))) break;
                    ok(hipStreamWaitEvent(by_rank[(size_t)r].stream, done, 0));
                }
            }
        }
    }
    // an event may be destroyed while work that refers to it is pending: its resources go when that work is done
    for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
    (void)hipSetDevice(saved);
    return out;
}

}  // namespace

extern "C" {

// what librtmi.so looks for to tell this stand-in from RCCL: contexts of one sum may share a GPU
__attribute__((visibility("default"))) int rtmi_rccl_stub_same_device = 1;

__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev <= 0 || ndev > kMaxRanks) return ncclInvalidArgument;
    World *w = new World;
    w->n = ndev;
    w->live = ndev;
    for (int i = 0; i < ndev; i++) w->devices.push_back(devlist ? devlist[i] : i);
    for (int i = 0; i < ndev; i++) comms[i] = reinterpret_cast<ncclComm_t>(new Comm{w, i});
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    std::lock_guard<std::mutex> g(g_lock);
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (--c->world->live == 0) delete c->world;
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart(void)
{
    std::lock_guard<std::mutex> g(g_lock);
    g_group_depth++;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd(void)
{
    std::lock_guard<std::mutex> g(g_lock);
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    return Flush();
}

__attribute__((visibility("default"))) ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype,
                                                                ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !sendbuff || datatype != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;
    std::lock_guard<std::mutex> g(g_lock);
    if (g_group_depth <= 0) return ncclInvalidUsage;   // the stand-in only knows the grouped form
    Pending p;
    p.send = static_cast<const float *>(sendbuff);
    p.recv = static_cast<float *>(recvbuff);
    p.count = count;
    p.root = root;
    p.stream = stream;
    p.set = true;
    g_ops.emplace_back(reinterpret_cast<Comm *>(comm), p);
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error (stub)";
    case ncclUnhandledCudaError: return "a HIP call failed (stub)";
    case ncclInvalidArgument: return "invalid argument (stub)";
    case ncclInvalidUsage: return "invalid usage (stub)";
    default: return "error (stub)";
    }
}

}  // extern "C"
