// Multi-GPU layer of the C ABI (include/ismpc_group.h): contiguous shards of a batch of independent gait instances, one
// handle + launch stream + side stream per device, and the path's one collective -- the all-gather of the 80-byte output
// records -- on RCCL.  Host C++ only (no kernel lives here): the kernels are the single-device entry points of
// include/ismpc.h / include/ismpc_a.h, called once per local device.
//
// Reference: Controller.cpp:105-106 constructs ONE MPCSolver and :346-348 calls it from one thread; there is no
// multi-device code in the reference to follow.  Layout per device (HBM):
//     d_all[2]   gathered output records of a step (batch x 80 B), two buffers: the kernel of step k writes its shard at
//                offset first_r x 80 of buffer k & 1, the all-gather fills in the other shards IN PLACE
//     d_in       this device's shard of host-given input records (host entry points only)
// Stream order per device and buffer b:   launch stream: wait gathered[b] -> kernel -> record computed[b]
//                                         side stream:   wait computed[b] -> ncclAllGather -> record gathered[b]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ismpc_group.h"

namespace {

thread_local std::string g_gerr = "";
int gfail(int code, const std::string& msg) { g_gerr = msg; return code; }
#define G_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return gfail(ISMPC_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// ---- RCCL, bound at run time ----------------------------------------------------------------------------------------
// A process that already maps a copy of RCCL (a torch process maps torch/lib/librccl.so, which has no SONAME) must not get a
// second one: the copy found in /proc/self/maps is opened by its path; otherwise $ISMPC_RCCL_LIB, then librccl.so.1.
struct Rccl {
    void* lib = nullptr; std::string err, path;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl g_rccl; std::once_flag g_rccl_once;

std::string mapped_rccl()
{
    FILE* f = std::fopen("/proc/self/maps", "r");
    if (!f) return "";
    char line[4096]; std::string found;
    while (std::fgets(line, sizeof line, f)) {
        const char* p = std::strstr(line, "librccl.so");
        if (!p) continue;
        const char* s = std::strchr(line, '/');
        if (!s) continue;
        found.assign(s); while (!found.empty() && (found.back() == '\n' || found.back() == ' ')) found.pop_back();
        break;
    }
    std::fclose(f);
    return found;
}

void load_rccl()
{
    Rccl& r = g_rccl;
    std::vector<std::string> tries;
    const std::string m = mapped_rccl();
    if (!m.empty()) tries.push_back(m);
    if (const char* e = std::getenv("ISMPC_RCCL_LIB")) tries.push_back(e);
    tries.push_back("librccl.so.1"); tries.push_back("/opt/rocm/lib/librccl.so.1"); tries.push_back("librccl.so");
    for (const std::string& t : tries) {
        r.lib = dlopen(t.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.lib) { r.path = t; break; }
        r.err += t + ": " + (dlerror() ? dlerror() : "?") + "; ";
    }
    if (!r.lib) return;
#define BIND(name) r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.lib, "nccl" #name)); if (!r.name) { r.err = "nccl" #name " missing in " + r.path; r.lib = nullptr; return; }
    BIND(GetUniqueId) BIND(CommInitRank) BIND(CommInitAll) BIND(CommDestroy) BIND(CommCount) BIND(AllGather) BIND(Broadcast)
    BIND(GroupStart) BIND(GroupEnd) BIND(GetErrorString) BIND(GetVersion)
#undef BIND
}
Rccl* rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.lib ? &g_rccl : nullptr;
}
#define G_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    return gfail(ISMPC_E_NO_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)

struct DevGuard {
    int prev = -1, dev; hipError_t err = hipSuccess;
    explicit DevGuard(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) err = hipSetDevice(dev); }
    ~DevGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

void shard(int batch, int rank, int world, int* first, int* count)
{
    const int base = batch / world, extra = batch % world;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + (rank < extra ? rank : extra);
}

// ---- what both formulations share: devices, streams, events, the gathered buffers, the communicator -------------------
struct Local {
    int device = 0, rank = 0;
    hipStream_t launch = nullptr, side = nullptr;
    hipEvent_t computed[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr}, after = nullptr;
    bool gathered_set[2] = {false, false};
    ncclComm_t comm = nullptr;
    unsigned char* d_all[2] = {nullptr, nullptr}; int all_cap = 0;     // gathered output records
    void* d_in[4] = {nullptr, nullptr, nullptr, nullptr}; size_t in_cap[4] = {0, 0, 0, 0};   // host entry points: the shard's inputs
};

struct Core {
    std::vector<Local> loc;
    int world = 0;              // as RCCL reports it
    size_t rec = 80;            // bytes of an output record (ismpc_tick_out and ismpc_a_out are both 80)

    int init_streams()
    {
        for (Local& l : loc) {
            DevGuard gd(l.device); G_HIP(gd.err);
            G_HIP(hipStreamCreateWithFlags(&l.launch, hipStreamNonBlocking));
            G_HIP(hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking));
            for (int b = 0; b < 2; ++b) {
                G_HIP(hipEventCreateWithFlags(&l.computed[b], hipEventDisableTiming));
                G_HIP(hipEventCreateWithFlags(&l.gathered[b], hipEventDisableTiming));
            }
            G_HIP(hipEventCreateWithFlags(&l.after, hipEventDisableTiming));
        }
        return ISMPC_OK;
    }
    int init_comm_all()
    {
        Rccl* r = rccl();
        if (!r) return gfail(ISMPC_E_NO_DEVICE, "RCCL could not be loaded: " + g_rccl.err);
        std::vector<int> devs; for (Local& l : loc) devs.push_back(l.device);
        std::vector<ncclComm_t> comms(loc.size(), nullptr);
        G_NCCL(r->CommInitAll(comms.data(), (int)loc.size(), devs.data()));
        for (size_t k = 0; k < loc.size(); ++k) { loc[k].comm = comms[k]; loc[k].rank = (int)k; }
        return count_world((int)loc.size());
    }
    int init_comm_rank(const void* id128, int rank, int nranks)
    {
        Rccl* r = rccl();
        if (!r) return gfail(ISMPC_E_NO_DEVICE, "RCCL could not be loaded: " + g_rccl.err);
        static_assert(sizeof(ncclUniqueId) == ISMPC_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
        ncclUniqueId id; std::memcpy(&id, id128, sizeof id);
        DevGuard gd(loc[0].device); G_HIP(gd.err);
        G_NCCL(r->CommInitRank(&loc[0].comm, nranks, id, rank));
        loc[0].rank = rank;
        return count_world(nranks);
    }
    int count_world(int expected)
    {
        int c = 0;
        G_NCCL(g_rccl.CommCount(loc[0].comm, &c));
        if (c != expected) return gfail(ISMPC_E_NO_DEVICE, "RCCL reports " + std::to_string(c) + " ranks, expected " + std::to_string(expected));
        world = c;
        return ISMPC_OK;
    }
    int sync()
    {
        for (Local& l : loc) {
            DevGuard gd(l.device); G_HIP(gd.err);
            G_HIP(hipStreamSynchronize(l.launch)); G_HIP(hipStreamSynchronize(l.side));
        }
        return ISMPC_OK;
    }
    int reserve_all(int batch)
    {
        for (Local& l : loc) {
            if (batch <= l.all_cap) continue;
            DevGuard gd(l.device); G_HIP(gd.err);
            G_HIP(hipStreamSynchronize(l.launch)); G_HIP(hipStreamSynchronize(l.side));
            for (int b = 0; b < 2; ++b) {
                if (l.d_all[b]) G_HIP(hipFree(l.d_all[b]));
                l.d_all[b] = nullptr;
                G_HIP(hipMalloc((void**)&l.d_all[b], rec * (size_t)batch));
                l.gathered_set[b] = false;
            }
            l.all_cap = batch;
        }
        return ISMPC_OK;
    }
    int reserve_in(Local& l, int slot, size_t bytes)
    {
        if (bytes <= l.in_cap[slot]) return ISMPC_OK;
        G_HIP(hipStreamSynchronize(l.launch));
        if (l.d_in[slot]) G_HIP(hipFree(l.d_in[slot]));
        l.d_in[slot] = nullptr; l.in_cap[slot] = 0;
        G_HIP(hipMalloc(&l.d_in[slot], bytes));
        l.in_cap[slot] = bytes;
        return ISMPC_OK;
    }
    // before the kernel of a step writes buffer b: the collective that last read it is done (stream-side)
    int before_launch(Local& l, int b)
    {
        if (l.gathered_set[b]) G_HIP(hipStreamWaitEvent(l.launch, l.gathered[b], 0));
        return ISMPC_OK;
    }
    // after every local kernel of the step is enqueued: the one collective, on the side streams
    int gather(int batch, int b)
    {
        for (Local& l : loc) {
            DevGuard gd(l.device); G_HIP(gd.err);
            G_HIP(hipEventRecord(l.computed[b], l.launch));
            G_HIP(hipStreamWaitEvent(l.side, l.computed[b], 0));
        }
        Rccl& r = g_rccl;
        // (ISMPC_GROUP_FORCE_RAGGED=1: the all-gather-v form for every batch -- how a one-GPU box tests it)
        static const bool force_ragged = std::getenv("ISMPC_GROUP_FORCE_RAGGED") != nullptr;
        const bool equal = batch % world == 0 && !force_ragged;
        G_NCCL(r.GroupStart());
        for (Local& l : loc) {
            int first, count; shard(batch, l.rank, world, &first, &count);
            if (equal) {
                // in place: this rank's block already sits at recvbuff + rank * sendcount
                ncclResult_t e = r.AllGather(l.d_all[b] + rec * (size_t)first, l.d_all[b], rec * (size_t)count, ncclUint8, l.comm, l.side);
                if (e != ncclSuccess) { (void)r.GroupEnd(); return gfail(ISMPC_E_NO_DEVICE, std::string("ncclAllGather: ") + r.GetErrorString(e)); }
            } else {
                // ragged shards: all-gather-v as one fused group of in-place broadcasts, root = owner of the block
                for (int root = 0; root < world; ++root) {
                    int f2, c2; shard(batch, root, world, &f2, &c2);
                    if (c2 == 0) continue;
                    unsigned char* blk = l.d_all[b] + rec * (size_t)f2;
                    ncclResult_t e = r.Broadcast(blk, blk, rec * (size_t)c2, ncclUint8, root, l.comm, l.side);
                    if (e != ncclSuccess) { (void)r.GroupEnd(); return gfail(ISMPC_E_NO_DEVICE, std::string("ncclBroadcast: ") + r.GetErrorString(e)); }
                }
            }
        }
        G_NCCL(r.GroupEnd());
        for (Local& l : loc) {
            DevGuard gd(l.device); G_HIP(gd.err);
            G_HIP(hipEventRecord(l.gathered[b], l.side));
            l.gathered_set[b] = true;
        }
        return ISMPC_OK;
    }
    // the launch stream of local device k will not start later work before everything enqueued so far on `stream` is done
    int order_after(int k, void* stream)
    {
        if (k < 0 || k >= (int)loc.size()) return gfail(ISMPC_E_INVALID, "bad argument");
        Local& l = loc[k];
        DevGuard gd(l.device); G_HIP(gd.err);
        G_HIP(hipEventRecord(l.after, static_cast<hipStream_t>(stream)));
        G_HIP(hipStreamWaitEvent(l.launch, l.after, 0));
        return ISMPC_OK;
    }
    void destroy()
    {
        for (Local& l : loc) {
            DevGuard gd(l.device);
            if (l.launch) (void)hipStreamSynchronize(l.launch);
            if (l.side) (void)hipStreamSynchronize(l.side);
            if (l.comm && g_rccl.lib) (void)g_rccl.CommDestroy(l.comm);
            for (int b = 0; b < 2; ++b) {
                if (l.d_all[b]) (void)hipFree(l.d_all[b]);
                if (l.computed[b]) (void)hipEventDestroy(l.computed[b]);
                if (l.gathered[b]) (void)hipEventDestroy(l.gathered[b]);
            }
            for (int k = 0; k < 4; ++k) if (l.d_in[k]) (void)hipFree(l.d_in[k]);
            if (l.after) (void)hipEventDestroy(l.after);
            if (l.launch) (void)hipStreamDestroy(l.launch);
            if (l.side) (void)hipStreamDestroy(l.side);
        }
        loc.clear();
    }
};

int check_devices(const int* devices, int n)
{
    if (!devices || n < 1) return gfail(ISMPC_E_INVALID, "a group needs at least one device");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { (void)hipGetLastError(); return gfail(ISMPC_E_NO_DEVICE, "no HIP device: the ISMPC hot path has no CPU fallback"); }
    for (int k = 0; k < n; ++k) {
        if (devices[k] < 0 || devices[k] >= have) return gfail(ISMPC_E_INVALID, "device ordinal " + std::to_string(devices[k]) + " outside [0, " + std::to_string(have) + ")");
        for (int j = 0; j < k; ++j) if (devices[j] == devices[k]) return gfail(ISMPC_E_INVALID, "a device appears twice in the group (RCCL needs one rank per GPU)");
    }
    return ISMPC_OK;
}

}  // namespace

struct ismpc_group   { Core core; std::vector<ismpc_handle*> h; };
struct ismpc_a_group { Core core; std::vector<ismpc_a_handle*> h; };

extern "C" {

const char* ismpc_group_last_error(void) { return g_gerr.c_str(); }

int ismpc_group_rccl_version(void)
{
    Rccl* r = rccl(); int v = 0;
    if (!r || r->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int ismpc_shard_range(int batch, int rank, int world, int* first, int* count)
{
    if (batch < 0 || world < 1 || rank < 0 || rank >= world || !first || !count) return gfail(ISMPC_E_INVALID, "bad argument");
    shard(batch, rank, world, first, count);
    return ISMPC_OK;
}

int ismpc_group_unique_id(void* id128)
{
    if (!id128) return gfail(ISMPC_E_INVALID, "null argument");
    Rccl* r = rccl();
    if (!r) return gfail(ISMPC_E_NO_DEVICE, "RCCL could not be loaded: " + g_rccl.err);
    ncclUniqueId id;
    G_NCCL(r->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return ISMPC_OK;
}

// ======================================================================================================================
// Formulation B
// ======================================================================================================================
static int b_build(ismpc_group* g, const ismpc_params* params, const double* ftsp, int rows)
{
    for (Local& l : g->core.loc) {
        ismpc_handle* h = nullptr;
        const int rc = ismpc_create(params, ftsp, rows, l.device, &h);
        if (rc != ISMPC_OK) return gfail(rc, std::string("ismpc_create on device ") + std::to_string(l.device) + ": " + ismpc_last_error());
        g->h.push_back(h);
    }
    return g->core.init_streams();
}

void ismpc_group_destroy(ismpc_group* g)
{
    if (!g) return;
    g->core.destroy();
    for (ismpc_handle* h : g->h) ismpc_destroy(h);
    delete g;
}

int ismpc_group_create(const ismpc_params* params, const double* ftsp, int rows, const int* devices, int n, ismpc_group** out)
{
    if (!out) return gfail(ISMPC_E_INVALID, "null argument");
    *out = nullptr;
    int rc = check_devices(devices, n);
    if (rc != ISMPC_OK) return rc;
    ismpc_group* g = new ismpc_group();
    g->core.loc.resize(n);
    for (int k = 0; k < n; ++k) g->core.loc[k].device = devices[k];
    rc = b_build(g, params, ftsp, rows);
    if (rc == ISMPC_OK) rc = g->core.init_comm_all();
    if (rc != ISMPC_OK) { const std::string keep = g_gerr; ismpc_group_destroy(g); g_gerr = keep; return rc; }
    *out = g;
    return ISMPC_OK;
}

int ismpc_group_create_rank(const ismpc_params* params, const double* ftsp, int rows, int device, const void* id128, int rank, int world, ismpc_group** out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return gfail(ISMPC_E_INVALID, "bad argument");
    *out = nullptr;
    int rc = check_devices(&device, 1);
    if (rc != ISMPC_OK) return rc;
    ismpc_group* g = new ismpc_group();
    g->core.loc.resize(1);
    g->core.loc[0].device = device;
    rc = b_build(g, params, ftsp, rows);
    if (rc == ISMPC_OK) rc = g->core.init_comm_rank(id128, rank, world);
    if (rc != ISMPC_OK) { const std::string keep = g_gerr; ismpc_group_destroy(g); g_gerr = keep; return rc; }
    *out = g;
    return ISMPC_OK;
}

int ismpc_group_world(const ismpc_group* g) { return g ? g->core.world : ISMPC_E_INVALID; }
int ismpc_group_local(const ismpc_group* g) { return g ? (int)g->core.loc.size() : ISMPC_E_INVALID; }
int ismpc_group_rank(const ismpc_group* g, int local) { return (g && local >= 0 && local < (int)g->core.loc.size()) ? g->core.loc[local].rank : ISMPC_E_INVALID; }
ismpc_handle* ismpc_group_handle(ismpc_group* g, int local) { return (g && local >= 0 && local < (int)g->h.size()) ? g->h[local] : nullptr; }
int ismpc_group_sync(ismpc_group* g) { return g ? g->core.sync() : gfail(ISMPC_E_INVALID, "null group"); }
int ismpc_group_order_after(ismpc_group* g, int local, void* stream) { return g ? g->core.order_after(local, stream) : gfail(ISMPC_E_INVALID, "null group"); }

int ismpc_group_reserve(ismpc_group* g, int max_batch)
{
    if (!g || max_batch < 0) return gfail(ISMPC_E_INVALID, "bad argument");
    int rc = g->core.reserve_all(max_batch);
    for (size_t k = 0; rc == ISMPC_OK && k < g->h.size(); ++k) {
        int first, count; shard(max_batch, g->core.loc[k].rank, g->core.world, &first, &count);
        rc = ismpc_reserve(g->h[k], count + 1);
        if (rc != ISMPC_OK) return gfail(rc, ismpc_last_error());
    }
    return rc;
}

int ismpc_group_step_device(ismpc_group* g, int batch, const ismpc_tick_in* const* in_dev, int buf)
{
    if (!g || batch < 0 || (buf != 0 && buf != 1) || (batch > 0 && !in_dev)) return gfail(ISMPC_E_INVALID, "bad argument");
    if (batch == 0) return ISMPC_OK;
    Core& c = g->core;
    int rc = c.reserve_all(batch);
    if (rc != ISMPC_OK) return rc;
    for (size_t k = 0; k < c.loc.size(); ++k) {
        Local& l = c.loc[k];
        int first, count; shard(batch, l.rank, c.world, &first, &count);
        DevGuard gd(l.device); G_HIP(gd.err);
        rc = c.before_launch(l, buf);
        if (rc != ISMPC_OK) return rc;
        if (count > 0) {
            if (!in_dev[k]) return gfail(ISMPC_E_INVALID, "null shard pointer");
            rc = ismpc_solve_batch_device(g->h[k], count, in_dev[k], reinterpret_cast<ismpc_tick_out*>(l.d_all[buf]) + first, nullptr, l.launch);
            if (rc != ISMPC_OK) return gfail(rc, ismpc_last_error());
        }
    }
    return c.gather(batch, buf);
}

int ismpc_group_result_device(ismpc_group* g, int local, int buf, ismpc_tick_out** out_dev)
{
    if (!g || !out_dev || local < 0 || local >= (int)g->core.loc.size() || (buf != 0 && buf != 1)) return gfail(ISMPC_E_INVALID, "bad argument");
    *out_dev = reinterpret_cast<ismpc_tick_out*>(g->core.loc[local].d_all[buf]);
    return ISMPC_OK;
}

int ismpc_group_wait_on(ismpc_group* g, int local, int buf, void* stream)
{
    if (!g || local < 0 || local >= (int)g->core.loc.size() || (buf != 0 && buf != 1)) return gfail(ISMPC_E_INVALID, "bad argument");
    Local& l = g->core.loc[local];
    DevGuard gd(l.device); G_HIP(gd.err);
    if (l.gathered_set[buf]) G_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), l.gathered[buf], 0));
    return ISMPC_OK;
}

int ismpc_group_solve_batch(ismpc_group* g, int batch, const ismpc_tick_in* in_host, ismpc_tick_out* out_host)
{
    if (!g || batch < 0 || (batch > 0 && (!in_host || !out_host))) return gfail(ISMPC_E_INVALID, "bad argument");
    if (batch == 0) return ISMPC_OK;
    Core& c = g->core;
    int rc = c.reserve_all(batch);
    if (rc != ISMPC_OK) return rc;
    std::vector<const ismpc_tick_in*> shards(c.loc.size(), nullptr);
    for (size_t k = 0; k < c.loc.size(); ++k) {
        Local& l = c.loc[k];
        int first, count; shard(batch, l.rank, c.world, &first, &count);
        DevGuard gd(l.device); G_HIP(gd.err);
        rc = c.reserve_in(l, 0, sizeof(ismpc_tick_in) * (size_t)(count > 0 ? count : 1));
        if (rc != ISMPC_OK) return rc;
        if (count > 0) G_HIP(hipMemcpyAsync(l.d_in[0], in_host + first, sizeof(ismpc_tick_in) * (size_t)count, hipMemcpyHostToDevice, l.launch));
        shards[k] = static_cast<const ismpc_tick_in*>(l.d_in[0]);
    }
    rc = ismpc_group_step_device(g, batch, shards.data(), 0);
    if (rc != ISMPC_OK) return rc;
    {   // every device holds all records now; the caller's copy comes from the first local one
        Local& l = c.loc[0];
        DevGuard gd(l.device); G_HIP(gd.err);
        G_HIP(hipMemcpyAsync(out_host, l.d_all[0], sizeof(ismpc_tick_out) * (size_t)batch, hipMemcpyDeviceToHost, l.side));
    }
    return c.sync();
}

// ======================================================================================================================
// Formulation A
// ======================================================================================================================
static int a_build(ismpc_a_group* g, const ismpc_a_params* p, const double* center)
{
    for (Local& l : g->core.loc) {
        ismpc_a_handle* h = nullptr;
        const int rc = ismpc_a_create(p, center, l.device, &h);
        if (rc != ISMPC_OK) return gfail(rc, std::string("ismpc_a_create on device ") + std::to_string(l.device) + ": " + ismpc_a_last_error());
        g->h.push_back(h);
    }
    return g->core.init_streams();
}

void ismpc_a_group_destroy(ismpc_a_group* g)
{
    if (!g) return;
    g->core.destroy();
    for (ismpc_a_handle* h : g->h) ismpc_a_destroy(h);
    delete g;
}

int ismpc_a_group_create(const ismpc_a_params* p, const double* center, const int* devices, int n, ismpc_a_group** out)
{
    if (!out) return gfail(ISMPC_E_INVALID, "null argument");
    *out = nullptr;
    int rc = check_devices(devices, n);
    if (rc != ISMPC_OK) return rc;
    ismpc_a_group* g = new ismpc_a_group();
    g->core.loc.resize(n);
    for (int k = 0; k < n; ++k) g->core.loc[k].device = devices[k];
    rc = a_build(g, p, center);
    if (rc == ISMPC_OK) rc = g->core.init_comm_all();
    if (rc != ISMPC_OK) { const std::string keep = g_gerr; ismpc_a_group_destroy(g); g_gerr = keep; return rc; }
    *out = g;
    return ISMPC_OK;
}

int ismpc_a_group_create_rank(const ismpc_a_params* p, const double* center, int device, const void* id128, int rank, int world, ismpc_a_group** out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return gfail(ISMPC_E_INVALID, "bad argument");
    *out = nullptr;
    int rc = check_devices(&device, 1);
    if (rc != ISMPC_OK) return rc;
    ismpc_a_group* g = new ismpc_a_group();
    g->core.loc.resize(1);
    g->core.loc[0].device = device;
    rc = a_build(g, p, center);
    if (rc == ISMPC_OK) rc = g->core.init_comm_rank(id128, rank, world);
    if (rc != ISMPC_OK) { const std::string keep = g_gerr; ismpc_a_group_destroy(g); g_gerr = keep; return rc; }
    *out = g;
    return ISMPC_OK;
}

int ismpc_a_group_world(const ismpc_a_group* g) { return g ? g->core.world : ISMPC_E_INVALID; }
int ismpc_a_group_local(const ismpc_a_group* g) { return g ? (int)g->core.loc.size() : ISMPC_E_INVALID; }
int ismpc_a_group_rank(const ismpc_a_group* g, int local) { return (g && local >= 0 && local < (int)g->core.loc.size()) ? g->core.loc[local].rank : ISMPC_E_INVALID; }
ismpc_a_handle* ismpc_a_group_handle(ismpc_a_group* g, int local) { return (g && local >= 0 && local < (int)g->h.size()) ? g->h[local] : nullptr; }
int ismpc_a_group_sync(ismpc_a_group* g) { return g ? g->core.sync() : gfail(ISMPC_E_INVALID, "null group"); }
int ismpc_a_group_order_after(ismpc_a_group* g, int local, void* stream) { return g ? g->core.order_after(local, stream) : gfail(ISMPC_E_INVALID, "null group"); }

int ismpc_a_group_add_plan(ismpc_a_group* g, const double* center)
{
    if (!g || !center) return gfail(ISMPC_E_INVALID, "bad argument");
    int idx = -1;
    for (ismpc_a_handle* h : g->h) {
        const int r = ismpc_a_add_plan(h, center);
        if (r < 0) return gfail(r, ismpc_a_last_error());
        if (idx >= 0 && r != idx) return gfail(ISMPC_E_INVALID, "plan indices differ between devices");
        idx = r;
    }
    return idx;
}

int ismpc_a_group_set_precision(ismpc_a_group* g, int fp32)
{
    if (!g) return gfail(ISMPC_E_INVALID, "null group");
    for (ismpc_a_handle* h : g->h) {
        const int r = ismpc_a_set_precision(h, fp32);
        if (r != ISMPC_OK) return gfail(r, ismpc_a_last_error());
    }
    return ISMPC_OK;
}

int ismpc_a_group_reserve(ismpc_a_group* g, int max_batch)
{
    if (!g || max_batch < 0) return gfail(ISMPC_E_INVALID, "bad argument");
    int rc = g->core.reserve_all(max_batch);
    for (size_t k = 0; rc == ISMPC_OK && k < g->h.size(); ++k) {
        int first, count; shard(max_batch, g->core.loc[k].rank, g->core.world, &first, &count);
        rc = ismpc_a_reserve(g->h[k], count + 1);
        if (rc != ISMPC_OK) return gfail(rc, ismpc_a_last_error());
    }
    return rc;
}

int ismpc_a_group_step_device(ismpc_a_group* g, int batch, ismpc_a_state* const* state_dev, const ismpc_a_inst* const* inst_dev,
                              const double* const* push_dev, int buf)
{
    if (!g || batch < 0 || (buf != 0 && buf != 1) || (batch > 0 && !state_dev)) return gfail(ISMPC_E_INVALID, "bad argument");
    if (batch == 0) return ISMPC_OK;
    Core& c = g->core;
    int rc = c.reserve_all(batch);
    if (rc != ISMPC_OK) return rc;
    for (size_t k = 0; k < c.loc.size(); ++k) {
        Local& l = c.loc[k];
        int first, count; shard(batch, l.rank, c.world, &first, &count);
        DevGuard gd(l.device); G_HIP(gd.err);
        rc = c.before_launch(l, buf);
        if (rc != ISMPC_OK) return rc;
        if (count > 0) {
            if (!state_dev[k]) return gfail(ISMPC_E_INVALID, "null shard pointer");
            ismpc_a_out* dst = reinterpret_cast<ismpc_a_out*>(l.d_all[buf]) + first;
            const double* push = push_dev ? push_dev[k] : nullptr;
            const ismpc_a_inst* inst = inst_dev ? inst_dev[k] : nullptr;
            rc = inst ? ismpc_a_tick_batch_inst_device(g->h[k], count, state_dev[k], inst, push, dst, l.launch)
                      : ismpc_a_tick_batch_device(g->h[k], count, state_dev[k], push, dst, l.launch);
            if (rc != ISMPC_OK) return gfail(rc, ismpc_a_last_error());
        }
    }
    return c.gather(batch, buf);
}

int ismpc_a_group_result_device(ismpc_a_group* g, int local, int buf, ismpc_a_out** out_dev)
{
    if (!g || !out_dev || local < 0 || local >= (int)g->core.loc.size() || (buf != 0 && buf != 1)) return gfail(ISMPC_E_INVALID, "bad argument");
    *out_dev = reinterpret_cast<ismpc_a_out*>(g->core.loc[local].d_all[buf]);
    return ISMPC_OK;
}

int ismpc_a_group_wait_on(ismpc_a_group* g, int local, int buf, void* stream)
{
    if (!g || local < 0 || local >= (int)g->core.loc.size() || (buf != 0 && buf != 1)) return gfail(ISMPC_E_INVALID, "bad argument");
    Local& l = g->core.loc[local];
    DevGuard gd(l.device); G_HIP(gd.err);
    if (l.gathered_set[buf]) G_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), l.gathered[buf], 0));
    return ISMPC_OK;
}

int ismpc_a_group_tick_batch(ismpc_a_group* g, int batch, ismpc_a_state* state_host, const ismpc_a_inst* inst_host,
                             const double* push_host, ismpc_a_out* out_host)
{
    if (!g || batch < 0 || (batch > 0 && (!state_host || !out_host))) return gfail(ISMPC_E_INVALID, "bad argument");
    if (batch == 0) return ISMPC_OK;
    Core& c = g->core;
    int rc = c.reserve_all(batch);
    if (rc != ISMPC_OK) return rc;
    const size_t n = c.loc.size();
    std::vector<ismpc_a_state*> st(n, nullptr); std::vector<const ismpc_a_inst*> in(n, nullptr); std::vector<const double*> pu(n, nullptr);
    for (size_t k = 0; k < n; ++k) {
        Local& l = c.loc[k];
        int first, count; shard(batch, l.rank, c.world, &first, &count);
        const size_t cn = (size_t)(count > 0 ? count : 1);
        DevGuard gd(l.device); G_HIP(gd.err);
        rc = c.reserve_in(l, 0, sizeof(ismpc_a_state) * cn);
        if (rc == ISMPC_OK && inst_host) rc = c.reserve_in(l, 1, sizeof(ismpc_a_inst) * cn);
        if (rc == ISMPC_OK && push_host) rc = c.reserve_in(l, 2, 2 * sizeof(double) * cn);
        if (rc != ISMPC_OK) return rc;
        if (count > 0) {
            G_HIP(hipMemcpyAsync(l.d_in[0], state_host + first, sizeof(ismpc_a_state) * (size_t)count, hipMemcpyHostToDevice, l.launch));
            if (inst_host) G_HIP(hipMemcpyAsync(l.d_in[1], inst_host + first, sizeof(ismpc_a_inst) * (size_t)count, hipMemcpyHostToDevice, l.launch));
            if (push_host) G_HIP(hipMemcpyAsync(l.d_in[2], push_host + 2 * (size_t)first, 2 * sizeof(double) * (size_t)count, hipMemcpyHostToDevice, l.launch));
        }
        st[k] = static_cast<ismpc_a_state*>(l.d_in[0]);
        in[k] = inst_host ? static_cast<const ismpc_a_inst*>(l.d_in[1]) : nullptr;
        pu[k] = push_host ? static_cast<const double*>(l.d_in[2]) : nullptr;
    }
    rc = ismpc_a_group_step_device(g, batch, st.data(), in.data(), pu.data(), 0);
    if (rc != ISMPC_OK) return rc;
    for (size_t k = 0; k < n; ++k) {          // the advanced state of this process's shards goes back in place
        Local& l = c.loc[k];
        int first, count; shard(batch, l.rank, c.world, &first, &count);
        DevGuard gd(l.device); G_HIP(gd.err);
        if (count > 0) G_HIP(hipMemcpyAsync(state_host + first, l.d_in[0], sizeof(ismpc_a_state) * (size_t)count, hipMemcpyDeviceToHost, l.launch));
    }
    {
        Local& l = c.loc[0];
        DevGuard gd(l.device); G_HIP(gd.err);
        G_HIP(hipMemcpyAsync(out_host, l.d_all[0], sizeof(ismpc_a_out) * (size_t)batch, hipMemcpyDeviceToHost, l.side));
    }
    return c.sync();
}

}  // extern "C"
