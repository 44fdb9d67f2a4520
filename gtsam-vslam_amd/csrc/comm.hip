// vslam_comm: the one collective of the hot path — an fp64 sum all-reduce of the partial reduced
// camera system [(6F)^2 + 6F doubles] and of the cost scalars, once per LM lambda trial.
// RCCL is loaded lazily (dlopen) so that libvslam_hip.so has no hard dependency on it: single-GPU
// users never touch it.  xGMI is point-to-point (7 links x ~153 GB/s); at <= 1.2 MB the all-reduce is
// latency-bound, so S | rhs travel as ONE buffer and the three cost scalars as one more.
#include "comm.hpp"
#include <dlfcn.h>

namespace {

struct NcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, char[128], int) = nullptr;   // ncclUniqueId is a 128-byte struct passed by value
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

struct Id128 { char b[128]; };

NcclApi& nccl_api() {
    static NcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = (int (*)(void*))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(void**, int, char[128], int))dlsym(api.lib, "ncclCommInitRank");
        api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(api.lib, "ncclAllReduce");
        api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
        api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy;
    });
    return api;
}

}  // namespace

namespace vslam {

vslam_status comm_allreduce(const vslam_comm* c, double* dbuf, size_t n, hipStream_t stream) {
    if (!c || n == 0) return VSLAM_OK;
    if (c->kind == 0) {
        NcclApi& api = nccl_api();
        const int r = api.AllReduce(dbuf, dbuf, n, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->nccl, stream);
        if (r != 0) { set_error("ncclAllReduce failed: %s", api.GetErrorString ? api.GetErrorString(r) : "?"); return VSLAM_ERR_COMM; }
        return VSLAM_OK;
    }
    if (c->kind == 2) {
        // caller-supplied transport: device -> host, the caller's all-reduce, host -> device (ordered on `stream`)
        std::vector<double>& st = const_cast<vslam_comm*>(c)->stage;
        st.resize(n);
        VS_HIP(hipMemcpyAsync(st.data(), dbuf, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        VS_HIP(hipStreamSynchronize(stream));
        const int r = c->cb(c->cbCtx, st.data(), n);
        if (r != 0) { set_error("caller-supplied all-reduce failed (status %d)", r); return VSLAM_ERR_COMM; }
        VS_HIP(hipMemcpyAsync(dbuf, st.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
        VS_HIP(hipStreamSynchronize(stream));
        return VSLAM_OK;
    }
    // local transport: device -> slot, barrier, fixed-order sum, barrier, -> device.
    // Error handling: a rank that fails never leaves early - it raises the group's `failed` flag and still passes BOTH
    // barriers, so its peers are not stranded; every rank then returns VSLAM_ERR_COMM.  (RCCL transport: a failure on
    // one rank leaves the others inside ncclAllReduce; the caller has to abort the job, as with any NCCL program.)
    vslam_local_group* g = c->grp.get();
    std::vector<double>& mine = g->slots[c->rank];
    mine.resize(n);
    bool ok = hipMemcpyAsync(mine.data(), dbuf, n * sizeof(double), hipMemcpyDeviceToHost, stream) == hipSuccess &&
              hipStreamSynchronize(stream) == hipSuccess;
    if (!ok) { set_error("local all-reduce: device -> host copy failed"); g->failed.store(1); }
    g->barrier();
    std::vector<double> sum(n, 0.0);
    for (int r = 0; r < c->world; r++) {
        const std::vector<double>& s = g->slots[r];
        if (s.size() != n) { set_error("local all-reduce: size mismatch between ranks"); g->failed.store(1); break; }
        for (size_t i = 0; i < n; i++) sum[i] += s[i];
    }
    g->barrier();
    if (g->failed.load()) return VSLAM_ERR_COMM;
    VS_HIP(hipMemcpyAsync(dbuf, sum.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
    VS_HIP(hipStreamSynchronize(stream));
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" {

vslam_status vslam_comm_unique_id(uint8_t id_out[128]) {
    if (!id_out) return VSLAM_ERR_INVALID;
    NcclApi& api = nccl_api();
    if (!api.ok) { set_error("RCCL (librccl.so) could not be loaded"); return VSLAM_ERR_COMM; }
    Id128 id;
    const int r = api.GetUniqueId(&id);
    if (r != 0) { set_error("ncclGetUniqueId failed"); return VSLAM_ERR_COMM; }
    memcpy(id_out, id.b, 128);
    return VSLAM_OK;
}

vslam_status vslam_comm_create_rccl(const uint8_t id[128], int32_t rank, int32_t world, int32_t device, vslam_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return VSLAM_ERR_INVALID;
    *out = nullptr;
    NcclApi& api = nccl_api();
    if (!api.ok) { set_error("RCCL (librccl.so) could not be loaded"); return VSLAM_ERR_COMM; }
    VS_HIP(hipSetDevice(device));
    vslam_comm* c = new vslam_comm();
    c->rank = rank; c->world = world; c->device = device; c->kind = 0;
    Id128 uid;
    memcpy(uid.b, id, 128);
    // ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId /*by value*/, int rank)
    typedef int (*init_fn)(void**, int, Id128, int);
    const int r = ((init_fn)api.CommInitRank)(&c->nccl, world, uid, rank);
    if (r != 0) { set_error("ncclCommInitRank failed: %s", api.GetErrorString ? api.GetErrorString(r) : "?"); delete c; return VSLAM_ERR_COMM; }
    *out = c;
    return VSLAM_OK;
}

vslam_status vslam_comm_create_local(int32_t world, vslam_comm** out_array) {
    if (world < 1 || !out_array) return VSLAM_ERR_INVALID;
    auto grp = std::make_shared<vslam_local_group>();
    grp->world = world;
    grp->slots.resize(world);
    for (int r = 0; r < world; r++) {
        vslam_comm* c = new vslam_comm();
        c->rank = r; c->world = world; c->kind = 1; c->grp = grp;
        out_array[r] = c;
    }
    return VSLAM_OK;
}

vslam_status vslam_comm_create_callback(int32_t rank, int32_t world, int32_t device, vslam_allreduce_fn allreduce, void* ctx, vslam_comm** out) {
    if (!out || !allreduce || world < 1 || rank < 0 || rank >= world) return VSLAM_ERR_INVALID;
    vslam_comm* c = new vslam_comm();
    c->rank = rank; c->world = world; c->device = device; c->kind = 2; c->cb = allreduce; c->cbCtx = ctx;
    *out = c;
    return VSLAM_OK;
}

void vslam_comm_destroy(vslam_comm* c) {
    if (!c) return;
    if (c->kind == 0 && c->nccl) nccl_api().CommDestroy(c->nccl);
    delete c;
}

}  // extern "C"
