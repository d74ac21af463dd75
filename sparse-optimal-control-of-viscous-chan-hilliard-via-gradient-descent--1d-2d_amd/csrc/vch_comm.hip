// vch_comm.hip — the one collective of the path behind the C ABI (include/vch.h "multi-GPU"):
// per PGD iteration ONE all-reduce (sum) of the cost scalars {J1, J2, J3, J4, J} over all trajectories of all
// ranks.  The per-trajectory scalars never leave the device on the way: every context keeps them in its J_dev
// buffer, a one-workgroup kernel adds them into the communicator's 5-entry device buffer, RCCL reduces that
// buffer in place over xGMI, and only the 5 global sums travel to the host.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library torch.distributed's "nccl" backend uses, so a
// process that also initialised torch shares one RCCL instance); libvch_hip.so itself has no link-time
// dependency on it and single-GPU users never load it.  The unique id is created on rank 0
// (vch_comm_unique_id) and handed to the other ranks by the caller over whatever channel it already has
// (bench.py: torch.distributed broadcast_object_list) -- the engine does no networking of its own.
#include <dlfcn.h>

// The few RCCL declarations the collective needs, stated here (they are part of NCCL's stable C ABI): the library is only
// ever dlopen'ed, so building the engine must not need the RCCL headers either.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;        // ncclFloat64
}

static const char *dl_why() {
    const char *e = dlerror();
    return e ? e : "unknown dlopen/dlsym failure";
}

struct vch_comm {
    int device, rank, world;
    void *dl;
    ncclComm_t comm;
    hipStream_t stream;
    hipEvent_t ev;
    double *sum_dev;          // [5]
    double *sum_host;         // pinned [5]
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    const char *(*GetErrorString)(ncclResult_t);
};

static void *rccl_open() {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return h;
}

extern "C" int vch_comm_unique_id(unsigned char *id_out) {
    if (!id_out) return vch_fail(VCH_ERR_ARG, "vch_comm_unique_id: NULL output");
    void *h = rccl_open();
    if (!h) return vch_fail(VCH_ERR_STATE, "vch_comm_unique_id: librccl.so.1 not loadable: %s", dl_why());
    auto get = (ncclResult_t(*)(ncclUniqueId *))dlsym(h, "ncclGetUniqueId");
    auto err = (const char *(*)(ncclResult_t))dlsym(h, "ncclGetErrorString");
    if (!get || !err) return vch_fail(VCH_ERR_STATE, "vch_comm_unique_id: RCCL symbols missing");
    ncclUniqueId id;
    ncclResult_t r = get(&id);
    if (r != ncclSuccess) return vch_fail(VCH_ERR_HIP, "ncclGetUniqueId: %s", err(r));
    static_assert(sizeof(id) == VCH_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" vch_comm *vch_comm_create(const unsigned char *id, int rank, int world, int device) {
    if (!id || world < 1 || rank < 0 || rank >= world) {
        vch_fail(VCH_ERR_ARG, "vch_comm_create: bad arguments (id, 0 <= rank < world)");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        vch_fail(VCH_ERR_HIP, "vch_comm_create: hipSetDevice(%d) failed", device);
        return nullptr;
    }
    vch_comm *m = new vch_comm();
    m->device = device; m->rank = rank; m->world = world;
    m->comm = nullptr; m->stream = nullptr; m->ev = nullptr; m->sum_dev = nullptr; m->sum_host = nullptr;
    m->dl = rccl_open();
    auto fail = [&](const char *what, const char *why) {
        vch_fail(VCH_ERR_HIP, "vch_comm_create: %s: %s", what, why);
        if (m->sum_dev) hipFree(m->sum_dev);
        if (m->sum_host) hipHostFree(m->sum_host);
        if (m->ev) hipEventDestroy(m->ev);
        if (m->stream) hipStreamDestroy(m->stream);
        delete m;
        return (vch_comm *)nullptr;
    };
    if (!m->dl) return fail("dlopen librccl.so.1", dl_why());
    auto init = (ncclResult_t(*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(m->dl, "ncclCommInitRank");
    m->AllReduce = (decltype(m->AllReduce))dlsym(m->dl, "ncclAllReduce");
    m->CommDestroy = (decltype(m->CommDestroy))dlsym(m->dl, "ncclCommDestroy");
    m->GetErrorString = (decltype(m->GetErrorString))dlsym(m->dl, "ncclGetErrorString");
    if (!init || !m->AllReduce || !m->CommDestroy || !m->GetErrorString) return fail("dlsym", "RCCL symbols missing");
    if (hipStreamCreate(&m->stream) != hipSuccess || hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&m->sum_dev, 5 * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void **)&m->sum_host, 5 * sizeof(double)) != hipSuccess)
        return fail("HIP resources", hipGetErrorString(hipGetLastError()));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = init(&m->comm, world, uid, rank);
    if (r != ncclSuccess) return fail("ncclCommInitRank", m->GetErrorString(r));
    return m;
}

extern "C" void vch_comm_destroy(vch_comm *m) {
    if (!m) return;
    hipSetDevice(m->device);
    hipStreamSynchronize(m->stream);
    if (m->comm) m->CommDestroy(m->comm);
    hipFree(m->sum_dev);
    hipHostFree(m->sum_host);
    hipEventDestroy(m->ev);
    hipStreamDestroy(m->stream);
    delete m;
}

// acc[0..5) += sum_b J[b][0..5)   (one workgroup; B is small)
__global__ void k_sum_costs(const double *__restrict__ J, int B, double *__restrict__ acc) {
    const int k = threadIdx.x;
    if (k >= 5) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += J[5 * b + k];
    acc[k] += s;
}

extern "C" int vch_comm_allreduce_cost(vch_comm *m, vch2d_ctx *const *ctxs, int nctx, long iteration, double *J_sum_out) {
    if (!m || !ctxs || nctx < 1 || !J_sum_out) return vch_fail(VCH_ERR_ARG, "vch_comm_allreduce_cost: NULL argument");
    // every argument is checked BEFORE anything is enqueued: a rank that returns an error here has started nothing, and
    // its peers are not left inside a collective this rank never joins (the caller ends the job, bench.py main)
    std::vector<const double *> src(nctx);
    for (int i = 0; i < nctx; ++i) {
        vch2d_ctx *c = ctxs[i];
        if (!c || !c->pgd_ready) return vch_fail(VCH_ERR_STATE, "vch_comm_allreduce_cost: context %d has no PGD problem loaded", i);
        if (c->device != m->device) return vch_fail(VCH_ERR_ARG, "vch_comm_allreduce_cost: context %d lives on another device", i);
        src[i] = c->J_dev;                               // iteration < 0: the current iterate
        if (iteration >= 0) {
            // the context's worker thread advances the counter while this (main) thread reads it
            const long done = c->pgd_iter_total.load(std::memory_order_acquire);
            if (iteration >= done || iteration < done - J_RING)
                return vch_fail(VCH_ERR_STATE, "vch_comm_allreduce_cost: iteration %ld of context %d is not in the ring (%ld done)",
                                iteration, i, done);
            src[i] = c->J_ring_dev + (size_t)(iteration % J_RING) * 5 * c->B;
        }
    }
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipMemsetAsync(m->sum_dev, 0, 5 * sizeof(double), m->stream));
    for (int i = 0; i < nctx; ++i) {
        vch2d_ctx *c = ctxs[i];
        // order after the context's own work (its J_dev upload), without blocking the host
        HIPCHK(hipEventRecord(m->ev, c->stream));
        HIPCHK(hipStreamWaitEvent(m->stream, m->ev, 0));
        hipLaunchKernelGGL(k_sum_costs, dim3(1), dim3(64), 0, m->stream, src[i], c->B, m->sum_dev);
        HIPCHK(hipGetLastError());
    }
    ncclResult_t r = m->AllReduce(m->sum_dev, m->sum_dev, 5, ncclDouble, ncclSum, m->comm, m->stream);
    if (r != ncclSuccess) return vch_fail(VCH_ERR_HIP, "ncclAllReduce: %s", m->GetErrorString(r));
    HIPCHK(hipMemcpyAsync(m->sum_host, m->sum_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    memcpy(J_sum_out, m->sum_host, 5 * sizeof(double));
    return 0;
}
