// tsdrgpu_rccl.hip — the one exchange step of the multi-GPU autocorrelation sweep: every rank (one
// process per GPU) keeps per-lag SUMS of |R| over the capture windows it owns (tsdrgpu_autocorr_run mode 1); one
// ncclAllReduce(ncclDouble, ncclSum) of frame_len + line_len doubles (5.4 MB at 100 MS/s) over xGMI, queued on the
// autocorrelation's own lane, then the division by the global window count, give every rank the plots the reference
// forms as a running mean (accummulate, frameratedetector.c:51-60).
//
// RCCL is resolved with dlopen at first use (librccl.so.1 of the ROCm install), so the library itself has no link-time
// dependency on it and single-GPU hosts never load it.
#include <dlfcn.h>

#include "tsdrgpu_internal.h"

// the slice of rccl.h this file needs (ABI of RCCL 2.x as shipped with ROCm 6/7)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess_ = 0 };
enum { ncclFloat64_ = 8 };  // ncclDataType_t: ncclDouble
enum { ncclSum_ = 0 };      // ncclRedOp_t

struct RcclApi {
    void *dl;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
};
static RcclApi g_rccl;

static const RcclApi *rccl_load();
// resolved once, also when several host threads (one per device, host/tsdr_sweep.c) come here at the same time
static const RcclApi *rccl()
{
    static const RcclApi *const api = rccl_load();
    return api;
}
static const RcclApi *rccl_load()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *dl = nullptr;
    for (const char *n : names)
        if ((dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!dl) return nullptr;
    RcclApi a;
    a.dl = dl;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(dl, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(dl, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(dl, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(dl, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(dl, "ncclGetErrorString");
    a.Broadcast = (decltype(a.Broadcast))dlsym(dl, "ncclBroadcast");
    a.AllGather = (decltype(a.AllGather))dlsym(dl, "ncclAllGather");
    a.CommCount = (decltype(a.CommCount))dlsym(dl, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(dl, "ncclCommUserRank");
    if (!a.Broadcast || !a.AllGather) { dlclose(dl); return nullptr; }
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        dlclose(dl);
        return nullptr;
    }
    g_rccl = a;
    return &g_rccl;
}

struct tsdrgpu_comm {
    tsdrgpu_t *g;
    ncclComm_t comm;
    int world, rank;
};

extern "C" int tsdrgpu_rccl_unique_id(void *id128)
{
    if (!id128) return TSDRGPU_EINVAL;
    const RcclApi *r = rccl();
    if (!r) return TSDRGPU_ESTATE;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess_) return TSDRGPU_EHIP;
    memcpy(id128, id.internal, sizeof(id.internal));
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_comm_create(tsdrgpu_t *g, tsdrgpu_comm_t **out, int world, int rank, const void *id128)
{
    if (!g || !out || !id128 || world < 1 || rank < 0 || rank >= world) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_comm_create", "bad argument") : TSDRGPU_EINVAL;
    const RcclApi *r = rccl();
    if (!r) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_comm_create", "librccl.so.1 not found");
    HIP_TRY(g, hipSetDevice(g->device));
    tsdrgpu_comm_t *c = (tsdrgpu_comm_t *)calloc(1, sizeof(*c));
    if (!c) return TSDRGPU_ENOMEM;
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    const ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess_) {
        free(c);
        return tsdr_fail(g, TSDRGPU_EHIP, "ncclCommInitRank", r->GetErrorString(rc));
    }
    c->g = g;
    c->world = world;
    c->rank = rank;
    *out = c;
    return TSDRGPU_OK;
}

// What RCCL ITSELF says about the communicator (ncclCommCount / ncclCommUserRank), not what the caller passed to
// tsdrgpu_comm_create: the record of a multi-GPU run quotes these, so that "RCCL saw N ranks" is checkable.
extern "C" int tsdrgpu_comm_count(tsdrgpu_comm_t *c, int *ranks, int *my_rank)
{
    if (!c) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = c->g;
    const RcclApi *r = rccl();
    if (!r || !r->CommCount || !r->CommUserRank) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_comm_count", "ncclCommCount not in librccl");
    int n = 0, me = -1;
    ncclResult_t rc = r->CommCount(c->comm, &n);
    if (rc == ncclSuccess_) rc = r->CommUserRank(c->comm, &me);
    if (rc != ncclSuccess_) return tsdr_fail(g, TSDRGPU_EHIP, "ncclCommCount", r->GetErrorString(rc));
    if (ranks) *ranks = n;
    if (my_rank) *my_rank = me;
    return TSDRGPU_OK;
}

extern "C" void tsdrgpu_comm_destroy(tsdrgpu_comm_t *c)
{
    if (!c) return;
    const RcclApi *r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    free(c);
}

// all-reduce of a caller buffer of doubles on a lane of the context (building block; also what the frame path's
// row-band exchange uses)
extern "C" int tsdrgpu_comm_allreduce_f64(tsdrgpu_comm_t *c, double *d_buf, int64_t count, int lane)
{
    if (!c || !d_buf || count < 0) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = c->g;
    const RcclApi *r = rccl();
    if (!r) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_comm_allreduce_f64", "librccl.so.1 not found");
    hipStream_t st = tsdr_lane_stream(g, lane);
    if (!st) return tsdr_fail(g, TSDRGPU_EINVAL, "allreduce", "bad lane");
    const ncclResult_t rc = r->AllReduce(d_buf, d_buf, (size_t)count, ncclFloat64_, ncclSum_, c->comm, st);
    if (rc != ncclSuccess_) return tsdr_fail(g, TSDRGPU_EHIP, "ncclAllReduce", r->GetErrorString(rc));
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_comm_allreduce_f32max(tsdrgpu_comm_t *c, float *d_buf, int64_t count, int lane)
{
    if (!c || !d_buf || count < 0) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = c->g;
    const RcclApi *r = rccl();
    if (!r) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_comm_allreduce_f32max", "librccl.so.1 not found");
    hipStream_t st = tsdr_lane_stream(g, lane);
    if (!st) return tsdr_fail(g, TSDRGPU_EINVAL, "allreduce", "bad lane");
    const ncclResult_t rc = r->AllReduce(d_buf, d_buf, (size_t)count, 7 /* ncclFloat32 */, 2 /* ncclMax */, c->comm, st);
    if (rc != ncclSuccess_) return tsdr_fail(g, TSDRGPU_EHIP, "ncclAllReduce", r->GetErrorString(rc));
    return TSDRGPU_OK;
}

// The two exchanges of the super-bandwidth stitch with one hop per GPU (tsdrgpu_superb_shard_*): hop 0's reference
// spectrum goes from its rank to everybody, every rank's hop spectrum to everybody (in place: rank r's part sits at
// r * count_per_rank of the buffer).
extern "C" int tsdrgpu_comm_broadcast_f32(tsdrgpu_comm_t *c, float *d_buf, int64_t count, int root, int lane)
{
    if (!c || !d_buf || count < 0 || root < 0 || root >= c->world) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = c->g;
    const RcclApi *r = rccl();
    if (!r) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_comm_broadcast_f32", "librccl.so.1 not found");
    hipStream_t st = tsdr_lane_stream(g, lane);
    if (!st) return tsdr_fail(g, TSDRGPU_EINVAL, "broadcast", "bad lane");
    const ncclResult_t rc = r->Broadcast(d_buf, d_buf, (size_t)count, 7 /* ncclFloat32 */, root, c->comm, st);
    if (rc != ncclSuccess_) return tsdr_fail(g, TSDRGPU_EHIP, "ncclBroadcast", r->GetErrorString(rc));
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_comm_allgather_f32(tsdrgpu_comm_t *c, float *d_buf, int64_t count_per_rank, int lane)
{
    if (!c || !d_buf || count_per_rank < 0) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = c->g;
    const RcclApi *r = rccl();
    if (!r) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_comm_allgather_f32", "librccl.so.1 not found");
    hipStream_t st = tsdr_lane_stream(g, lane);
    if (!st) return tsdr_fail(g, TSDRGPU_EINVAL, "allgather", "bad lane");
    const ncclResult_t rc = r->AllGather(d_buf + (size_t)c->rank * (size_t)count_per_rank, d_buf, (size_t)count_per_rank, 7 /* ncclFloat32 */, c->comm, st);
    if (rc != ncclSuccess_) return tsdr_fail(g, TSDRGPU_EHIP, "ncclAllGather", r->GetErrorString(rc));
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_allreduce(tsdrgpu_autocorr_t *ac, tsdrgpu_comm_t *c, uint64_t total_windows)
{
    if (!ac || !c || total_windows == 0) return TSDRGPU_EINVAL;
    double *d_plots = nullptr;
    int64_t count = 0;
    int rc = tsdrgpu_autocorr_device_sums(ac, &d_plots, &count);  // the lags + the lag-0 scale of the certificate
    if (rc) return rc;
    rc = tsdrgpu_comm_allreduce_f64(c, d_plots, count, tsdrgpu_autocorr_lane(ac));
    if (rc) return rc;
    return tsdrgpu_autocorr_finalize_sums(ac, total_windows);
}
