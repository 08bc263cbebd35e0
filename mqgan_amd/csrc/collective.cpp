// Gradient-bucket all-reduce over RCCL behind the C ABI (`allreduce_bucket(ptr, n, stream)`; 8e: one exchange step
// per backward phase, SUM of the flat gradient arena's buckets over the ranks, launched on a side stream as buckets become ready).
//
// The reference is single-device (train.py:229-233 picks ONE torch.device; there is no DistributedDataParallel anywhere in it), so
// there is no reference code to cite for the arithmetic: the sum of the per-rank gradients of the per-rank mini-batches is what
// train.py:400-404 / :489-493 (backward -> clip_grad_norm_ -> step) would see on the global batch when the loss normalisers are
// global (mqgan_amd/comm.py, losses.py).
//
// RCCL is NOT a link-time dependency of libmqgan_hip.so: the five entry points it needs are resolved with dlopen / dlsym at the
// first mq_comm_* call.  A process that already maps an RCCL (PyTorch-ROCm ships its own librccl.so and loads it with
// libtorch_hip.so) gets THAT copy - two RCCL instances in one process would each own a set of IPC handles and proxy threads for
// the same GPU; a plain C consumer gets the system library.  MQGAN_RCCL_LIB names a file explicitly.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

#include "common.h"

namespace mq {
namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        const char *env = getenv("MQGAN_RCCL_LIB");
        if (env && *env) {
            r.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *n : names)                               // a copy this process already maps wins
                if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            for (const char *n : names)
                if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "RCCL not found (%s); set MQGAN_RCCL_LIB to librccl.so", dlerror());
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
            snprintf(r.why, sizeof(r.why), "the RCCL library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / "
                                           "ncclAllReduce / ncclGetErrorString");
            r.handle = nullptr;
        }
    });
    return &r;
}
}  // namespace
}  // namespace mq

struct mqComm {
    ncclComm_t comm;
    int world, rank;
};

#define MQ_RCCL(call, what)                                                                              \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess) return ::mq::fail("%s: %s", what, R->GetErrorString(r_));                 \
    } while (0)

extern "C" int mq_comm_unique_id(void *id_host)
{
    static_assert(sizeof(ncclUniqueId) == MQ_COMM_ID_BYTES, "mqgan_hip.h: MQ_COMM_ID_BYTES");
    mq::Rccl *R = mq::rccl();
    MQ_REQUIRE(R->handle, "mq_comm_unique_id: %s", R->why);
    MQ_REQUIRE(id_host, "mq_comm_unique_id: id_host is NULL");
    ncclUniqueId id;
    MQ_RCCL(R->GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_host, &id, sizeof(id));
    return 0;
}

extern "C" int mq_comm_init(mqComm **comm, int world, int rank, const void *id_host)
{
    mq::Rccl *R = mq::rccl();
    MQ_REQUIRE(R->handle, "mq_comm_init: %s", R->why);
    MQ_REQUIRE(comm && id_host, "mq_comm_init: comm / id_host is NULL");
    MQ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mq_comm_init: rank %d of %d", rank, world);
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t c;
    MQ_RCCL(R->CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    *comm = new mqComm{c, world, rank};
    return 0;
}

extern "C" int mq_allreduce_bucket(mqComm *comm, float *buf, size_t n, mqStream stream)
{
    mq::Rccl *R = mq::rccl();
    MQ_REQUIRE(R->handle, "mq_allreduce_bucket: %s", R->why);
    MQ_REQUIRE(comm, "mq_allreduce_bucket: comm is NULL");
    MQ_REQUIRE(buf || n == 0, "mq_allreduce_bucket: buf is NULL");
    if (n == 0) return 0;
    MQ_RCCL(R->AllReduce(buf, buf, n, ncclFloat32, ncclSum, comm->comm, mq::S(stream)), "ncclAllReduce");
    return 0;
}

extern "C" int mq_comm_destroy(mqComm *comm)
{
    if (!comm) return 0;
    mq::Rccl *R = mq::rccl();
    MQ_REQUIRE(R->handle, "mq_comm_destroy: %s", R->why);
    ncclResult_t r = R->CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) return mq::fail("ncclCommDestroy: %s", R->GetErrorString(r));
    return 0;
}
